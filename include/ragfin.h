/*
 * ragfin.h -- C ABI of libragfin_hip.so: the MI355X (gfx950) engine behind the
 * vector-retrieval hot path of rag-fin.
 *
 * Every entry point replaces a call the reference makes into one of its two
 * un-vendored dependencies (sentence-transformers, pymilvus -> Milvus server);
 * the reference call site each one stands in for is cited as
 * <file>:<line> relative to the reference tree.
 *
 * Conventions
 *   - plain C: pointers, sizes, opaque handles; no C++/torch types cross the ABI.
 *   - every function returns an int status (RF_OK == 0, negative on error);
 *     rf_last_error() gives the message for the calling thread.
 *   - the CALLER owns all device memory (corpus storage, workspaces, inputs,
 *     outputs); the library owns only small host-side handles.  Nothing here
 *     calls hipMalloc/hipFree, and nothing synchronises the host with the
 *     stream: work is enqueued on `stream` (a hipStream_t passed as void*) in
 *     order, so a caller may capture it into a hipGraph.
 *   - device pointers must be 16-byte aligned.
 *   - threading: an rf_index_t / rf_encoder_t may be used by any number of host threads at
 *     once as long as each concurrent call has its OWN workspace (and normally its own
 *     stream); the handles hold no per-call state.  rf_index_add_f16 / rf_index_reset must
 *     not run concurrently with a search on the same index.  A process may hold indexes and
 *     encoders on several devices; the calling thread's current HIP device must be the
 *     handle's device (hipSetDevice / torch.cuda.device).
 */
#ifndef RAGFIN_H
#define RAGFIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_OK 0
#define RF_ERR_INVALID (-1)    /* bad argument (null, misaligned, out of range) */
#define RF_ERR_UNSUPPORTED (-2) /* dim / k / device not supported by the kernels */
#define RF_ERR_CAPACITY (-3)   /* index full, or workspace/storage too small */
#define RF_ERR_HIP (-4)        /* a HIP runtime call failed */
#define RF_ERR_DEVICE (-5)     /* no gfx950 device */

/* per-query flag bits written by rf_search into flags_dev */
#define RF_FLAG_CAND_OVERFLOW 1u /* candidate buffer overflowed: result not proven exact */
#define RF_FLAG_TIE_OVERFLOW 2u  /* more near-ties than the rescoring set holds */

#define RF_MAX_K 64      /* largest top-k the fused scan path serves */
#define RF_QCHUNK 64     /* queries per corpus sweep */

typedef struct rf_index rf_index_t;
typedef struct rf_encoder rf_encoder_t;
typedef struct rf_comm rf_comm_t;      /* one rank's membership of a sharded search job (RCCL communicator) */

/* ---- library ---------------------------------------------------------- */
int rf_version(void);
/* Hex digest of the sources this binary was compiled from (every .hip / .h / .cpp under csrc/ and
 * this header), stamped at build time; rag_fin_amd/_lib.py compares it with the sources on disk and
 * rebuilds (hipcc present) or refuses to load (hipcc absent) a stale library. */
const char* rf_build_id(void);
const char* rf_last_error(void);
/* RF_OK when `device` is a gfx950 part the kernels were built for. */
int rf_device_check(int device);

/* ---- corpus index: replaces the Milvus collection -----------------------
 * Reference: schema + index build "chunking_storing (1).py":14-29 (FLOAT_VECTOR
 * dim 384, COSINE); the vectors live here as fp16 in an MFMA-fragment tiled
 * layout, scalar fields stay host-side in Python. */
size_t rf_index_storage_bytes(int dim, int64_t capacity_rows);
int rf_index_create(rf_index_t** out, int dim, int64_t capacity_rows,
                    void* storage_dev, size_t storage_bytes, int device);
int rf_index_destroy(rf_index_t* ix);
/* Collection.insert/flush/load -- "chunking_storing (1).py":383-396.
 * rows_dev: fp16 [n, dim] row-major.  Appends n rows (ids size .. size+n-1). */
int rf_index_add_f16(rf_index_t* ix, const void* rows_dev, int64_t n, void* stream);
/* Collection.num_entities -- vector_rag_mcp/main.py:113,120,164 */
int64_t rf_index_size(const rf_index_t* ix);
int rf_index_dim(const rf_index_t* ix);
/* drop + recreate -- "chunking_storing (1).py":25-28 */
int rf_index_reset(rf_index_t* ix, void* stream);
/* Collection.query(expr="id in [...]") vector fetch -- graph_cons.py:308-311.
 * rows_dev: int64 [n] row numbers; out_dev: fp16 [n, dim] row-major. */
int rf_index_get_rows_f16(const rf_index_t* ix, const int64_t* rows_dev, int64_t n,
                          void* out_dev, void* stream);
/* fp32 [n, dim] -> L2-normalised fp16 [n, dim] (what COSINE needs so that the
 * scan can use the inner product).  Reference: the normalise step of
 * SentenceTransformer.encode feeding Collection.insert, same file :380-394. */
int rf_normalize_f32_to_f16(const float* in_dev, int64_t n, int dim, int normalize,
                            void* out_dev, void* stream);

/* ---- search: replaces Collection.search(..., COSINE, top_k) ---------------
 * Reference: vector_rag_mcp/main.py:51-57, retrieve.py:28-34,
 * "chunking_storing (1).py":411-417, graph_cons.py:275-281.
 *
 * q_dev      fp16 [B, dim] row-major queries (L2-normalised by the caller for
 *            COSINE; raw for inner product)
 * scores_dev fp32 [B, k]   score of rank j (descending), -inf past the end
 * ids_dev    int64 [B, k]  row id + id_base, -1 past the end
 * exact_dev  fp64 [B, k]   (nullable) the un-rounded ranking scores, used by
 *            the cross-shard merge
 * flags_dev  uint32 [B]    RF_FLAG_* bits; 0 means the result is proven equal
 *            to the exact ranking by (score desc, id asc)
 * The four output pointers may also be device-visible host memory (pinned,
 * host-coherent): the last kernel then stores the result straight into it
 * and the caller only synchronises the stream -- worthwhile for query-sized
 * results (no copy command between the kernel and the host).
 */
size_t rf_search_workspace_bytes(const rf_index_t* ix);
int rf_search(const rf_index_t* ix, const void* q_dev, int B, int k, int64_t id_base,
              float* scores_dev, int64_t* ids_dev, double* exact_dev,
              uint32_t* flags_dev, void* workspace_dev, size_t workspace_bytes,
              void* stream);
/* Profiling variant of rf_search for the FIRST corpus sweep of the batch (min(B, 64) queries,
 * or min(B, 256) where rf_search would take the wide sweep: dim 384, B > 64): same
 * launches, with HIP events around each stage.  SYNCHRONISES the stream.
 * stage_ms_host (host memory) receives {sample scan, threshold, emit scan, merge}
 * in milliseconds.  Measurement hook for bench.py; no reference counterpart. */
int rf_search_profile(const rf_index_t* ix, const void* q_dev, int B, int k, int64_t id_base,
                      float* scores_dev, int64_t* ids_dev, double* exact_dev,
                      uint32_t* flags_dev, void* workspace_dev, size_t workspace_bytes,
                      void* stream, float* stage_ms_host);
/* Slow, unconditionally exact path (fp64 scores of every row); used for the
 * queries rf_search flagged.  Same outputs. */
int rf_search_exhaustive(const rf_index_t* ix, const void* q_dev, int B, int k,
                         int64_t id_base, float* scores_dev, int64_t* ids_dev,
                         double* exact_dev, void* workspace_dev, size_t workspace_bytes,
                         void* stream);
/* Paging for limits above RF_MAX_K (the reference's hybrid consumer asks for
 * limit=1000, graph_cons.py:275-281): the next k hits ranked strictly AFTER the
 * per-query bound (after_score_dev fp64 [B], after_id_dev int64 [B] = the last hit
 * of the previous page, ids including id_base).  Exhaustive fp64 path. */
int rf_search_exhaustive_after(const rf_index_t* ix, const void* q_dev, int B, int k,
                               int64_t id_base, const double* after_score_dev,
                               const int64_t* after_id_dev, float* scores_dev, int64_t* ids_dev,
                               double* exact_dev, void* workspace_dev, size_t workspace_bytes,
                               void* stream);
/* Cross-shard merge after the RCCL all-gather: in [W, B, k] (exact fp64, id
 * int64) -> out [B, k] by (score desc, id asc).  New in this build (the
 * reference is single-process); see SURVEY.md 8e. */
int rf_merge_shards(const double* exact_dev, const int64_t* ids_dev, int W, int B, int k,
                    float* scores_out_dev, int64_t* ids_out_dev, void* stream);
/* Same merge on the all-gather's own layout.  Each rank's send buffer is
 * rf_packed_shard_words(B, k) int64 words:
 *   [0, B k)        the fp64 ranking scores (bit patterns)   <- rf_search exact_dev
 *   [B k, 2 B k)    the global row ids                       <- rf_search ids_dev
 *   [2 B k, ...)    uint32 flags[B] (RF_FLAG_*), zero-padded <- rf_search flags_dev
 * so rf_search writes its outputs straight into ONE send buffer and no repacking kernel runs
 * between the scan, the collective and the merge.  packed_dev = the gathered [W] buffers.
 * flags_out_dev (nullable) uint32 [B]: OR of the W shards' flags -- identical on every rank, so
 * all ranks agree on which queries to re-run through rf_search_exhaustive (a query whose LOCAL
 * answer was unproven on ANY shard has an unproven merged answer). */
size_t rf_packed_shard_words(int B, int k);
int rf_merge_shards_packed(const int64_t* packed_dev, int W, int B, int k,
                           float* scores_out_dev, int64_t* ids_out_dev,
                           uint32_t* flags_out_dev, void* stream);
/* Shards that are not contiguous in the global row numbering (a store that grows by appending a
 * slice of every insert to every rank): rf_search runs with id_base 0 and this call turns the
 * LOCAL row numbers it wrote into global ids through the shard's table, in place, on `stream`:
 * ids_dev[i] = id_map_dev[ids_dev[i]] for ids >= 0 (-1 = "no hit" stays; a row number past n_map
 * becomes -1).  Between the scan and the all-gather: the N > 1 step is then four enqueues
 * (rf_search, rf_map_ids, ncclAllGather, rf_merge_shards_packed) and no host library touches
 * the ids.  New in this build; stands beside vector_rag_mcp/main.py:51-57. */
int rf_map_ids(int64_t* ids_dev, int64_t n, const int64_t* id_map_dev, int64_t n_map, void* stream);
/* ---- the sharded step as ONE call (rf_comm_init / rf_search_sharded; 8e) ----------
 * One process per GPU; rank r holds a row shard in its rf_index_t.  The reference has no counterpart
 * (one Milvus server answers vector_rag_mcp/main.py:51-57); this is what stands in for it at N > 1.
 *   rf_comm_unique_id   rank 0 draws the job's 128-byte id (RF_COMM_ID_BYTES, host memory); the HOST
 *                       ships it to the other ranks (MPI, a socket, torch.distributed, a file)
 *   rf_comm_init        every rank: ncclCommInitRank on `device` -- blocks until all `world` ranks arrive
 *   rf_search_sharded   every rank, same B and k, calls in the same order on all ranks: rf_search into
 *                       the packed send buffer, rf_map_ids when id_map_dev is given (id_base is then
 *                       ignored), ncclAllGather of rf_packed_shard_words(B, k) words, rf_merge_shards_packed.
 *                       Four enqueues on `stream`, no host synchronisation; scores_dev / ids_dev hold the
 *                       GLOBAL top-k on every rank, flags_dev (nullable) the OR of the shards' RF_FLAG_*
 *                       bits -- a flagged query is re-run through rf_search_exhaustive on every shard and
 *                       merged again by the host, exactly as for one GPU.
 *                       scratch_dev: rf_search_sharded_scratch_words(comm, B, k) int64 words of device
 *                       memory owned by the caller, not shared between steps that may be in flight together.
 * One collective at a time per communicator: the caller serialises rf_search_sharded calls on one rf_comm_t
 * (different streams are fine as long as every rank enqueues them in the same order).
 * RCCL is bound at run time (dlopen of librccl.so, or RAGFIN_RCCL_PATH; a copy the process has already
 * loaded is reused): where it is absent these calls return RF_ERR_UNSUPPORTED and the rest of the
 * library works.  rag_fin_amd/sharded.py runs the same four enqueues from Python (rag_fin_amd/rccl.py
 * binds the same library). */
#define RF_COMM_ID_BYTES 128
int rf_comm_unique_id(void* id_out);
int rf_comm_init(int rank, int world, const void* id, int device, rf_comm_t** out);
int rf_comm_destroy(rf_comm_t* comm);
int rf_comm_rank(const rf_comm_t* comm);
int rf_comm_world(const rf_comm_t* comm);
size_t rf_search_sharded_scratch_words(const rf_comm_t* comm, int B, int k);
int rf_search_sharded(const rf_index_t* ix, rf_comm_t* comm, const void* q_dev, int B, int k,
                      int64_t id_base, const int64_t* id_map_dev, int64_t n_map,
                      float* scores_dev, int64_t* ids_dev, uint32_t* flags_dev,
                      void* workspace_dev, size_t workspace_bytes,
                      int64_t* scratch_dev, size_t scratch_words, void* stream);
/* Test hook: raw MFMA scan scores fp32 [B, n] for the first n rows. */
int rf_debug_scores(const rf_index_t* ix, const void* q_dev, int B, int64_t n,
                    float* out_dev, void* stream);

#ifdef RF_EXPERIMENTS
/* ---- experiments build only (python -m rag_fin_amd.build --experiments ->
 * libragfin_hip_exp.so; used by tools/, never by the product or the tests) -------------------
 * The shipped library has no run-time tuning surface: the knobs are compile-time constants
 * (csrc/rf_internal.h).  In the experiments build they are process-wide ints, NOT thread-safe
 * against concurrent searches.  Keys: "ring24" (6|8|12|24), "emit_wgs_per_cu" (0..4),
 * "sample_bpw" (1..8), "wide_sample_pairs" (1..8), "wide_dbg"; encoder: "linear_dma" (0..3),
 * "linear_small", "encode_graph" (0|1), "k384_ntb", "ffn2_ntb" (2|4), "linear_dbg", "debug_epi". */
int rf_set_tuning(const char* key, int value);
/* byte offset of a named array ("pmax", "cand", "thr") inside a search workspace */
size_t rf_debug_workspace_offset(const char* field);
/* a device buffer (>= 64 KiB, or NULL to switch off) that instrumented kernels fill with
 * clock stamps (encoder k_linear_dma: 8 floats per wave) */
int rf_debug_set_buffer(void* dev_ptr);
#endif

/* ---- embedder: replaces SentenceTransformer('all-MiniLM-L6-v2').encode ----
 * Reference: vector_rag_mcp/main.py:41,50; retrieve.py:14,27;
 * "chunking_storing (1).py":8,380,408. */
typedef struct rf_encoder_config {
  int32_t vocab_size, hidden, layers, heads, intermediate, max_position, type_vocab;
  float ln_eps;
} rf_encoder_config;

/* All pointers are device fp16 unless noted; per-layer arrays are
 * [layers] x tensor, contiguous.  Linear weights are stored [out, in]
 * (PyTorch nn.Linear layout). */
typedef struct rf_encoder_weights {
  const void* word_emb;   /* [vocab, H] */
  const void* pos_emb;    /* [max_position, H] */
  const void* type_emb;   /* [type_vocab, H] */
  const void* emb_ln_g;   /* [H] */
  const void* emb_ln_b;   /* [H] */
  const void* qkv_w;      /* [L, 3H, H]  (q;k;v stacked on the out axis) */
  const void* qkv_b;      /* [L, 3H] */
  const void* ao_w;       /* [L, H, H] */
  const void* ao_b;       /* [L, H] */
  const void* ln1_g;      /* [L, H] */
  const void* ln1_b;      /* [L, H] */
  const void* ff1_w;      /* [L, I, H] */
  const void* ff1_b;      /* [L, I] */
  const void* ff2_w;      /* [L, H, I] */
  const void* ff2_b;      /* [L, H] */
  const void* ln2_g;      /* [L, H] */
  const void* ln2_b;      /* [L, H] */
} rf_encoder_weights;

/* Bytes of caller-owned device storage the encoder needs for its MFMA-tiled copy
 * of the four Linear weights per layer (0 if cfg is unsupported). */
size_t rf_encoder_storage_bytes(const rf_encoder_config* cfg);
/* Tiles the Linear weights into storage_dev on `stream` and keeps the remaining
 * pointers of `w` (embeddings, biases, LayerNorm) -- the caller keeps those
 * tensors alive for the encoder's lifetime.  Supported: hidden == 384,
 * head_dim == 32, intermediate == 1536 (the all-MiniLM-L{6,12}-H384 family); anything else
 * returns RF_ERR_UNSUPPORTED. */
int rf_encoder_create(rf_encoder_t** out, const rf_encoder_config* cfg,
                      const rf_encoder_weights* w, void* storage_dev, size_t storage_bytes,
                      int device, void* stream);
int rf_encoder_destroy(rf_encoder_t* enc);
size_t rf_encode_workspace_bytes(const rf_encoder_t* enc, int B, int T);
/* Query-sized calls (B * T <= 1024): the launch sequence is captured once per (B, T, buffer
 * pointers) into a hipGraph owned by the encoder handle and replayed on `stream`
 * afterwards; callers that want the replay keep their buffers at fixed addresses
 * (rag_fin_amd.embedder does).
 * ids_dev int32 [B, T] (padded), lens_dev int32 [B] (valid tokens per row).
 * out_f16_dev fp16 [B, H] and/or out_f32_dev fp32 [B, H] (either nullable):
 * masked mean-pool + L2-normalise of the last hidden state.
 * T <= max_position (RF_ERR_INVALID beyond).  T <= 256 -- the reference model's max_seq_length,
 * vector_rag_mcp/main.py:41 -- runs the MFMA attention; 256 < T <= max_position is supported and
 * tested (tests/test_encoder_gpu.py, T = 257, 300, 384, 512 against the fp64 oracle at the same
 * tolerances) but takes a scalar attention kernel: correct, several times slower per token.
 * Batches of >= 8192 token slots take the fused per-layer path (csrc/encoder_post.hip); the cached
 * hipGraphs are an LRU of 32 entries, an evicted one is destroyed after its last launch has finished. */
int rf_encode(const rf_encoder_t* enc, const int32_t* ids_dev, const int32_t* lens_dev,
              int B, int T, void* out_f16_dev, float* out_f32_dev,
              void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- tokenizer: the text -> token-id stage in front of rf_encode ------------------------
 * Reference: the WordPiece tokenizer SentenceTransformer('all-MiniLM-L6-v2') loads by name
 * (vector_rag_mcp/main.py:41,50; "chunking_storing (1).py":8,380).  Host code, multi-threaded.
 * ASCII text is tokenised end to end; text with non-ASCII characters must be pre-normalised by
 * the caller (rag_fin_amd/tokenizer.py does it with Python's unicodedata: clean, NFC, lower,
 * NFD, strip Mn, CJK / non-ASCII punctuation padded with spaces).
 * vocab_utf8: the vocabulary file's bytes, one token per line, line i = id i. */
typedef struct rf_tokenizer rf_tokenizer_t;
int rf_tokenizer_create(rf_tokenizer_t** out, const char* vocab_utf8, size_t vocab_bytes,
                        int do_lower_case, int max_chars_per_word);
int rf_tokenizer_destroy(rf_tokenizer_t* t);
/* Non-ASCII code points to split off as punctuation (Unicode category P*), so that text whose
 * non-ASCII characters are all "simple" (caseless, no decomposition, not a mark / space /
 * control / CJK ideograph) needs no pre-normalisation.  cps: int32 [n], any order. */
int rf_tokenizer_set_punctuation(rf_tokenizer_t* t, const int32_t* cps, int n);
/* ids5 <- { [UNK], [CLS], [SEP], [PAD], [MASK] (-1 if absent) } */
int rf_tokenizer_special_ids(const rf_tokenizer_t* t, int32_t* ids5);
/* char_offsets int64 [n + 1] (ascending, in code points, char_offsets[0] = 0) -> byte_offsets int64 [n + 1]
 * into text_bytes (the n texts' UTF-8 bytes back to back, n_bytes in all): lets a host encode a whole batch
 * with one call of its runtime and hand over per-text CHARACTER counts.  -1 if the counts do not match the
 * blob.  Host helper of rf_tokenize_batch; no reference counterpart. */
int rf_utf8_offsets(const char* text_bytes, int64_t n_bytes, const int64_t* char_offsets, int n,
                    int64_t* byte_offsets);
/* text_bytes: the n texts' UTF-8 bytes back to back, text i = [offsets[i], offsets[i+1]).
 * ids_out int32 [n, max_len] ([CLS] ids [SEP], padded with [PAD]); lens_out int32 [n].
 * n_threads <= 0: one per hardware thread (at most 64). */
int rf_tokenize_batch(const rf_tokenizer_t* t, const char* text_bytes, const int64_t* offsets, int n,
                      int max_len, int32_t* ids_out, int32_t* lens_out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* RAGFIN_H */
