// extern "C" search entry points: orchestration of the scan / merge launches.
// Reference call site replaced: Collection.search(query_embedding, "embedding",
// {"metric_type": "COSINE"}, top_k, ...) -- vector_rag_mcp/main.py:51-57.
#include "rf_internal.h"
#include <stdlib.h>
#include <string.h>

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static size_t carve(unsigned char* base, rf_workspace* ws) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    unsigned char* p = base ? base + off : nullptr;
    off = align_up(off + bytes, 256);
    return p;
  };
  float* thr = (float*)take(RF_QWIDE * sizeof(float));
  float* eps = (float*)take(RF_QWIDE * sizeof(float));
  uint32_t* cnt = (uint32_t*)take((size_t)RF_QWIDE * RF_CAND_SHARDS * sizeof(uint32_t));
  float* pmax = (float*)take((size_t)RF_QWIDE * RF_SAMPLE_WGS * sizeof(float));
  uint2* cand = (uint2*)take((size_t)RF_QWIDE * RF_CAND_SHARDS * RF_SHARD_CAP * sizeof(uint2));
  double* exs = (double*)take((size_t)RF_QCHUNK * RF_EX_WGS * RF_MAX_K * sizeof(double));
  int64_t* exr = (int64_t*)take((size_t)RF_QCHUNK * RF_EX_WGS * RF_MAX_K * sizeof(int64_t));
  if (ws) {
    ws->thr = thr;
    ws->eps = eps;
    ws->cand_cnt = cnt;
    ws->pmax = pmax;
    ws->cand = cand;
    ws->ex_score = exs;
    ws->ex_row = exr;
  }
  return off;
}

#ifdef RF_EXPERIMENTS
// Diagnostic hook: byte offset of a named workspace array ("pmax", "cand", "thr").
extern "C" size_t rf_debug_workspace_offset(const char* field) {
  unsigned char* base = (unsigned char*)(uintptr_t)4096;   // never dereferenced
  rf_workspace ws;
  carve(base, &ws);
  if (field && !strcmp(field, "pmax")) return (size_t)((unsigned char*)ws.pmax - base);
  if (field && !strcmp(field, "cand")) return (size_t)((unsigned char*)ws.cand - base);
  if (field && !strcmp(field, "thr")) return (size_t)((unsigned char*)ws.thr - base);
  return (size_t)-1;
}
#endif

extern "C" size_t rf_search_workspace_bytes(const rf_index_t* ix) {
  (void)ix;
  return carve(nullptr, nullptr);
}

static int check_search_args(const char* fn, const rf_index_t* ix, const void* q, int B, int k,
                             const void* scores, const void* ids, const void* ws, size_t ws_bytes) {
  if (!ix || !q || !scores || !ids || !ws) {
    rf_set_error("%s: null argument", fn);
    return RF_ERR_INVALID;
  }
  if (B <= 0) {
    rf_set_error("%s: B = %d", fn, B);
    return RF_ERR_INVALID;
  }
  if (k <= 0 || k > RF_MAX_K) {
    rf_set_error("%s: k = %d outside 1..%d", fn, k, RF_MAX_K);
    return RF_ERR_UNSUPPORTED;
  }
  if ((((uintptr_t)q) & 15) || (((uintptr_t)ws) & 15)) {
    rf_set_error("%s: q / workspace must be 16-byte aligned", fn);
    return RF_ERR_INVALID;
  }
  if (ws_bytes < rf_search_workspace_bytes(ix)) {
    rf_set_error("%s: workspace %zu B < required %zu B", fn, ws_bytes,
                 rf_search_workspace_bytes(ix));
    return RF_ERR_CAPACITY;
  }
  return RF_OK;
}

static void fill_empty(int B, int k, float* scores, int64_t* ids, double* exact, uint32_t* flags,
                       hipStream_t st);

__global__ void k_fill_empty(int n, int B, float* scores, int64_t* ids, double* exact,
                             uint32_t* flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    scores[i] = -INFINITY;
    ids[i] = -1;
    if (exact) exact[i] = -INFINITY;
  }
  if (flags && i < B) flags[i] = 0u;
}

static void fill_empty(int B, int k, float* scores, int64_t* ids, double* exact, uint32_t* flags,
                       hipStream_t st) {
  const int n = B * k;
  hipLaunchKernelGGL(k_fill_empty, dim3((n + 255) / 256), dim3(256), 0, st, n, B, scores, ids,
                     exact, flags);
}

static int search_enqueue(const rf_index_t* ix, const void* q_dev, int B, int k, int64_t id_base, float* scores_dev,
                          int64_t* ids_dev, double* exact_dev, uint32_t* flags_dev, void* workspace_dev,
                          hipStream_t st);

// More than one 64-query sweep left and dim 384: one wide sweep of up to 256 queries.
// Small corpora -- every row a candidate -- stay on the 64-query kernel (its inline flushes take
// any hit density, the wide kernel's bounded staging would flag every query), and so do large k
// (the k-th of ~64 partition maxima is a weak threshold) and k-dense searches of mid-sized
// corpora (expected hits per wave and phase ~ 2^15 k / N against room for 96).
static bool take_wide(const rf_index_t* ix, int left, int k) {
  return rf_wide_supported(ix) && left > RF_QCHUNK && ix->size > RF_SMALL_ROWS && k <= 16 &&
         ix->size >= (int64_t)k * 1024;
}

extern "C" int rf_search(const rf_index_t* ix, const void* q_dev, int B, int k, int64_t id_base,
                         float* scores_dev, int64_t* ids_dev, double* exact_dev,
                         uint32_t* flags_dev, void* workspace_dev, size_t workspace_bytes,
                         void* stream) {
  int rc = check_search_args("rf_search", ix, q_dev, B, k, scores_dev, ids_dev, workspace_dev,
                             workspace_bytes);
  if (rc != RF_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  return search_enqueue(ix, q_dev, B, k, id_base, scores_dev, ids_dev, exact_dev, flags_dev, workspace_dev, st);
}

static int search_enqueue(const rf_index_t* ix, const void* q_dev, int B, int k, int64_t id_base, float* scores_dev,
                          int64_t* ids_dev, double* exact_dev, uint32_t* flags_dev, void* workspace_dev,
                          hipStream_t st) {
  int rc = RF_OK;
  if (ix->size == 0) {
    fill_empty(B, k, scores_dev, ids_dev, exact_dev, flags_dev, st);
    RF_HIP(hipGetLastError());
    return RF_OK;
  }
  rf_workspace ws;
  carve((unsigned char*)workspace_dev, &ws);
  const int dim = ix->dim;
  for (int q0 = 0; q0 < B;) {
    const int left = B - q0;
    const bool wide = take_wide(ix, left, k);
    const int nb = wide ? (left < RF_QWIDE ? left : RF_QWIDE) : (left < RF_QCHUNK ? left : RF_QCHUNK);
    const int JB = nb <= 32 ? 1 : 2;
    const _Float16* qc = (const _Float16*)q_dev + (size_t)q0 * dim;
    if (wide) {
      int P = 0;
      if (ix->size > RF_SMALL_ROWS) {
        rc = rf_launch_wide_sample(ix, qc, nb, ws, &P, st);
        if (rc != RF_OK) return rc;
      }
      rc = rf_launch_threshold(ix, qc, nb, k, P, ws, st);
      if (rc != RF_OK) return rc;
      rc = rf_launch_wide_emit(ix, qc, nb, ws, st);
      if (rc != RF_OK) return rc;
    } else {
      int P = 0;
      if (ix->size > RF_SMALL_ROWS) {
        rc = rf_launch_sample(ix, qc, nb, JB, ws, &P, st);
        if (rc != RF_OK) return rc;
      }
      rc = rf_launch_threshold(ix, qc, nb, k, P, ws, st);
      if (rc != RF_OK) return rc;
      rc = rf_launch_emit(ix, qc, nb, JB, ws, st);
      if (rc != RF_OK) return rc;
    }
    rc = rf_launch_merge(ix, qc, nb, k, id_base, ws, scores_dev + (size_t)q0 * k,
                         ids_dev + (size_t)q0 * k, exact_dev ? exact_dev + (size_t)q0 * k : nullptr,
                         flags_dev ? flags_dev + q0 : nullptr, st);
    if (rc != RF_OK) return rc;
    q0 += nb;
  }
  return RF_OK;
}

extern "C" int rf_search_profile(const rf_index_t* ix, const void* q_dev, int B, int k,
                                 int64_t id_base, float* scores_dev, int64_t* ids_dev,
                                 double* exact_dev, uint32_t* flags_dev, void* workspace_dev,
                                 size_t workspace_bytes, void* stream, float* stage_ms_host) {
  int rc = check_search_args("rf_search_profile", ix, q_dev, B, k, scores_dev, ids_dev,
                             workspace_dev, workspace_bytes);
  if (rc != RF_OK) return rc;
  if (!stage_ms_host || ix->size == 0) {
    rf_set_error("rf_search_profile: null stage buffer or empty index");
    return RF_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t ev[5];
  for (int i = 0; i < 5; ++i) RF_HIP(hipEventCreate(&ev[i]));
  rf_workspace ws;
  carve((unsigned char*)workspace_dev, &ws);
  const bool wide = take_wide(ix, B, k);   // the first sweep rf_search would run for this batch
  const int nb = wide ? (B < RF_QWIDE ? B : RF_QWIDE) : (B < RF_QCHUNK ? B : RF_QCHUNK);
  const int JB = nb <= 32 ? 1 : 2;
  int P = 0;
  RF_HIP(hipEventRecord(ev[0], st));
  if (ix->size > RF_SMALL_ROWS)
    rc = wide ? rf_launch_wide_sample(ix, q_dev, nb, ws, &P, st) : rf_launch_sample(ix, q_dev, nb, JB, ws, &P, st);
  RF_HIP(hipEventRecord(ev[1], st));
  if (rc == RF_OK) rc = rf_launch_threshold(ix, q_dev, nb, k, P, ws, st);
  RF_HIP(hipEventRecord(ev[2], st));
  if (rc == RF_OK) rc = wide ? rf_launch_wide_emit(ix, q_dev, nb, ws, st) : rf_launch_emit(ix, q_dev, nb, JB, ws, st);
  RF_HIP(hipEventRecord(ev[3], st));
  if (rc == RF_OK)
    rc = rf_launch_merge(ix, q_dev, nb, k, id_base, ws, scores_dev, ids_dev, exact_dev, flags_dev, st);
  RF_HIP(hipEventRecord(ev[4], st));
  RF_HIP(hipEventSynchronize(ev[4]));
  for (int i = 0; i < 4; ++i) RF_HIP(hipEventElapsedTime(&stage_ms_host[i], ev[i], ev[i + 1]));
  for (int i = 0; i < 5; ++i) (void)hipEventDestroy(ev[i]);
  return rc;
}

static int exhaustive_impl(const char* fn, const rf_index_t* ix, const void* q_dev, int B, int k,
                           int64_t id_base, float* scores_dev, int64_t* ids_dev, double* exact_dev,
                           const double* after_s, const int64_t* after_r, void* workspace_dev,
                           size_t workspace_bytes, void* stream) {
  int rc = check_search_args(fn, ix, q_dev, B, k, scores_dev, ids_dev, workspace_dev, workspace_bytes);
  if (rc != RF_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (ix->size == 0) {
    fill_empty(B, k, scores_dev, ids_dev, exact_dev, nullptr, st);
    RF_HIP(hipGetLastError());
    return RF_OK;
  }
  rf_workspace ws;
  carve((unsigned char*)workspace_dev, &ws);
  for (int q0 = 0; q0 < B; q0 += RF_QCHUNK) {
    const int nb = (B - q0) < RF_QCHUNK ? (B - q0) : RF_QCHUNK;
    rc = rf_launch_exhaustive(ix, (const _Float16*)q_dev + (size_t)q0 * ix->dim, nb, k, id_base, ws,
                              scores_dev + (size_t)q0 * k, ids_dev + (size_t)q0 * k,
                              exact_dev ? exact_dev + (size_t)q0 * k : nullptr,
                              after_s ? after_s + q0 : nullptr, after_r ? after_r + q0 : nullptr, st);
    if (rc != RF_OK) return rc;
  }
  return RF_OK;
}

extern "C" int rf_search_exhaustive(const rf_index_t* ix, const void* q_dev, int B, int k,
                                    int64_t id_base, float* scores_dev, int64_t* ids_dev,
                                    double* exact_dev, void* workspace_dev, size_t workspace_bytes,
                                    void* stream) {
  return exhaustive_impl("rf_search_exhaustive", ix, q_dev, B, k, id_base, scores_dev, ids_dev,
                         exact_dev, nullptr, nullptr, workspace_dev, workspace_bytes, stream);
}

extern "C" int rf_search_exhaustive_after(const rf_index_t* ix, const void* q_dev, int B, int k,
                                          int64_t id_base, const double* after_score_dev,
                                          const int64_t* after_id_dev, float* scores_dev,
                                          int64_t* ids_dev, double* exact_dev, void* workspace_dev,
                                          size_t workspace_bytes, void* stream) {
  if (!after_score_dev || !after_id_dev) {
    rf_set_error("rf_search_exhaustive_after: null bound arrays");
    return RF_ERR_INVALID;
  }
  return exhaustive_impl("rf_search_exhaustive_after", ix, q_dev, B, k, id_base, scores_dev, ids_dev,
                         exact_dev, after_score_dev, after_id_dev, workspace_dev, workspace_bytes, stream);
}

extern "C" int rf_merge_shards(const double* exact_dev, const int64_t* ids_dev, int W, int B, int k,
                               float* scores_out_dev, int64_t* ids_out_dev, void* stream) {
  if (!exact_dev || !ids_dev || !scores_out_dev || !ids_out_dev || W <= 0 || B <= 0 || k <= 0) {
    rf_set_error("rf_merge_shards: bad argument");
    return RF_ERR_INVALID;
  }
  return rf_launch_merge_shards(exact_dev, ids_dev, (size_t)B * k, W, B, k, scores_out_dev, ids_out_dev, nullptr, 0,
                                nullptr, (hipStream_t)stream);
}

extern "C" size_t rf_packed_shard_words(int B, int k) {
  if (B <= 0 || k <= 0) return 0;
  return (size_t)2 * B * k + ((size_t)B + 1) / 2;
}

extern "C" int rf_merge_shards_packed(const int64_t* packed_dev, int W, int B, int k,
                                      float* scores_out_dev, int64_t* ids_out_dev,
                                      uint32_t* flags_out_dev, void* stream) {
  if (!packed_dev || !scores_out_dev || !ids_out_dev || W <= 0 || B <= 0 || k <= 0) {
    rf_set_error("rf_merge_shards_packed: bad argument");
    return RF_ERR_INVALID;
  }
  // shard w = { fp64 score bits [B, k], int64 ids [B, k], uint32 flags [B] (padded to a whole word) }
  const size_t words = rf_packed_shard_words(B, k);
  return rf_launch_merge_shards((const double*)packed_dev, packed_dev + (size_t)B * k, words, W, B, k, scores_out_dev,
                                ids_out_dev, (const uint32_t*)(packed_dev + (size_t)2 * B * k), words * 2, flags_out_dev,
                                (hipStream_t)stream);
}

// local row numbers -> global ids through a table, in place (-1 = "no hit" stays -1)
__global__ void __launch_bounds__(256) k_map_ids(int64_t* __restrict__ ids, int64_t n, const int64_t* __restrict__ id_map,
                                                 int64_t n_map) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t v = ids[i];
  if (v >= 0) ids[i] = v < n_map ? id_map[v] : (int64_t)-1;
}

extern "C" int rf_map_ids(int64_t* ids_dev, int64_t n, const int64_t* id_map_dev, int64_t n_map, void* stream) {
  if (!ids_dev || n < 0 || n_map < 0 || (n_map > 0 && !id_map_dev)) {
    rf_set_error("rf_map_ids: bad argument");
    return RF_ERR_INVALID;
  }
  if (n == 0) return RF_OK;
  hipLaunchKernelGGL(k_map_ids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids_dev, n,
                     id_map_dev, n_map);
  RF_HIP(hipGetLastError());
  return RF_OK;
}

extern "C" int rf_debug_scores(const rf_index_t* ix, const void* q_dev, int B, int64_t n,
                               float* out_dev, void* stream) {
  if (!ix || !q_dev || !out_dev || B <= 0 || n <= 0 || n > ix->size) {
    rf_set_error("rf_debug_scores: bad argument");
    return RF_ERR_INVALID;
  }
  return rf_launch_debug_scores(ix, q_dev, B, n, out_dev, (hipStream_t)stream);
}
