// Internal declarations shared by the HIP translation units of libragfin_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "../../include/ragfin.h"

// ---- tiled corpus layout ---------------------------------------------------
// The corpus is stored as 32-row blocks.  Block b holds KS = dim/16 "fragments"
// of 1 KiB; fragment kk is exactly the A operand of one
// v_mfma_f32_32x32x16_f16: lane l (r = l & 31, h = l >> 5) owns the 16 bytes
// row[32 b + r][16 kk + 8 h .. +8).  One wave-wide 16-byte load therefore reads
// 1 KiB of contiguous HBM straight into MFMA operand registers.
#define RF_BLOCK_ROWS 32
#define RF_FRAG_BYTES 1024

__host__ __device__ inline size_t rf_chunk_index(int64_t row, int chunk, int KS) {
  // index (in uint4 units) of 16-byte chunk `chunk` (dims 8*chunk..+8) of `row`
  const int64_t b = row >> 5;
  const int r = (int)(row & 31);
  return ((size_t)b * KS + (chunk >> 1)) * 64 + (size_t)((chunk & 1) * 32 + r);
}

struct rf_index {
  int dim;
  int KS;              // dim / 16
  int device;
  int64_t capacity;    // rows
  int64_t size;        // rows (host-side counter; adds are stream-ordered)
  uint4* tiles;        // device: capacity_blocks * KS * 64 uint4
  uint32_t* max_norm2; // device: bits of max squared row norm (float >= 0)
  size_t storage_bytes;
  int num_cus;         // compute units of `device`
};
// An index is immutable during searches (no mutable host state: any number of threads may
// search one index concurrently, each with its own workspace and stream); rf_index_add_f16 /
// rf_index_reset must not run concurrently with a search (include/ragfin.h, "Threading").

// ---- tuning knobs ------------------------------------------------------------------------
// The shipped library has NO run-time tuning surface: every knob below is a compile-time
// constant.  Built with -DRF_EXPERIMENTS (python -m rag_fin_amd.build --experiments ->
// libragfin_hip_exp.so, used by tools/ only) the same names are process-wide ints set through
// rf_set_tuning, for A/B runs in one process.
#ifdef RF_EXPERIMENTS
#define RF_KNOB(name, dflt) extern int name;
#else
#define RF_KNOB(name, dflt) static constexpr int name = dflt;
#endif
RF_KNOB(rf_knob_ring24, 8)             // register-ring depth (fragments) of the dim-384 emit sweep: 6 | 8 | 12 | 24
RF_KNOB(rf_knob_emit_wgs_per_cu, 0)    // emit grid = CUs x this (0 = default for the dim)
RF_KNOB(rf_knob_sample_bpw, 2)         // sample blocks per wave
RF_KNOB(rf_knob_wide_sample_pairs, 4)  // wide sample pass: block pairs per workgroup, at most
RF_KNOB(rf_knob_wide_dbg, 0)           // wide sweep diagnostic bits (clock stamps, cached-KiB ablation)
RF_KNOB(rf_knob_wide_form, 0)          // wide sweep kernel: 0 = eight waves x 32 queries (k_scan_w16), 1 = four waves x 64 queries (k_scan_w64)
RF_KNOB(rf_knob_wide_ne, 0)            // wide sweep: LDS-DMA pieces per phase of waves 0-3 (0 = the product's split)
RF_KNOB(rf_knob_linear_dma, 1)         // encoder: K = 384 plain-epilogue GEMMs through the LDS-DMA ring (0 off, 1 auto, 2 always 256-token, 3 never 256-token)
RF_KNOB(rf_knob_linear_small, 1)       // encoder: feature-split GEMMs + separate LayerNorm at <= 1024 token slots
RF_KNOB(rf_knob_k384_ntb, 4)           // encoder: token blocks per workgroup of the K = 384 LayerNorm GEMM at large batch
RF_KNOB(rf_knob_ffn2_ntb, 4)           // encoder: the same for the K = 1536 LayerNorm GEMM
RF_KNOB(rf_knob_gemm_tile, 3)          // encoder: GEMMs on k_gemm_tile (both operands through the LDS-DMA ring): 1 FFN2, 2 out-proj, 4 QKV, 8 FFN1
RF_KNOB(rf_knob_encode_graph, 1)       // encoder: query-sized forwards replay a cached hipGraph
RF_KNOB(rf_knob_linear_dbg, 0)         // encoder: k_linear_dma ablation bits (results wrong)
RF_KNOB(rf_knob_debug_epi, 1)          // encoder: which kernel writes clock stamps (0 QKV, 1 FFN1, 2 attention, 3 FFN2, 4 out-projection, 5 post block)
RF_KNOB(rf_knob_att_heads, 1)          // encoder: heads per attention workgroup (1 | 2)
RF_KNOB(rf_knob_one_query, 1)          // encoder: a single sequence of <= 32 tokens takes the fused QKV + attention launch
RF_KNOB(rf_knob_post_block, 1)         // encoder: out-projection + MLP of a layer as one launch at >= 8192 token slots (encoder_post.hip)
RF_KNOB(rf_knob_post_qkv, 1)           // encoder: k_post_block also computes the next layer's QKV projection
RF_KNOB(rf_knob_post_dbg, 0)           // encoder: k_post_block ablation bits (results wrong)
RF_KNOB(rf_knob_gemm_tile_dma, 0)      // encoder: k_gemm_tile LDS-DMA issue: 0 = halves take turns, 8 pieces in a burst (product); 1 = every wave 4 pieces between its MFMAs; 2 = none (ablation)
#undef RF_KNOB
#ifdef RF_EXPERIMENTS
extern int rf_tuning_generation;   // bumped by rf_set_tuning: cached encode graphs of older settings are not replayed
extern void* rf_debug_buffer;      // rf_debug_set_buffer: clock stamps of the diagnostic runs
#else
static constexpr int rf_tuning_generation = 0;
static constexpr void* rf_debug_buffer = nullptr;
#endif

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: cache what has
// been set per device, not per process (a second index / encoder on another GPU of the same
// process must get the attribute too).  Racing first calls both set it: idempotent.
#define RF_MAX_DEVICES 64
struct rf_lds_attr {
  std::atomic<uint32_t> bytes[RF_MAX_DEVICES];
};
static inline hipError_t rf_ensure_lds(rf_lds_attr& a, const void* fn, size_t lds) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= RF_MAX_DEVICES) return hipErrorInvalidDevice;
  if (a.bytes[dev].load(std::memory_order_relaxed) >= lds) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) a.bytes[dev].store((uint32_t)lds, std::memory_order_relaxed);
  return e;
}

// queries per wide sweep (scan_wide.hip); every per-query workspace array is sized for it
#define RF_QWIDE 256
// per-query candidate capacity of the fused scan: RF_CAND_SHARDS lists (picked by
// workgroup id) of RF_SHARD_CAP entries; the merge handles RF_CAND_CAP in total
#define RF_CAND_CAP 8192
#define RF_CAND_SHARDS 8
#define RF_SHARD_CAP 2048
// rows below which the sample pass is skipped (every row becomes a candidate)
#define RF_SMALL_ROWS 8192
// partition maxima per query produced by the sample pass (one per workgroup)
#define RF_SAMPLE_WGS 256
// rescoring-set capacity per query
#define RF_RESCORE_CAP 256

struct rf_workspace {
  float* thr;          // [64]
  float* eps;          // [64]
  uint32_t* cand_cnt;  // [64][RF_CAND_SHARDS]
  float* pmax;         // [64][RF_SAMPLE_WGS]
  uint2* cand;         // [64][RF_CAND_SHARDS][RF_SHARD_CAP]  {row, score bits}
  // exhaustive path
  double* ex_score;    // [RF_EX_LISTS][RF_MAX_K]
  int64_t* ex_row;     // [RF_EX_LISTS][RF_MAX_K]
};
#define RF_EX_WGS 256

void rf_set_error(const char* fmt, ...);
#define RF_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      rf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                   __LINE__);                                                     \
      return RF_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)

// index.hip: row-major fp16 [n, 16*KS] -> fragment-tiled (also used for encoder weights)
void rf_launch_tile_rows(const void* rows, uint4* tiles, int64_t first_row, int64_t n, int KS,
                         hipStream_t st);
// scan.hip
int rf_launch_sample(const rf_index* ix, const void* q, int B, int JB, const rf_workspace& ws,
                     int* P_out, hipStream_t st);
int rf_launch_emit(const rf_index* ix, const void* q, int B, int JB, const rf_workspace& ws,
                   hipStream_t st);
int rf_launch_debug_scores(const rf_index* ix, const void* q, int B, int64_t n, float* out,
                           hipStream_t st);
int rf_scan_supported_dim(int dim);
// scan_wide.hip
int rf_wide_supported(const rf_index* ix);
int rf_launch_wide_sample(const rf_index* ix, const void* q, int B, const rf_workspace& ws, int* P_out,
                          hipStream_t st);
int rf_launch_wide_emit(const rf_index* ix, const void* q, int B, const rf_workspace& ws, hipStream_t st);
// merge.hip
int rf_launch_threshold(const rf_index* ix, const void* q, int B, int k, int P,
                        const rf_workspace& ws, hipStream_t st);
int rf_launch_merge(const rf_index* ix, const void* q, int B, int k, int64_t id_base,
                    const rf_workspace& ws, float* scores, int64_t* ids, double* exact,
                    uint32_t* flags, hipStream_t st);
int rf_launch_exhaustive(const rf_index* ix, const void* q, int B, int k, int64_t id_base,
                         const rf_workspace& ws, float* scores, int64_t* ids, double* exact,
                         const double* after_s, const int64_t* after_r, hipStream_t st);
int rf_launch_merge_shards(const double* exact, const int64_t* ids, size_t shard_stride, int W, int B, int k,
                           float* scores_out, int64_t* ids_out, const uint32_t* flags_in, size_t flag_stride,
                           uint32_t* flags_out, hipStream_t st);

// order-preserving map float -> uint32 (larger float <=> larger uint)
__host__ __device__ inline uint32_t rf_f2ord(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float rf_ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __builtin_bit_cast(float, u);
}
