// Internal declarations shared by the HIP translation units of libragfin_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
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
  int num_cus;         // compute units of `device` (sizes the co-resident fused grid)
  mutable const void* ws_clean[8];  // workspaces whose control block this index has zeroed
  mutable int ws_clean_next;
  // rf_search calls that repeat with the same buffers (a serving lane, a shard's step) replay a
  // cached hipGraph of their 4-5 launches: a launch costs ~4.5 us of host time, which -- not the
  // kernels -- bounds the step on small corpora / shards
  struct Graph {
    const void* q;
    int B, k;
    int64_t id_base, size;
    void *scores, *ids, *exact, *flags, *ws;
    int tuning_gen;
    hipGraphExec_t exec;   // nullptr: key seen once (that plain run also zeroes the workspace, sets attributes)
    bool dead;
  };
  mutable std::vector<Graph> graphs;
  mutable hipStream_t cap_stream = nullptr;
  mutable std::mutex graph_mu;
};
extern int rf_tuning_generation;   // bumped by rf_set_tuning: cached graphs of older settings are not replayed

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
  uint32_t* gmax;      // [64][RF_MAX_K] fused scan: ordered-uint group maxima (zero = empty)
  uint32_t* bar;       // [16] fused scan: [0] arrivals, [1] give-up marker
  size_t ctl_bytes;    // bytes from the workspace base that must be zero before a search
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
// scan_fused.hip
int rf_launch_fused(const rf_index* ix, const void* q, int B, int JB, int k, const rf_workspace& ws,
                    hipStream_t st);
// merge.hip
int rf_launch_threshold(const rf_index* ix, const void* q, int B, int k, int P,
                        const rf_workspace& ws, hipStream_t st);
int rf_launch_merge(const rf_index* ix, const void* q, int B, int k, int64_t id_base,
                    const rf_workspace& ws, float* scores, int64_t* ids, double* exact,
                    uint32_t* flags, hipStream_t st);
int rf_launch_exhaustive(const rf_index* ix, const void* q, int B, int k, int64_t id_base,
                         const rf_workspace& ws, float* scores, int64_t* ids, double* exact,
                         const double* after_s, const int64_t* after_r, hipStream_t st);
int rf_launch_merge_shards(const double* exact, const int64_t* ids, size_t shard_stride, size_t lane_stride,
                           int W, int L, int B, int k, float* scores_out, int64_t* ids_out, hipStream_t st);

// order-preserving map float -> uint32 (larger float <=> larger uint)
__host__ __device__ inline uint32_t rf_f2ord(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float rf_ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __builtin_bit_cast(float, u);
}
