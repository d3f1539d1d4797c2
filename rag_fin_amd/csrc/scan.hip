// Streaming query x corpus scan on the matrix cores with a fused top-k filter.
// This is the arithmetic Milvus performs for the reference's
// Collection.search(..., {"metric_type": "COSINE"}, top_k)
// (vector_rag_mcp/main.py:51-57, retrieve.py:28-34), restated for gfx950:
//
//   * the corpus lives in HBM as 32-row blocks of KS = dim/16 MFMA A-fragments
//     (rf_internal.h); a wave streams a block with KS 1-KiB loads that land
//     directly in the operand registers of v_mfma_f32_32x32x16_f16 -- no LDS
//     round trip, no bank conflicts, every byte read exactly once;
//   * the (<= 64) queries of the sweep sit in LDS in B-fragment order and are
//     re-read per k-step (48 KB at dim 384);
//   * each lane owns ONE query column of the 32x32 result, so the top-k filter
//     is a lane-local compare against that query's threshold;
//   * the loads form a register ring: fragment t+R is requested right after
//     fragment t has been consumed, so every wave keeps R KiB (24 KiB at dim
//     384) in flight across block boundaries and filter work.
//
// Two filters share the loop:
//   MODE_SAMPLE  running max per lane over a strided sample of blocks ->
//                one partition maximum per workgroup and query.  The k-th
//                largest of those maxima is a lower bound of the final k-th
//                best score (merge.hip turns it into the emit threshold).
//   MODE_EMIT    every score >= threshold is appended to that query's
//                candidate list (wave-level ballot compaction into LDS,
//                flushed with one global atomic per entry).
#include "rf_internal.h"
#include <stdlib.h>
#include <string.h>

#include "scan_common.h"

template <int KS, int R, int JB, int MODE, bool LAST>
__device__ __forceinline__ void block_step(u32x4 (&ring)[R], const uint4* cur,
                                           const uint4* nxt, const u32x4* smemQ, int lane,
                                           uint32_t row0, float (&th)[JB], float (&pm)[JB],
                                           EmitState& es, const ScanParams& p) {
  static_assert(KS % R == 0, "ring must divide the block");
  // keep the query-fragment LDS reads inside the block: hoisted out of the
  // block loop they would pin JB*KS*4 registers and spill
  asm volatile("" ::: "memory");
  f32x16 acc[JB];
#pragma unroll
  for (int jb = 0; jb < JB; ++jb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[jb][i] = 0.f;

#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const half8 a = __builtin_bit_cast(half8, ring[kk % R]);
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
      const half8 b = __builtin_bit_cast(half8, smemQ[(jb * KS + kk) * 64 + lane]);
      acc[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[jb], 0, 0, 0);
    }
    // re-arm this ring slot with the fragment R steps ahead
    if (kk + R < KS) {
      ring[kk % R] = ld_frag(cur + (kk + R) * 64);
    } else if (!LAST) {
      ring[kk % R] = ld_frag(nxt + (kk + R - KS) * 64);
    }
  }

  if (MODE == MODE_SAMPLE) {
    if (row0 + 32u > p.n_rows) {  // wave-uniform: only the corpus' last block
      const int h = lane >> 5;
#pragma unroll
      for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (row0 + acc_row(i, h) >= p.n_rows) acc[jb][i] = -INFINITY;
    }
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) pm[jb] = fmaxf(pm[jb], max16(acc[jb]));
  } else {
    bool hit = false;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) hit |= (max16(acc[jb]) >= th[jb]);
    if (__ballot(hit) != 0ull) emit_slow<JB>(acc, th, row0, lane, es, p);
  }
}

template <int KS, int R, int JB, int WAVES, int MODE>
__global__ void __launch_bounds__(WAVES * 64, 2) k_scan(ScanParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* smemQ = (u32x4*)smem_raw;                                   // JB*KS*64 uint4
  unsigned char* tail = smem_raw + (size_t)JB * KS * RF_FRAG_BYTES;  // per-mode scratch

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dim = KS * 16;

  float th[JB];
  float pm[JB];
#pragma unroll
  for (int jb = 0; jb < JB; ++jb) {
    pm[jb] = -INFINITY;
    th[jb] = (MODE == MODE_EMIT) ? p.thr[jb * 32 + (lane & 31)] : 0.f;
  }
  EmitState es;
  es.cnt = 0;
  es.q_base = 0;
  if (MODE == MODE_EMIT) {
    es.s_row = (uint32_t*)tail + wave * SCAP;
    es.s_score = (float*)((uint32_t*)tail + WAVES * SCAP) + wave * SCAP;
    es.s_q = (uint32_t*)tail + 2 * WAVES * SCAP + wave * SCAP;
  } else {
    es.s_row = nullptr;
    es.s_score = nullptr;
    es.s_q = nullptr;
  }

  // work items w = gw, gw + W, ...  (one item = one 32-row block)
  const uint32_t W = gridDim.x * WAVES;
  const uint32_t gw = blockIdx.x * WAVES + wave;
  const uint32_t cnt = (p.n_work > gw) ? (p.n_work - gw + W - 1) / W : 0u;

  u32x4 ring[R];
  if (cnt > 0) {
    const uint4* src = p.corpus + (size_t)gw * p.bstride * (KS * 64) + lane;
#pragma unroll
    for (int s = 0; s < R; ++s) ring[s] = ld_frag(src + s * 64);
  }
  // (the corpus stream starts BEFORE the queries are staged: the first HBM round trip runs
  // under the staging loop and its barrier instead of after them)
  // stage the queries in B-fragment order: lane (j = l & 31, h = l >> 5) of
  // fragment (jb, kk) holds q[32 jb + j][16 kk + 8 h .. +8)
  for (int idx = tid; idx < JB * KS * 64; idx += WAVES * 64) {
    const int l = idx & 63;
    const int kk = (idx >> 6) % KS;
    const int jb = idx / (64 * KS);
    const int qi = jb * 32 + (l & 31);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (qi < p.B) v = *(const u32x4*)(p.q + (size_t)qi * dim + kk * 16 + (l >> 5) * 8);
    smemQ[idx] = v;
  }

  __syncthreads();
  if (cnt > 0) {

    uint32_t w = gw;
    for (uint32_t i = 0; i + 1 < cnt; ++i, w += W) {
      const uint32_t b = w * p.bstride;
      const uint4* cur = p.corpus + (size_t)b * (KS * 64) + lane;
      const uint4* nxt = p.corpus + (size_t)(b + W * p.bstride) * (KS * 64) + lane;
      block_step<KS, R, JB, MODE, false>(ring, cur, nxt, smemQ, lane, b * 32u, th, pm, es, p);
    }
    {
      const uint32_t b = w * p.bstride;
      const uint4* cur = p.corpus + (size_t)b * (KS * 64) + lane;
      block_step<KS, R, JB, MODE, true>(ring, cur, cur, smemQ, lane, b * 32u, th, pm, es, p);
    }
  }

  if (MODE == MODE_EMIT) {
    if (es.cnt > 0) emit_flush(es, p, lane);
  } else {
    // workgroup partition maximum per query: max over waves and lane halves
    float* red = (float*)tail;  // [WAVES*2][JB*32]
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
      red[(wave * 2 + (lane >> 5)) * (JB * 32) + jb * 32 + (lane & 31)] = pm[jb];
    __syncthreads();
    if (tid < JB * 32) {
      float m = -INFINITY;
      for (int s = 0; s < WAVES * 2; ++s) m = fmaxf(m, red[s * (JB * 32) + tid]);
      p.pmax[(size_t)tid * p.P + blockIdx.x] = m;
    }
  }
}

// ---- raw score dump (test hook) ---------------------------------------------
template <int KS>
__global__ void __launch_bounds__(64) k_debug_scores(const uint4* corpus, const _Float16* q, int B,
                                                      uint32_t n, float* out) {
  // one wave per (block, 32-query group); plain loads, no ring
  const int lane = threadIdx.x;
  const uint32_t b = blockIdx.x;
  const int jb = blockIdx.y;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int qi = jb * 32 + (lane & 31);
  for (int kk = 0; kk < KS; ++kk) {
    const uint4 av = corpus[((size_t)b * KS + kk) * 64 + lane];
    uint4 bv = make_uint4(0, 0, 0, 0);
    if (qi < B) bv = *(const uint4*)(q + (size_t)qi * (KS * 16) + kk * 16 + (lane >> 5) * 8);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, av),
                                                 __builtin_bit_cast(half8, bv), acc, 0, 0, 0);
  }
  if (qi < B) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t row = b * 32u + acc_row(i, lane >> 5);
      if (row < n) out[(size_t)qi * n + row] = acc[i];
    }
  }
}

// ---- host side ----------------------------------------------------------------
template <int KS, int R, int JB, int WAVES, int MODE>
static int launch_scan(const ScanParams& p, int grid, hipStream_t st) {
  size_t lds = (size_t)JB * KS * RF_FRAG_BYTES;
  if (MODE == MODE_EMIT) lds += (size_t)3 * WAVES * SCAP * 4;
  else lds += (size_t)WAVES * 2 * JB * 32 * 4;
  auto kern = k_scan<KS, R, JB, WAVES, MODE>;
  static rf_lds_attr attr;  // per instantiation, per device
  RF_HIP(rf_ensure_lds(attr, (const void*)kern, lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, p);
  RF_HIP(hipGetLastError());
  return RF_OK;
}

// ---- tuning knobs: compile-time constants in the shipped library (rf_internal.h); the
// experiments build (tools/ only) makes them process-wide ints behind rf_set_tuning -----------
#ifdef RF_EXPERIMENTS
int rf_knob_ring24 = 8, rf_knob_emit_wgs_per_cu = 0, rf_knob_sample_bpw = 2;
int rf_knob_wide_sample_pairs = 4, rf_knob_wide_dbg = 0, rf_knob_wide_ne = 0, rf_knob_wide_form = 0;
int rf_knob_linear_dma = 1, rf_knob_linear_small = 1, rf_knob_k384_ntb = 4, rf_knob_ffn2_ntb = 4;
int rf_knob_gemm_tile = 3, rf_knob_encode_graph = 1, rf_knob_linear_dbg = 0, rf_knob_debug_epi = 1, rf_knob_att_heads = 1, rf_knob_gemm_tile_dma = 0, rf_knob_one_query = 1, rf_knob_post_block = 1, rf_knob_post_dbg = 0, rf_knob_post_qkv = 1;
int rf_tuning_generation = 0;
void* rf_debug_buffer = nullptr;
extern "C" int rf_debug_set_buffer(void* dev_ptr) {
  rf_debug_buffer = dev_ptr;
  return RF_OK;
}
extern "C" int rf_set_tuning(const char* key, int value) {
  if (!key) return RF_ERR_INVALID;
  ++rf_tuning_generation;   // cached encode graphs were captured under the old settings
  struct K { const char* name; int* var; int lo, hi; };
  const K keys[] = {{"ring24", &rf_knob_ring24, 6, 24}, {"emit_wgs_per_cu", &rf_knob_emit_wgs_per_cu, 0, 4},
                    {"sample_bpw", &rf_knob_sample_bpw, 1, 8}, {"wide_sample_pairs", &rf_knob_wide_sample_pairs, 1, 8},
                    {"wide_dbg", &rf_knob_wide_dbg, 0, 127}, {"wide_ne", &rf_knob_wide_ne, 0, 112}, {"wide_form", &rf_knob_wide_form, 0, 1}, {"linear_dma", &rf_knob_linear_dma, 0, 3},
                    {"linear_small", &rf_knob_linear_small, 0, 1}, {"k384_ntb", &rf_knob_k384_ntb, 2, 4},
                    {"ffn2_ntb", &rf_knob_ffn2_ntb, 2, 4}, {"gemm_tile", &rf_knob_gemm_tile, 0, 15}, {"encode_graph", &rf_knob_encode_graph, 0, 1},
                    {"linear_dbg", &rf_knob_linear_dbg, 0, 63}, {"debug_epi", &rf_knob_debug_epi, 0, 5}, {"post_block", &rf_knob_post_block, 0, 1}, {"post_dbg", &rf_knob_post_dbg, 0, 511}, {"post_qkv", &rf_knob_post_qkv, 0, 1}, {"att_heads", &rf_knob_att_heads, 1, 2}, {"one_query", &rf_knob_one_query, 0, 1}, {"gemm_tile_dma", &rf_knob_gemm_tile_dma, 0, 2}};
  for (const K& k : keys)
    if (!strcmp(key, k.name) && value >= k.lo && value <= k.hi) {
      if (k.var == &rf_knob_ring24 && value != 6 && value != 8 && value != 12 && value != 24) break;
      if ((k.var == &rf_knob_k384_ntb || k.var == &rf_knob_ffn2_ntb) && value == 3) break;
      *k.var = value;
      return RF_OK;
    }
  rf_set_error("rf_set_tuning: unknown key or bad value (%s = %d)", key, value);
  return RF_ERR_INVALID;
}
#endif

template <int MODE>
static int dispatch_scan(int KS, int JB, const ScanParams& p, int grid4, int grid8,
                         hipStream_t st) {
#define RF_CASE(ks, r, waves, grid)                                              \
  case ks:                                                                       \
    return JB == 1 ? launch_scan<ks, r, 1, waves, MODE>(p, grid, st)             \
                   : launch_scan<ks, r, 2, waves, MODE>(p, grid, st);
  // dim 384: the emit sweep runs best with a SHALLOW ring (8 fragments = 8 KiB per
  // wave in flight: 119 us vs 124 us at 24 -- deeper queues only add latency once
  // HBM is saturated), the short sample pass with the full-block ring (16 vs 21 us:
  // it has two blocks per wave and must prefetch the second during the first).
  if (KS == 24 && MODE == MODE_EMIT) {
    // short sweeps (a few blocks per wave: shards of a strong-scaled job, BASELINE configs[1]) are
    // latency-bound like the sample pass and prefer the full-block ring too: 21.7 vs 23.3 us at 100 k rows
    const int ring = (p.n_work < 6u * 4u * (uint32_t)grid4) ? 24 : rf_knob_ring24;
    switch (ring) {
      case 8:
        return JB == 1 ? launch_scan<24, 8, 1, 4, MODE>(p, grid4, st) : launch_scan<24, 8, 2, 4, MODE>(p, grid4, st);
#ifdef RF_EXPERIMENTS
      case 6:
        return JB == 1 ? launch_scan<24, 6, 1, 4, MODE>(p, grid4, st) : launch_scan<24, 6, 2, 4, MODE>(p, grid4, st);
      case 12:
        return JB == 1 ? launch_scan<24, 12, 1, 4, MODE>(p, grid4, st) : launch_scan<24, 12, 2, 4, MODE>(p, grid4, st);
#endif
      default:
        break;
    }
#undef RF_CASE
#define RF_CASE(ks, r, waves, grid)                                              \
  case ks:                                                                       \
    return JB == 1 ? launch_scan<ks, r, 1, waves, MODE>(p, grid, st)             \
                   : launch_scan<ks, r, 2, waves, MODE>(p, grid, st);
  }
  switch (KS) {
    RF_CASE(4, 4, 4, grid4)
    RF_CASE(8, 8, 4, grid4)
    RF_CASE(16, 16, 4, grid4)
    RF_CASE(24, 24, 4, grid4)
    RF_CASE(32, 16, 4, grid4)
    RF_CASE(48, 16, 8, grid8)
    RF_CASE(64, 16, 8, grid8)
    default:
      break;
  }
#undef RF_CASE
  rf_set_error("no scan kernel for dim %d", KS * 16);
  return RF_ERR_UNSUPPORTED;
}

int rf_scan_supported_dim(int dim) {
  switch (dim) {
    case 64: case 128: case 256: case 384: case 512: case 768: case 1024:
      return 1;
    default:
      return 0;
  }
}

static inline int waves_per_wg(int KS) { return KS >= 48 ? 8 : 4; }
static inline int wgs_per_cu(int KS) { return KS >= 48 ? 1 : 2; }

int rf_launch_sample(const rf_index* ix, const void* q, int B, int JB, const rf_workspace& ws,
                     int* P_out, hipStream_t st) {
  const int KS = ix->KS;
  const uint32_t nblk = (uint32_t)((ix->size + 31) / 32);
  const int WAVES = waves_per_wg(KS);
  // Sample ~1/16 of the corpus, spread evenly: candidates per query ~ k * N / n_sample
  // stay ~16 k whatever N is, and a small corpus does not pay a sample pass as long as
  // its scan.  At least 64 workgroups (partitions) so the k-th largest exists for
  // k <= 64, at most RF_SAMPLE_WGS workgroups x SAMPLE_BPW blocks per wave.
  const int SAMPLE_BPW = rf_knob_sample_bpw;
  uint32_t n_work = nblk / 16;
  const uint32_t lo = 64u * WAVES, hi = (uint32_t)RF_SAMPLE_WGS * WAVES * SAMPLE_BPW;
  if (n_work < lo) n_work = lo;
  if (n_work > hi) n_work = hi;
  if (n_work > nblk) n_work = nblk;
  // as many workgroups as there are waves' worth of blocks, capped at one per partition
  // slot; waves take blocks round-robin, so the load is balanced for any n_work
  int grid = (int)((n_work + WAVES - 1) / WAVES);
  if (grid > RF_SAMPLE_WGS) grid = RF_SAMPLE_WGS;
  if (grid < 1) grid = 1;
  const uint32_t bstride = nblk / n_work;  // >= 1
  ScanParams p{};
  p.corpus = ix->tiles;
  p.q = (const _Float16*)q;
  p.B = B;
  p.n_rows = (uint32_t)ix->size;
  p.n_work = n_work;
  p.bstride = bstride;
  p.pmax = ws.pmax;
  p.P = grid;
  *P_out = grid;
  return dispatch_scan<MODE_SAMPLE>(KS, JB, p, grid, grid, st);
}

int rf_launch_emit(const rf_index* ix, const void* q, int B, int JB, const rf_workspace& ws,
                   hipStream_t st) {
  const int KS = ix->KS;
  const uint32_t nblk = (uint32_t)((ix->size + 31) / 32);
  const int WAVES = waves_per_wg(KS);
  const int wgs_env = rf_knob_emit_wgs_per_cu;
  int grid = ix->num_cus * (wgs_env > 0 ? wgs_env : wgs_per_cu(KS));
  const uint32_t need = (nblk + WAVES - 1) / WAVES;
  if ((uint32_t)grid > need) grid = (int)need;
  if (grid < 1) grid = 1;
  ScanParams p{};
  p.corpus = ix->tiles;
  p.q = (const _Float16*)q;
  p.B = B;
  p.n_rows = (uint32_t)ix->size;
  p.n_work = nblk;
  p.bstride = 1;
  p.thr = ws.thr;
  p.cand_cnt = ws.cand_cnt;
  p.cand = ws.cand;
  p.cap = RF_SHARD_CAP;
  return dispatch_scan<MODE_EMIT>(KS, JB, p, grid, grid, st);
}

int rf_launch_debug_scores(const rf_index* ix, const void* q, int B, int64_t n, float* out,
                           hipStream_t st) {
  const uint32_t nblk = (uint32_t)((n + 31) / 32);
  const dim3 grid(nblk, (B + 31) / 32);
#define RF_DBG(ks)                                                                         \
  case ks:                                                                                 \
    hipLaunchKernelGGL(k_debug_scores<ks>, grid, dim3(64), 0, st, ix->tiles,               \
                       (const _Float16*)q, B, (uint32_t)n, out);                           \
    break;
  switch (ix->KS) {
    RF_DBG(4) RF_DBG(8) RF_DBG(16) RF_DBG(24) RF_DBG(32) RF_DBG(48) RF_DBG(64)
    default:
      rf_set_error("no debug kernel for dim %d", ix->dim);
      return RF_ERR_UNSUPPORTED;
  }
#undef RF_DBG
  RF_HIP(hipGetLastError());
  return RF_OK;
}
