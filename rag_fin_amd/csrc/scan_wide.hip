// Wide sweep: up to 256 queries against the corpus in ONE pass (dim 384).
//
// The 64-query kernel (scan.hip) keeps the queries in LDS and streams corpus
// fragments straight into registers; at 256 queries the roles flip:
//   * each of the 8 waves of a workgroup owns 32 queries and keeps their 24 B-operand
//     fragments in REGISTERS for the whole sweep (96 VGPRs, loaded once);
//   * a corpus block (32 rows, 24 KiB) is brought from HBM ONCE per workgroup and
//     shared by the 8 waves through LDS: wave w loads fragments 3w..3w+2 into
//     registers three blocks ahead (72 KiB in flight per CU), writes them to the
//     current LDS slot, one barrier, then every wave reads all 24 fragments
//     (linear 1-KiB images: conflict-free ds_read_b128) and runs 24 MFMAs.
// Arithmetic intensity is 256 flop per corpus byte: at the HBM rate the matrix pipe
// must run at ~2/3 of its peak, so this is the configuration where HBM and MFMA are
// both near their roofs (BASELINE.json configs[2], "HBM-roofline run").
// The lane-local filter, the sample/emit modes and the candidate lists are the ones
// of scan.hip (one query block per wave).
// Tried and dropped: every wave streaming the corpus itself through a register ring (no
// LDS, no barrier; 7 of 8 reads are L1/L2 hits) -- 447 us per 256-query step against
// 284 us for this form: the 8x load-instruction count saturates the TA path.
#include "scan_common.h"
#include <stdlib.h>

#define WIDE_KS 24
#define WIDE_WAVES 8
#define WIDE_DEPTH 3   // blocks in flight per wave (register ring)
#define WIDE_PIECES (WIDE_KS / WIDE_WAVES)

struct WideParams {
  const uint4* corpus;
  const _Float16* q;
  int B;
  uint32_t n_rows;
  uint32_t n_work;
  uint32_t bstride;
  const float* thr;
  uint32_t* cand_cnt;
  uint2* cand;
  uint32_t cap;
  float* pmax;
  int P;
};

template <int MODE>
__global__ void __launch_bounds__(WIDE_WAVES * 64, 2) k_scan_wide(WideParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* slots = (u32x4*)smem_raw;                                  // [2][24 * 64]
  uint32_t* stage = (uint32_t*)(slots + 2 * WIDE_KS * 64);          // 3 * WAVES * SCAP words

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int qi = wave * 32 + c;

  // this wave's 32 queries as B-operand fragments, resident for the whole sweep
  u32x4 qf[WIDE_KS];
#pragma unroll
  for (int kk = 0; kk < WIDE_KS; ++kk) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (qi < p.B) v = *(const u32x4*)(p.q + (size_t)qi * (WIDE_KS * 16) + kk * 16 + h * 8);
    qf[kk] = v;
  }
  float th[1];
  th[0] = (MODE == MODE_EMIT) ? p.thr[qi] : 0.f;
  float pm = -INFINITY;
  EmitState es;
  es.cnt = 0;
  es.q_base = (uint32_t)(wave * 32);
  es.s_row = stage + wave * SCAP;
  es.s_score = (float*)(stage + WIDE_WAVES * SCAP) + wave * SCAP;
  es.s_q = stage + 2 * WIDE_WAVES * SCAP + wave * SCAP;

  // work items of this WORKGROUP: w = blockIdx.x, + gridDim.x, ...
  const uint32_t G = gridDim.x;
  const uint32_t cnt = (p.n_work > blockIdx.x) ? (p.n_work - blockIdx.x + G - 1) / G : 0u;
  if (cnt == 0) return;   // whole workgroup (cnt is workgroup-uniform): no barrier is skipped by a subset
  // Every load below is UNCONDITIONAL (indices past the end are clamped to the last
  // block and their results discarded): with loads under `if (i + 3 < cnt)` hipcc
  // cannot count them and waits vmcnt(0) before every LDS write, which exposes a full
  // HBM latency per block (measured 260 us instead of ~130 us).
  auto piece = [&](uint32_t i, int j) {
    const uint32_t ic = i < cnt ? i : cnt - 1;
    const uint32_t b = (blockIdx.x + ic * G) * p.bstride;
    return p.corpus + ((size_t)b * WIDE_KS + wave * WIDE_PIECES + j) * 64 + lane;
  };

  u32x4 ring[WIDE_DEPTH][WIDE_PIECES];
#pragma unroll
  for (int d = 0; d < WIDE_DEPTH; ++d)
#pragma unroll
    for (int j = 0; j < WIDE_PIECES; ++j) ring[d][j] = ld_frag(piece(d, j));

  // Block i is computed from LDS slot i & 1 while block i+1 is being published to the
  // other slot: one barrier per block, and the LDS writes (with their wait on HBM) run
  // under the previous block's MFMAs instead of in front of the barrier.
  //   barrier_i  => every wave has finished block i-1 (its slot may be overwritten)
  //                 and every piece of block i (written during step i-1) is in LDS
  auto publish = [&](uint32_t blk, u32x4 (&mine)[WIDE_PIECES]) {
    u32x4* dst = slots + (blk & 1u) * (WIDE_KS * 64);
#pragma unroll
    for (int j = 0; j < WIDE_PIECES; ++j) dst[(wave * WIDE_PIECES + j) * 64 + lane] = mine[j];
  };
  auto step = [&](uint32_t i, u32x4 (&mine)[WIDE_PIECES]) {   // `mine` holds block i+1
    const u32x4* slot = slots + (i & 1u) * (WIDE_KS * 64);
    __syncthreads();
    publish(i + 1, mine);
#pragma unroll
    for (int j = 0; j < WIDE_PIECES; ++j) mine[j] = ld_frag(piece(i + 1 + WIDE_DEPTH, j));
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
    // A fragments come from LDS in batches of 6, double-buffered and pinned with
    // sched_barrier: left alone hipcc keeps only TWO fragments in flight, so every
    // second MFMA waits a full LDS latency (measured: 31 % MFMA utilisation, 252 us).
    {
      constexpr int NB = 6;
      u32x4 a0[NB], a1[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) a0[j] = slot[(0 * NB + j) * 64 + lane];
#pragma unroll
      for (int g = 0; g < WIDE_KS / NB; ++g) {
        u32x4(&cur)[NB] = (g & 1) ? a1 : a0;
        u32x4(&nxt)[NB] = (g & 1) ? a0 : a1;
        if (g + 1 < WIDE_KS / NB) {
#pragma unroll
          for (int j = 0; j < NB; ++j) nxt[j] = slot[((g + 1) * NB + j) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, cur[j]),
                                                          __builtin_bit_cast(half8, qf[g * NB + j]),
                                                          acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // padding steps (i >= cnt) re-score the last block; row0 = n_rows masks every row
    const uint32_t row0 = i < cnt ? (blockIdx.x + i * G) * p.bstride * 32u : p.n_rows;
    if (MODE == MODE_SAMPLE) {
      if (row0 + 32u > p.n_rows) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (row0 + acc_row(r, h) >= p.n_rows) acc[0][r] = -INFINITY;
      }
      pm = fmaxf(pm, max16(acc[0]));
    } else {
      // 16 compares OR-ed on the scalar unit: fmaxf on MFMA results costs an extra
      // canonicalising v_max per element, a compare does not
      bool hit = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) hit |= acc[0][r] >= th[0];
      if (__ballot(hit) != 0ull) {
        emit_slow<1>(acc, th, row0, lane, es, p);
      }
    }
  };

  publish(0, ring[0]);
#pragma unroll
  for (int j = 0; j < WIDE_PIECES; ++j) ring[0][j] = ld_frag(piece(WIDE_DEPTH, j));
  for (uint32_t i = 0; i < cnt; i += WIDE_DEPTH) {   // cnt rounded up to a multiple of the ring depth
    step(i, ring[1]);        // ring[(i + 1) % 3] holds block i + 1
    step(i + 1, ring[2]);
    step(i + 2, ring[0]);
  }

  if (MODE == MODE_EMIT) {
    if (es.cnt > 0) emit_flush(es, p, lane);
  } else {
    pm = fmaxf(pm, __shfl_xor(pm, 32));
    if (h == 0 && qi < p.B) p.pmax[(size_t)qi * p.P + blockIdx.x] = pm;
  }
}

// ---- host side --------------------------------------------------------------------------
template <int MODE>
static int launch_wide(const WideParams& p, int grid, hipStream_t st) {
  const size_t lds = (size_t)2 * WIDE_KS * RF_FRAG_BYTES + (size_t)3 * WIDE_WAVES * SCAP * 4;
  auto kern = k_scan_wide<MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    RF_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WIDE_WAVES * 64), lds, st, p);
  RF_HIP(hipGetLastError());
  return RF_OK;
}

int rf_wide_supported(const rf_index* ix) { return ix->KS == WIDE_KS; }

int rf_launch_wide_sample(const rf_index* ix, const void* q, int B, const rf_workspace& ws, int* P_out,
                          hipStream_t st) {
  const uint32_t nblk = (uint32_t)((ix->size + 31) / 32);
  // same sampling rule as scan.hip: ~1/16 of the corpus, 64..RF_SAMPLE_WGS partitions
  uint32_t n_work = nblk / 16;
  if (n_work < 64u) n_work = 64u;
  if (n_work > (uint32_t)RF_SAMPLE_WGS * 8) n_work = (uint32_t)RF_SAMPLE_WGS * 8;
  if (n_work > nblk) n_work = nblk;
  int grid = (int)(n_work < (uint32_t)RF_SAMPLE_WGS ? n_work : (uint32_t)RF_SAMPLE_WGS);
  WideParams p{};
  p.corpus = ix->tiles;
  p.q = (const _Float16*)q;
  p.B = B;
  p.n_rows = (uint32_t)ix->size;
  p.n_work = n_work;
  p.bstride = nblk / n_work;
  p.pmax = ws.pmax;
  p.P = grid;
  *P_out = grid;
  return launch_wide<MODE_SAMPLE>(p, grid, st);
}

int rf_launch_wide_emit(const rf_index* ix, const void* q, int B, const rf_workspace& ws,
                        hipStream_t st) {
  const uint32_t nblk = (uint32_t)((ix->size + 31) / 32);
  int grid = ix->num_cus;   // one 8-wave workgroup per CU
  if ((uint32_t)grid > nblk) grid = (int)nblk;
  if (grid < 1) grid = 1;
  WideParams p{};
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
  return launch_wide<MODE_EMIT>(p, grid, st);
}
