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
#include "lds_ring.h"
#include <stdlib.h>

#define WIDE_KS 24

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
  uint32_t dbg;   // diagnostic bits (experiments build, tools/bench_wide.py): 1 = LDS-DMA pieces all re-read one cached KiB,
                  // 4 = clock stamps (cycles, 100-MHz ticks, cycles in wait+barrier) per workgroup into pmax
};

int rf_wide_supported(const rf_index* ix) { return ix->KS == WIDE_KS; }

// The corpus arrives by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, the
// fragment image is lane-linear so HBM order == LDS order): no staging registers, no ds_write
// pass.  A phase = 2 blocks (64 rows, 48 KiB contiguous in HBM); 3 LDS slots; loads run two
// phases (96 KiB per CU) ahead behind a counted vmcnt and a raw s_barrier, one piece per group
// of MFMAs (an LDS-DMA issue holds its wave for tens of cycles);
//   * the A-fragment reads are inline asm with hand-counted lgkmcnt waits (as compiler-visible
//     LDS loads each would get an s_waitcnt vmcnt(0): hipcc cannot tell them from the DMA
//     writes in flight), one group of 4 ahead of the MFMAs;
//   * the filter of a phase's first block is plain VALU in the same basic block as the second
//     block's MFMAs; the second block's filter runs at the start of the NEXT phase, in the
//     shadow of that phase's first LDS reads.
// NW = 8: two waves per SIMD with 32 queries each (one LDS read per MFMA, the partner wave's
// MFMAs cover the other's non-matrix work).  Forms tried and dropped (in the
// history before round 2): a register-staged ring with one barrier per block (264 us per step),
// NW = 4 with 64 queries per wave (271 us), queries pinned to the accumulator half (+3 us).
#define WL_SLOTS 3
#define WL_PB 2                               // corpus blocks per phase
#define WL_FRAGS (WIDE_KS * WL_PB)            // 1-KiB fragments per phase (48)
#define WL_STAGE_WORDS 3072                   // emit staging, all waves: NW * CAP entries * 3 words

__device__ __forceinline__ float vmax3(float a, float b, float c) {
  // v_max3_f32 straight on the MFMA results (fmaxf adds a canonicalising v_max per element)
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max16_v3(const f32x16& a_in) {
  // The v_max3 below are inline asm: hipcc's hazard recognizer does not see them read MFMA
  // results, so the wait states an XDL write needs before a VALU read (11 for the 8-pass
  // 32x32x16) must be supplied by hand when the accumulator lives in arch VGPRs (with AGPR
  // accumulators the compiler's own v_accvgpr_read carried them).  The nops take the
  // accumulator as an in/out operand, which orders them after the MFMA and before the reads.
  f32x16 a = a_in;
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a));
  float m = vmax3(a[0], a[1], a[2]);
  m = vmax3(m, a[3], a[4]);
  m = vmax3(m, a[5], a[6]);
  m = vmax3(m, a[7], a[8]);
  m = vmax3(m, a[9], a[10]);
  m = vmax3(m, a[11], a[12]);
  m = vmax3(m, a[13], a[14]);
  return vmax3(m, a[15], a[15]);
}

// ABL: compile-time ablation of the diagnostic builds (tools/bench_wide.py --dbg 8|16|32, results
// are then wrong): 1 = no LDS-DMA in the loop, 2 = no filters, 4 = no LDS fragment reads
template <int MODE, int AUX, int NW, int ABL = 0, int PIN = 0>
__global__ void __launch_bounds__(NW * 64, 1) k_scan_ldsdma(WideParams p) {
  constexpr int JBW = RF_QWIDE / 32 / NW;        // query blocks per wave: 2 | 1
  // NW = 4 must park its 192 query registers in the accumulator half; NW = 8 (96 of them, 256
  // registers per wave) may keep EVERYTHING in arch VGPRs, and then the MFMA results need no
  // v_accvgpr_read before the filter's v_max3 (PIN = 1 pins for NW = 8 too: wide_variant 3, the A/B arm)
  constexpr bool PIN_Q = (NW == 4) || PIN;
  constexpr int PW = WL_FRAGS / NW;              // LDS-DMA pieces per wave and phase: 12 | 6
  constexpr int CAP = WL_STAGE_WORDS / 3 / NW;   // emit staging entries per wave: 256 | 128
  static_assert(JBW * 32 * NW == RF_QWIDE && PW * NW == WL_FRAGS && WIDE_KS % PW == 0, "tiling");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // ONE shared array: [WL_SLOTS][48 fragments][64 lanes] uint4, the emit staging words, and a
  // 1-KiB dump area for the pieces issued past the end of the stream
  u32x4* slots = (u32x4*)smem_raw;
  uint32_t* stage = (uint32_t*)(slots + WL_SLOTS * WL_FRAGS * 64);
  u32x4* const dump = (u32x4*)(stage + WL_STAGE_WORDS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;

  // work items (block PAIRS) of this workgroup: u = blockIdx.x, + gridDim.x, ...
  const uint32_t G = gridDim.x;
  const uint32_t cnt = (p.n_work > blockIdx.x) ? (p.n_work - blockIdx.x + G - 1) / G : 0u;
  if (cnt == 0) return;  // workgroup-uniform
  const uint32_t nblk = (p.n_rows + 31u) >> 5;

  // wave w brings fragments PW w .. PW w + PW - 1 of a phase (block (PW w) / 24 of the pair).
  // Every phase issues exactly PW pieces per wave, so the vmcnt arithmetic is the same in the
  // last phases: past the end the pieces re-read the corpus' last block into the dump area (an
  // L2 hit, no HBM traffic).
  struct Pieces {
    const uint4* src;
    u32x4* dst;
    int dstep, sstep;
  };
  auto pieces_of = [&](uint32_t ph) {
    const bool live = ph < cnt;
    uint32_t b = (blockIdx.x + ph * G) * p.bstride * WL_PB + (uint32_t)((wave * PW) / WIDE_KS);
    b = (live && b < nblk) ? b : nblk - 1u;   // odd tail: re-read the last block (masked by row0 below)
    Pieces pc;
    pc.src = p.corpus + ((size_t)b * WIDE_KS + (wave * PW) % WIDE_KS) * 64 + lane;
    pc.dst = live ? slots + ((ph % WL_SLOTS) * WL_FRAGS + wave * PW) * 64 : dump;
    pc.dstep = live ? 64 : 0;
    pc.sstep = 64;
    if (p.dbg & 1u) {   // ablation: same instruction stream, every piece re-reads one cached KiB
      pc.src = p.corpus + lane;
      pc.sstep = 0;
    }
    return pc;
  };
  auto issue_piece = [&](const Pieces& pc, int j) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pc.src + j * pc.sstep),
                                     (__attribute__((address_space(3))) void*)(pc.dst + j * pc.dstep), 16, 0, AUX);
  };

  // the first two phases of the corpus stream start before anything else
  {
    const Pieces p0 = pieces_of(0), p1 = pieces_of(1);
#pragma unroll
    for (int j = 0; j < PW; ++j) issue_piece(p0, j);
#pragma unroll
    for (int j = 0; j < PW; ++j) issue_piece(p1, j);
  }

  // this wave's 32 JBW queries as B-operand fragments, resident for the whole sweep
  u32x4 qf[JBW][WIDE_KS];
  float th[JBW];
#pragma unroll
  for (int jb = 0; jb < JBW; ++jb) {
    const int qi = (wave * JBW + jb) * 32 + c;
    const int qc = qi < p.B ? qi : p.B - 1;   // unconditional loads (no branch per fragment)
#pragma unroll
    for (int kk = 0; kk < WIDE_KS; ++kk)
      qf[jb][kk] = *(const u32x4*)(p.q + (size_t)qc * (WIDE_KS * 16) + kk * 16 + h * 8);
    th[jb] = (MODE == MODE_EMIT) ? p.thr[qc] : 0.f;
    if (qi >= p.B || p.dbg) th[jb] = INFINITY;
  }
  // all loads are in flight before the first is touched.  Then pin the resident query fragments
  // to the ACCUMULATOR half of the register file (MFMA reads A/B operands from either half): the
  // arch half stays free for the LDS fragment pipeline -- left alone hipcc packs them into arch
  // VGPRs and serialises the ds_reads.
#pragma unroll
  for (int jb = 0; jb < JBW; ++jb) {
    const int qi = (wave * JBW + jb) * 32 + c;
#pragma unroll
    for (int kk = 0; kk < WIDE_KS; ++kk) {
      u32x4 v = qf[jb][kk];
      if (qi >= p.B) v = u32x4{0u, 0u, 0u, 0u};
      if (PIN_Q) asm volatile("" : "+a"(v));
      else asm volatile("" : "+v"(v));
      qf[jb][kk] = v;
    }
  }
  float pm[JBW];
#pragma unroll
  for (int jb = 0; jb < JBW; ++jb) pm[jb] = -INFINITY;
  EmitState es;
  es.cnt = 0;
  es.q_base = (uint32_t)(wave * JBW * 32);
  es.s_row = stage + wave * CAP;
  es.s_score = (float*)(stage + NW * CAP) + wave * CAP;
  es.s_q = stage + 2 * NW * CAP + wave * CAP;

  // Score filter of one 32-row block (all query blocks of the wave).  Sample: running maximum
  // per lane (= per query); emit: any score >= the query's threshold sends the wave down the
  // compaction path.  max16 by v_max3 on the raw MFMA results.
  auto filter = [&](f32x16 (&a)[JBW], uint32_t row0) {
    if (ABL & 2) {
#pragma unroll
      for (int jb = 0; jb < JBW; ++jb) asm volatile("" : "+a"(a[jb]));   // keep the MFMAs alive
      return;
    }
    if (MODE == MODE_SAMPLE) {
      if (row0 + 32u > p.n_rows) {   // wave-uniform: the corpus' last block / a block past the end
#pragma unroll
        for (int jb = 0; jb < JBW; ++jb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (row0 + acc_row(r, h) >= p.n_rows) a[jb][r] = -INFINITY;
      }
#pragma unroll
      for (int jb = 0; jb < JBW; ++jb) pm[jb] = vmax3(pm[jb], max16_v3(a[jb]), pm[jb]);
    } else {
      unsigned long long jb_hit[JBW], any = 0ull;
#pragma unroll
      for (int jb = 0; jb < JBW; ++jb) {
        jb_hit[jb] = __ballot(max16_v3(a[jb]) >= th[jb]);
        any |= jb_hit[jb];
      }
      if (any != 0ull) emit_scan<JBW, CAP>(a, th, jb_hit, row0, lane, es, p);
    }
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 acc0[JBW], acc1[JBW];   // acc1 outlives its phase: it is filtered at the start of the next one
#pragma unroll
  for (int jb = 0; jb < JBW; ++jb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[jb][r] = -INFINITY;
  uint32_t row1_prev = p.n_rows;
  uint64_t t_wait = 0, t_c0 = 0, t_r0 = 0;
  if (p.dbg & 4u) {
    t_c0 = __builtin_amdgcn_s_memtime();
    t_r0 = __builtin_amdgcn_s_memrealtime();
  }
  for (uint32_t ph = 0; ph < cnt; ++ph) {
    // my pieces of phase ph have landed (the PW pieces of phase ph+1 may stay in flight) ...
    uint64_t ts0 = 0;
    if (p.dbg & 4u) ts0 = __builtin_amdgcn_s_memtime();
    if (ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
    // ... and after the barrier everybody's have, and everybody has consumed phase ph-1
    __builtin_amdgcn_s_barrier();
    if (p.dbg & 4u) t_wait += __builtin_amdgcn_s_memtime() - ts0;
    // phase ph+2 goes into the slot phase ph-1 has just vacated, ONE piece per MFMA group
    const Pieces nxt = pieces_of(ph + 2);

    const u32x4* slot = slots + (ph % WL_SLOTS) * (WL_FRAGS * 64) + lane;
    const uint32_t sa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)slot;
    u32x4 fa[2][WL_GRP];
    constexpr int NG = WL_FRAGS / WL_GRP;
    lds_read_group<0>(fa[0], sa);
    if (ABL & 4) lds_read_group<4>(fa[1], sa);
    __builtin_amdgcn_sched_barrier(0);   // the reads go first: the filter below is their latency cover
    // the one flush site of the loop (emit_scan never flushes)
    if (MODE == MODE_EMIT && es.cnt >= (uint32_t)CAP / 4) emit_flush(es, p, lane);
    // the filter of the PREVIOUS phase's second block
    filter(acc1, row1_prev);
    const uint32_t b0 = (blockIdx.x + ph * G) * p.bstride * WL_PB;
    const uint32_t row0 = (b0 < nblk) ? b0 * 32u : p.n_rows;
    row1_prev = (b0 + 1u < nblk) ? (b0 + 1u) * 32u : p.n_rows;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (ABL & 4) {
        lds_wait_group<0>(fa[g & 1]);
      } else if (g + 1 < NG) {
        lds_read_group_dyn(fa[(g + 1) & 1], sa, (g + 1) * WL_GRP);
        lds_wait_group<WL_GRP>(fa[g & 1]);
      } else {
        lds_wait_group<0>(fa[g & 1]);
      }
      if (g < PW && !(ABL & 1)) issue_piece(nxt, g);
#pragma unroll
      for (int j = 0; j < WL_GRP; ++j) {
        const int f = g * WL_GRP + j;          // fragment of the phase: block f / 24, k-step f % 24
        const int blk = f / WIDE_KS, kk = f % WIDE_KS;
        const half8 a = __builtin_bit_cast(half8, fa[g & 1][j]);
        f32x16(&dst)[JBW] = blk ? acc1 : acc0;
#pragma unroll
        for (int jb = 0; jb < JBW; ++jb)
          dst[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, qf[jb][kk]),
                                                           kk ? dst[jb] : zero16, 0, 0, 0);
      }
      // the first block's filter: plain VALU in the same basic block as the second block's MFMAs
      if (g == NG / 2) filter(acc0, row0);
    }
  }
  filter(acc1, row1_prev);   // the last phase's second block

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dump pieces must land before the LDS is handed on
  if ((p.dbg & 4u) && p.pmax && lane == 0) {   // diagnostic run only (tools/bench_wide.py --dbg 4)
    float* o = p.pmax + ((size_t)blockIdx.x * NW + wave) * 4;
    o[0] = (float)(__builtin_amdgcn_s_memtime() - t_c0);
    o[1] = (float)(__builtin_amdgcn_s_memrealtime() - t_r0);
    o[2] = (float)t_wait;
    o[3] = (float)cnt;
  }
  if (MODE == MODE_EMIT) {
    if (es.cnt > 0) emit_flush(es, p, lane);
  } else {
#pragma unroll
    for (int jb = 0; jb < JBW; ++jb) {
      const int qi = (wave * JBW + jb) * 32 + c;
      const float m = fmaxf(pm[jb], __shfl_xor(pm[jb], 32));
      if (h == 0 && qi < p.B) p.pmax[(size_t)qi * p.P + blockIdx.x] = m;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------
template <int MODE, int AUX, int NW, int ABL = 0, int PIN = 0>
static int launch_ldsdma(const WideParams& p, int grid, hipStream_t st) {
  const size_t lds = (size_t)WL_SLOTS * WL_FRAGS * RF_FRAG_BYTES + (size_t)WL_STAGE_WORDS * 4 +
                     (size_t)RF_FRAG_BYTES;   // slots, emit staging, dump area
  auto kern = k_scan_ldsdma<MODE, AUX, NW, ABL, PIN>;
  static rf_lds_attr attr;   // per instantiation, per device
  RF_HIP(rf_ensure_lds(attr, (const void*)kern, lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, p);
  RF_HIP(hipGetLastError());
  return RF_OK;
}
template <int MODE>
static int dispatch_ldsdma(const WideParams& p, int grid, hipStream_t st) {
#ifdef RF_EXPERIMENTS
  if (MODE == MODE_EMIT) {
    switch (rf_knob_wide_dbg >> 3) {   // compile-time ablations (results wrong): 1 no DMA, 2 no filters, 4 no LDS reads
#define WL_ABL(x) case x: return launch_ldsdma<MODE_EMIT, 2, 8, x>(p, grid, st);
      WL_ABL(1) WL_ABL(2) WL_ABL(4) WL_ABL(7)
#undef WL_ABL
      default: break;
    }
  }
#endif
  return launch_ldsdma<MODE, 2, 8>(p, grid, st);   // non-temporal LDS-DMA (aux = 2), 8 waves
}

int rf_launch_wide_sample(const rf_index* ix, const void* q, int B, const rf_workspace& ws, int* P_out,
                          hipStream_t st) {
  const uint32_t nblk = (uint32_t)((ix->size + 31) / 32);
  const uint32_t npair = (nblk + 1) / 2;
  // ~1/16 of the corpus in block pairs, 64..RF_SAMPLE_WGS partitions of up to 4 pairs
  uint32_t n_work = npair / 16;
  if (n_work < 64u) n_work = 64u;
  if (n_work > (uint32_t)RF_SAMPLE_WGS * rf_knob_wide_sample_pairs) n_work = (uint32_t)RF_SAMPLE_WGS * rf_knob_wide_sample_pairs;
  if (n_work > npair) n_work = npair;
  const int grid = (int)(n_work < (uint32_t)RF_SAMPLE_WGS ? n_work : (uint32_t)RF_SAMPLE_WGS);
  WideParams p{};
  p.corpus = ix->tiles;
  p.q = (const _Float16*)q;
  p.B = B;
  p.n_rows = (uint32_t)ix->size;
  p.n_work = n_work;
  p.bstride = npair / n_work;
  p.pmax = ws.pmax;
  p.P = grid;
  *P_out = grid;
  return dispatch_ldsdma<MODE_SAMPLE>(p, grid, st);
}

int rf_launch_wide_emit(const rf_index* ix, const void* q, int B, const rf_workspace& ws, hipStream_t st) {
  const uint32_t nblk = (uint32_t)((ix->size + 31) / 32);
  const uint32_t npair = (nblk + 1) / 2;
  int grid = ix->num_cus;   // one workgroup per CU (all of its LDS)
  if ((uint32_t)grid > npair) grid = (int)npair;
  if (grid < 1) grid = 1;
  WideParams p{};
  p.corpus = ix->tiles;
  p.q = (const _Float16*)q;
  p.B = B;
  p.n_rows = (uint32_t)ix->size;
  p.n_work = npair;
  p.bstride = 1;
  p.thr = ws.thr;
  p.cand_cnt = ws.cand_cnt;
  p.cand = ws.cand;
  p.cap = RF_SHARD_CAP;
  p.dbg = (uint32_t)rf_knob_wide_dbg;
  if (p.dbg & 4u) p.pmax = ws.pmax;   // stamp buffer of the diagnostic run
  return dispatch_ldsdma<MODE_EMIT>(p, grid, st);
}
