// Wide sweep: up to 256 queries against the corpus in ONE pass (dim 384).
//
// The 64-query kernel (scan.hip) keeps the queries in LDS and streams corpus fragments straight
// into registers; at 256 queries the roles flip:
//   * each of the 8 waves of a workgroup owns 32 queries and keeps them as B-operand fragments in
//     REGISTERS for the whole sweep (96 VGPRs, loaded once);
//   * a corpus block (32 rows, 24 KiB) is brought from HBM ONCE per workgroup by LDS-DMA into a
//     3-slot ring of 2-block phases and shared by the 8 waves through LDS.
// Arithmetic intensity is 256 flop per corpus byte: HBM and the matrix pipe are both near their
// roofs (BASELINE.json configs[2], "HBM-roofline run").  The product kernel is k_scan_w16 below
// (round 2: v_mfma_f32_16x16x32_f16, unequal wave halves, priority by progress); the round-1 kernel
// k_scan_ldsdma (32x32x16, waves in lockstep) is compiled into the experiments build only, as the
// A/B arm of tools/bench_wide.py.  DESIGN.md 4.1b has the measurements and the forms that were
// tried and dropped (register-staged ring, 4 waves x 64 queries, every wave streaming the corpus
// itself: 447 us per step -- 8x the load instructions saturate the TA path).
#include "scan_common.h"
#include "lds_ring.h"
#include <stdlib.h>
#include <type_traits>

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
#ifdef RF_EXPERIMENTS   // the round-1 form (v_mfma_f32_32x32x16_f16, waves in lockstep): A/B arm of tools/bench_wide.py
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

#endif

// =========================================================================================
// The sweep (round 2): v_mfma_f32_16x16x32_f16, staggered wave halves
// =========================================================================================
// Same data path as before -- corpus by LDS-DMA into a 3-slot ring of 64-row phases, queries
// resident in registers, counted vmcnt + one raw s_barrier per phase, inline-asm ds_read_b128
// with counted lgkmcnt -- with three changes that the round-1 counters asked for
// (profiles/r01q_pmc_wide.json: matrix pipe busy 47 %, 2.4 VALU instructions per MFMA):
//   * 16x16x32 MFMAs.  Same FLOP per cycle as 32x32x16, but the chip -- which is power-bound
//     in this kernel (1.4-1.5 GHz under the 32x32x16 loop) -- holds a higher clock under them
//     (MI355X_MICROARCH.md "DVFS give-back" item 7: 1.12-1.15x the FLOP/s; own microbench
//     tools/micro/mfma_operand_bench.hip: 5-10 %).  The corpus image is unchanged: lane l of a
//     16-row x 32-k A operand reads 16 bytes of fragment 2 ks + (g >> 1), g = l >> 4, at source
//     lane 32 (g & 1) + 16 rg + (l & 15) -- within each 16-lane service group of ds_read_b128
//     the sixteen 16-byte slots are distinct mod 256 B, i.e. the permuted read is conflict-free.
//     A wave's tile per block is 2 row groups x 2 query groups; every A read feeds two MFMAs,
//     and the four accumulators rotate, so no MFMA waits on its predecessor.
//   * The two waves of a SIMD (w and w + 4) no longer run the phase in lockstep.  Waves 0-3
//     issue their six LDS-DMA pieces (60-185 cycles of issue stall each) in the first half of
//     the phase and filter at MFMA groups 0 and 6; waves 4-7 issue in the second half and filter
//     at groups 3 and 9: one wave's non-matrix work falls under its partner's MFMAs.
//   * Filter without copies: v_max3 reads the accumulators in place (the hazard nops take the
//     accumulator itself as in/out operand), 3 max3 + 1 max + 1 compare per query group and
//     block; the slow path is one v_cmp per accumulator register whose SGPR result IS the
//     ballot, and runs only for the query group that hit.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define W16_KS 12   // k-steps of 32 per row (dim 384)
#ifndef W16_AHEAD
#define W16_AHEAD 2   // groups of A-operand reads in flight ahead of their MFMAs (1 | 2)
#endif

// four A-operand reads at immediate offsets O0..O3 from the lane's slot address
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void w16_read4(rf_u32x4 (&d)[4], uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[0]) : "v"(addr), "n"(O0));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[1]) : "v"(addr), "n"(O1));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[2]) : "v"(addr), "n"(O2));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[3]) : "v"(addr), "n"(O3));
}
// group gi (0..11) of a phase: block gi / 6, k-steps 2 (gi % 6) and + 1, row groups 0 and 1
template <int GI>
__device__ __forceinline__ void w16_read_group(rf_u32x4 (&d)[4], uint32_t addr) {
  constexpr int base = (GI / 6) * (WIDE_KS * RF_FRAG_BYTES) + (2 * (GI % 6)) * 2048;
  w16_read4<base, base + 256, base + 2048, base + 2048 + 256>(d, addr);
}

struct W16Acc {
  f32x4 t[2][2];   // [row group][query group]: lane l = query 16 qg + (l & 15), rows 16 rg + 4 (l >> 4) + j
};

__device__ __forceinline__ float w16_max8(W16Acc& a, int qg) {
  // MFMA write -> VALU read wait states, by hand (the v_max3 are inline asm, invisible to the
  // hazard recognizer): the nops take the accumulator registers they guard as in/out
  asm volatile("s_nop 7\n\ts_nop 3" : "+v"(a.t[0][qg]), "+v"(a.t[1][qg]));
  const float m0 = vmax3(a.t[0][qg][0], a.t[0][qg][1], a.t[0][qg][2]);
  const float m1 = vmax3(a.t[0][qg][3], a.t[1][qg][0], a.t[1][qg][1]);
  const float m2 = vmax3(a.t[1][qg][2], a.t[1][qg][3], m0);
  return vmax3(m1, m2, m2);
}

// Emit staging (LDS, per wave): entries of 12 words = { row of score 0, query, the query's
// threshold, pad, 8 scores }: a lane whose 8-row column of a block holds a score >= threshold
// appends the WHOLE column (rows base + j and base + 16 + j, j = 0..3) with two ds_write_b128 and
// one ds_write_b96 -- a handful of instructions, the same for every hit, so a hit does not make
// its wave late at the phase barrier (the round-1 form scanned the sixteen accumulator registers
// with a ballot each: ~200 cycles per hit, and with ~3 hits per phase and workgroup nearly every
// phase waited for somebody's slow path).  The threshold test per score and the row bound are
// applied when the staging area is flushed, 64 scores at a time.
#define W16_ENTRY_WORDS 12
struct W16Stage {
  uint32_t* base;       // this wave's entries
  uint32_t base_addr;   // ... as an LDS byte address
  uint32_t cnt;         // wave-uniform
};

template <int CAP_E, class P>
__device__ __forceinline__ void w16_append(const W16Acc& a, int qg, unsigned long long mask, float th, uint32_t row0,
                                           uint32_t q, int lane, W16Stage& st, const P& p) {
  const uint32_t n = (uint32_t)__popcll(mask);
  const bool ok = (mask >> lane) & 1ull;
  if (st.cnt + n <= (uint32_t)CAP_E) {
    if (ok) {
      // inline-asm stores: a compiler-visible LDS store gets an s_waitcnt vmcnt(0) in front of it (hipcc
      // cannot tell it from the LDS-DMA writes in flight), i.e. every hit would drain the corpus ring
      const uint32_t e = st.base_addr + (st.cnt + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))) * (W16_ENTRY_WORDS * 4);
      const u32x4 hdr = {row0 + 4u * (uint32_t)(lane >> 4), q, __builtin_bit_cast(uint32_t, th), 0u};
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32"
                   :: "v"(e), "v"(hdr), "v"(a.t[0][qg]), "v"(a.t[1][qg]) : "memory");
    }
    st.cnt += n;
  } else if (ok) {   // staging full within one phase (adversarial duplicates): flag the query
    atomicAdd(&p.cand_cnt[q * RF_CAND_SHARDS + (blockIdx.x & (RF_CAND_SHARDS - 1))], p.cap + 1u);
  }
}

template <class P>
__device__ __forceinline__ void w16_flush(W16Stage& st, const P& p, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  for (uint32_t i = lane; i < st.cnt * 8u; i += 64) {
    const uint32_t* e = st.base + (i >> 3) * W16_ENTRY_WORDS;
    const uint32_t s = i & 7u;
    const float score = __builtin_bit_cast(float, e[4 + s]);
    const uint32_t row = e[0] + 16u * (s >> 2) + (s & 3u);
    if (score >= __builtin_bit_cast(float, e[2]) && row < p.n_rows) {
      const uint32_t list = e[1] * RF_CAND_SHARDS + (blockIdx.x & (RF_CAND_SHARDS - 1));
      const uint32_t slot = atomicAdd(&p.cand_cnt[list], 1u);
      if (slot < p.cap) p.cand[(size_t)list * p.cap + slot] = make_uint2(row, __builtin_bit_cast(uint32_t, score));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  st.cnt = 0;
}

// NE: LDS-DMA pieces per phase issued by each of waves 0-3 (waves 4-7 issue 12 - NE each).  The two
// waves of a SIMD are NOT treated alike by the hardware: at equal priority the older one wins every
// arbitration, the younger runs in what is left.  With the stall-prone work (LDS-DMA issue: 60-185
// cycles each) split evenly, the winner finished its phase after ~3 000 cycles and then waited
// ~1 650 cycles at the barrier, while the loser ran its own MFMAs AND its own stalls in the time
// that was left (in-kernel stamps, profiles/r02e_wide_stamps.txt: 4 684 cycles per phase against
// 3 072 of matrix work).  So the roles are made unequal on purpose: waves 0-3 run at s_setprio 1
// and carry (nearly) all the LDS-DMA issues -- their stalls are the slots in which waves 4-7, which
// do almost nothing but MFMAs, get the matrix pipe.
template <int MODE, int DBG, int NE>
__global__ void __launch_bounds__(512, 1) k_scan_w16(WideParams p) {
  constexpr int NW = 8;
  constexpr int NL = WL_FRAGS / 4 - NE;          // pieces per phase of each of waves 4-7
  static_assert(NE >= 6 && NE <= 12, "pieces per early wave");
  constexpr int CAP_E = WL_STAGE_WORDS / NW / W16_ENTRY_WORDS;   // emit staging entries per wave: 32
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // ONE shared array: [WL_SLOTS][48 fragments][64 lanes] uint4, the emit staging words, and a
  // 1-KiB dump area for the pieces issued past the end of the stream
  u32x4* slots = (u32x4*)smem_raw;
  uint32_t* stage = (uint32_t*)(slots + WL_SLOTS * WL_FRAGS * 64);
  u32x4* const dump = (u32x4*)(stage + WL_STAGE_WORDS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g = lane >> 4;

  // work items (block PAIRS) of this workgroup: u = blockIdx.x, + gridDim.x, ...
  const uint32_t G = gridDim.x;
  const uint32_t cnt = (p.n_work > blockIdx.x) ? (p.n_work - blockIdx.x + G - 1) / G : 0u;
  if (cnt == 0) return;  // workgroup-uniform
  const uint32_t nblk = (p.n_rows + 31u) >> 5;

  // A phase is 48 fragments (two blocks, contiguous in HBM).  Wave w < 4 brings fragments
  // NE w .. NE w + NE - 1, wave w >= 4 fragments 4 NE + NL (w - 4) .. + NL - 1.  Every phase issues
  // exactly the same number of pieces per wave, so the vmcnt arithmetic is the same in the last
  // phases: past the end the pieces re-read the corpus' last block into the dump area (an L2 hit,
  // no HBM traffic), and so does the second block of an odd tail (its rows are masked by row1).
  const bool early = wave < NW / 2;
  const uint32_t f0 = early ? (uint32_t)(wave * NE) : (uint32_t)(4 * NE + (wave - 4) * NL);
  struct Pieces {
    const uint4* src;   // lane's address of fragment 0 of the phase
    u32x4* dst;         // fragment 0 of the slot (or the dump area)
    int dstep;
    uint32_t cut;       // fragments >= cut come from 24 fragments further back (odd tail: the pair's second block does not exist)
  };
  auto pieces_of = [&](uint32_t ph) {
    const bool live = ph < cnt;
    uint32_t b = (blockIdx.x + ph * G) * p.bstride * WL_PB;
    Pieces pc;
    pc.cut = (live && b + 1u < nblk) ? 2u * WIDE_KS : (uint32_t)WIDE_KS;
    b = (live && b < nblk) ? b : nblk - 1u;
    pc.src = p.corpus + (size_t)b * WIDE_KS * 64 + lane;
    pc.dst = live ? slots + (ph % WL_SLOTS) * WL_FRAGS * 64 : dump;
    pc.dstep = live ? 64 : 0;
    return pc;
  };
  auto issue_piece = [&](const Pieces& pc, int j) {
    const uint32_t f = f0 + (uint32_t)j;
    const uint32_t fs = (DBG & 1) ? 0u : (f >= pc.cut ? f - (uint32_t)WIDE_KS : f);   // DBG 1: every piece re-reads one cached KiB
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pc.src + (size_t)fs * 64),
                                     (__attribute__((address_space(3))) void*)(pc.dst + f * pc.dstep), 16, 0, 2 /* nt */);
  };

  // the first two phases of the corpus stream start before anything else
  {
    const Pieces p0 = pieces_of(0), p1 = pieces_of(1);
    if (early) {
#pragma unroll
      for (int j = 0; j < NE; ++j) issue_piece(p0, j);
#pragma unroll
      for (int j = 0; j < NE; ++j) issue_piece(p1, j);
    } else {
#pragma unroll
      for (int j = 0; j < NL; ++j) issue_piece(p0, j);
#pragma unroll
      for (int j = 0; j < NL; ++j) issue_piece(p1, j);
    }
  }

  // this wave's 32 queries as B operands of the 16x16x32 MFMA, resident for the whole sweep:
  // qf[qg][ks], lane l = query 32 wave + 16 qg + (l & 15), dims 32 ks + 8 (l >> 4) .. + 8
  u32x4 qf[2][W16_KS];
  float th[2];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    const int qi = wave * 32 + qg * 16 + c16;
    const int qc = qi < p.B ? qi : p.B - 1;   // unconditional loads (no branch per fragment)
#pragma unroll
    for (int ks = 0; ks < W16_KS; ++ks)
      qf[qg][ks] = *(const u32x4*)(p.q + (size_t)qc * (WIDE_KS * 16) + ks * 32 + g * 8);
    th[qg] = (MODE == MODE_EMIT) ? p.thr[qc] : 0.f;
    if (qi >= p.B || (DBG & 1)) th[qg] = INFINITY;
  }
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    const int qi = wave * 32 + qg * 16 + c16;
#pragma unroll
    for (int ks = 0; ks < W16_KS; ++ks) {
      u32x4 v = qf[qg][ks];
      if (qi >= p.B) v = u32x4{0u, 0u, 0u, 0u};
      asm volatile("" : "+v"(v));   // resident: not to be re-materialised inside the loop
      qf[qg][ks] = v;
    }
  }
  float pm[2] = {-INFINITY, -INFINITY};
  W16Stage st;
  st.cnt = 0;
  st.base = stage + wave * (WL_STAGE_WORDS / NW);
  st.base_addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)st.base;

  // score filter of one 32-row block.  Sample: running maximum per lane and query group;
  // emit: any score >= the query's threshold sends the wave down the append path.
  auto filter = [&](W16Acc& a, uint32_t row0) __attribute__((always_inline)) {
    if (MODE == MODE_SAMPLE) {
      if (row0 + 32u > p.n_rows) {   // wave-uniform: the corpus' last block / a block past the end
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
          for (int qg = 0; qg < 2; ++qg)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (row0 + (uint32_t)(16 * rg + 4 * g + j) >= p.n_rows) a.t[rg][qg][j] = -INFINITY;
      }
#pragma unroll
      for (int qg = 0; qg < 2; ++qg) pm[qg] = vmax3(pm[qg], w16_max8(a, qg), pm[qg]);
    } else {
      const float m0 = w16_max8(a, 0), m1 = w16_max8(a, 1);
      const unsigned long long h0 = __builtin_amdgcn_fcmpf(m0, th[0], 3 /* FCMP_OGE */),
                               h1 = __builtin_amdgcn_fcmpf(m1, th[1], 3);
      if ((h0 | h1) != 0ull) {
        const uint32_t q0 = (uint32_t)(wave * 32 + c16);
        if (h0 != 0ull) w16_append<CAP_E>(a, 0, h0, th[0], row0, q0, lane, st, p);
        if (h1 != 0ull) w16_append<CAP_E>(a, 1, h1, th[1], row0, q0 + 16u, lane, st, p);
      }
    }
  };

  // lane's byte address of A-operand (block 0, k-step 0, row group 0) inside a slot
  const uint32_t lane_a = (uint32_t)(((g >> 1) * 64 + (g & 1) * 32 + c16) * 16);
  W16Acc acc0, acc1;   // acc1 outlives its phase: it is filtered inside the next one
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc0.t[rg][qg][j] = acc1.t[rg][qg][j] = -INFINITY;
  uint32_t row1_prev = p.n_rows;
  uint64_t t_wait = 0, t_bar = 0, t_c0 = 0, t_r0 = 0, t_flush = 0;
  float n_flush = 0.f;
  if (DBG & 4) {
    t_c0 = __builtin_amdgcn_s_memtime();
    t_r0 = __builtin_amdgcn_s_memrealtime();
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // one phase for the EARLY (waves 0-3) or LATE (waves 4-7) half of the workgroup
  auto phase = [&](uint32_t ph, auto late_tag) __attribute__((always_inline)) {
    constexpr bool LATE = decltype(late_tag)::value;
    constexpr int F1_AT = LATE ? 4 : 1;    // MFMA group in front of which the PREVIOUS phase's second block is filtered
    constexpr int F0_AT = LATE ? 10 : 7;   // ... and this phase's first block
    constexpr int NP = LATE ? NL : NE;     // my LDS-DMA pieces per phase, spread evenly over the 12 MFMA groups
    // my pieces of phase ph have landed (my NP pieces of phase ph+1 may stay in flight) ...
    uint64_t ts0 = 0, ts1 = 0;
    if (DBG & 4) ts0 = __builtin_amdgcn_s_memtime();
    if (DBG & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    if (DBG & 4) {
      ts1 = __builtin_amdgcn_s_memtime();
      t_wait += ts1 - ts0;
    }
    // ... and after the barrier everybody's have, and everybody has consumed phase ph-1
    __builtin_amdgcn_s_barrier();
    if (DBG & 4) t_bar += __builtin_amdgcn_s_memtime() - ts1;
    const Pieces nxt = pieces_of(ph + 2);  // goes into the slot phase ph - 1 has just vacated
    const u32x4* slot = slots + (ph % WL_SLOTS) * (WL_FRAGS * 64);
    const uint32_t sa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)slot + lane_a;
    const uint32_t b0 = (blockIdx.x + ph * G) * p.bstride * WL_PB;
    const uint32_t row0 = (b0 < nblk) ? b0 * 32u : p.n_rows;
    const uint32_t row1 = (b0 + 1u < nblk) ? (b0 + 1u) * 32u : p.n_rows;
    // A operands: groups of 4 reads, W16_AHEAD groups in flight ahead of the MFMAs that consume them
    rf_u32x4 fa[W16_AHEAD + 1][4];
    if (DBG & 64) {   // ablation: no A-operand reads at all (stale registers: wrong results)
#pragma unroll
      for (int i = 0; i < W16_AHEAD + 1; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "=v"(fa[i][j]));
    } else {
      w16_read_group<0>(fa[0], sa);
      if (W16_AHEAD > 1) w16_read_group<1>(fa[1], sa);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the one flush site of the loop (the append path never flushes)
    if (MODE == MODE_EMIT && st.cnt >= (uint32_t)CAP_E / 4) {   // (3/4 instead: no flush left in the loop, the sweep no faster: 165.4 against 165.0 us)
      uint64_t tf0 = 0;
      if (DBG & 4) tf0 = __builtin_amdgcn_s_memtime();
      w16_flush(st, p, lane);
      if (DBG & 4) {
        t_flush += __builtin_amdgcn_s_memtime() - tf0;
        n_flush += 1.f;
      }
    }
#define W16_GROUP(GI)                                                                                          \
    {                                                                                                          \
      constexpr int SLOT = GI % (W16_AHEAD + 1);                                                               \
      if (GI % 3 == 0 && !(DBG & 16)) __builtin_amdgcn_s_setprio(3 - GI / 3);   /* the wave that is behind wins */ \
      if (DBG & 64) {                                                                                          \
      } else if (GI + W16_AHEAD < 12) {                                                                        \
        w16_read_group<(GI + W16_AHEAD < 12 ? GI + W16_AHEAD : 0)>(fa[(GI + W16_AHEAD) % (W16_AHEAD + 1)], sa); \
        if (!(DBG & 32)) lds_wait_group<4 * W16_AHEAD>(fa[SLOT]);   /* DBG 32: no wait for the operands (wrong results) */ \
      } else {                                                                                                 \
        if (!(DBG & 32)) lds_wait_group<4 * (11 - GI)>(fa[SLOT]);                                              \
      }                                                                                                        \
      if (GI == F1_AT && !(DBG & 2)) filter(acc1, row1_prev);                                                  \
      if (GI == F0_AT && !(DBG & 2)) filter(acc0, row0);                                                       \
      if (!(DBG & 8)) {                                                                                        \
        _Pragma("unroll") for (int j = (GI * NP) / 12; j < ((GI + 1) * NP) / 12; ++j) issue_piece(nxt, j);     \
      }                                                                                                        \
      W16Acc& dst = (GI < 6) ? acc0 : acc1;                                                                    \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {         /* k-step 2 (GI % 6) + u */                      \
        const int ks = 2 * (GI % 6) + u;                                                                       \
        _Pragma("unroll") for (int rg = 0; rg < 2; ++rg) {                                                     \
          const half8 a = __builtin_bit_cast(half8, fa[SLOT][2 * u + rg]);                                     \
          _Pragma("unroll") for (int qg = 0; qg < 2; ++qg)                                                     \
            dst.t[rg][qg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(half8, qf[qg][ks]),   \
                                                                   ks ? dst.t[rg][qg] : zero4, 0, 0, 0);       \
        }                                                                                                      \
      }                                                                                                        \
    }
    W16_GROUP(0) W16_GROUP(1) W16_GROUP(2) W16_GROUP(3) W16_GROUP(4) W16_GROUP(5)
    W16_GROUP(6) W16_GROUP(7) W16_GROUP(8) W16_GROUP(9) W16_GROUP(10) W16_GROUP(11)
#undef W16_GROUP
    row1_prev = row1;
  };

  for (uint32_t ph = 0; ph < cnt; ++ph) {
    if (early) phase(ph, std::false_type{});
    else phase(ph, std::true_type{});
  }
  if (!(DBG & 2)) filter(acc1, row1_prev);   // the last phase's second block
  if (DBG & 2) asm volatile("" : "+v"(acc0.t[0][0]), "+v"(acc0.t[1][1]), "+v"(acc1.t[0][0]), "+v"(acc1.t[1][1]));

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dump pieces must land before the LDS is handed on
  if ((DBG & 4) && p.pmax && lane == 0) {   // diagnostic run only (tools/bench_wide.py --dbg 4)
    float* o = p.pmax + ((size_t)blockIdx.x * NW + wave) * 8;
    o[0] = (float)(__builtin_amdgcn_s_memtime() - t_c0);
    o[1] = (float)(__builtin_amdgcn_s_memrealtime() - t_r0);
    o[2] = (float)t_wait;   // cycles in the vmcnt wait (my LDS-DMA pieces of the phase not landed yet)
    o[3] = (float)cnt;
    o[4] = (float)t_bar;    // cycles in s_barrier (waiting for the slowest wave)
    o[5] = (float)t_flush;  // cycles in the staging flushes of the loop (global atomics)
    o[6] = n_flush;
  }
  if (MODE == MODE_EMIT) {
    if (st.cnt > 0) w16_flush(st, p, lane);
  } else {
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      const int qi = wave * 32 + qg * 16 + c16;
      float m = fmaxf(pm[qg], __shfl_xor(pm[qg], 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      if (g == 0 && qi < p.B) p.pmax[(size_t)qi * p.P + blockIdx.x] = m;
    }
  }
}

// =========================================================================================
// The sweep, round 3: FOUR waves of 64 queries, one wave per SIMD
// =========================================================================================
// k_scan_w16 above puts two waves on every SIMD (238 registers each).  Its stamps and ablations
//  say that this two-wave structure itself packs the matrix pipe to ~80 % at best:
// one barrier per 96 MFMAs for eight waves, every A operand read from LDS feeding two MFMAs, two
// in-order instruction streams arbitrating for one pipe.  Round 3's encoder kernel (encoder_post.hip)
// showed what ONE wave per SIMD with the whole register file does on this part when its stream is
// laid out for in-order issue -- LDS reads a few MFMAs ahead, LDS-DMA pieces from scalar addresses,
// nothing else in the loop: 83-88 % of the pipe in its MFMA-only phases.  Here:
//   * a wave keeps 64 queries (four 16-query groups x 12 k-steps = 192 registers, accumulator
//     half of the file) resident; every A operand read from LDS (16 rows x 32 k, 1 KiB) feeds FOUR
//     MFMAs: half the LDS traffic and half the read instructions per MFMA of the 8-wave form, and a
//     quarter of the workgroup barriers' participants;
//   * the accumulators live in the vector half (inline-asm MFMAs with vector-register C / D), so
//     the filter's v_max3 read them in place;
//   * the corpus ring is unchanged (3 slots x 2 blocks, two phases ahead, counted vmcnt + one raw
//     s_barrier per phase); its pieces are issued from wave-uniform scalar addresses (inline asm:
//     no address VALU, and the compiler-visible LDS / memory operations of the rare flush draw no
//     vmcnt(0));
//   * filter, append and flush are k_scan_w16's (same staging entries).
// (First form tried: 16x16x32 MFMAs as in k_scan_w16 -- correct, 206 us against 182: a 16-cycle MFMA holds the
// vector issue for 8 of its 16 cycles, which leaves a single in-order wave no room for its LDS reads, LDS-DMA
// pieces and filter: 164 us without the filters, 188 without the pieces.  The 32x32x16 form below has 24 free
// issue cycles per MFMA; its filter is cut into single operations that ride in the MFMA gaps, far enough behind
// the accumulator's last MFMA to need no hazard nops.)
#define W64_ENTRY_WORDS 20   // { row of score 0, query, threshold, pad, 16 scores }: a lane's whole 16-row column
template <class P>
__device__ __forceinline__ void w64_flush(W16Stage& st, const P& p, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  for (uint32_t i = lane; i < st.cnt * 16u; i += 64) {
    const uint32_t* e = st.base + (i >> 4) * W64_ENTRY_WORDS;
    const uint32_t s = i & 15u;
    const float score = __builtin_bit_cast(float, e[4 + s]);
    const uint32_t row = e[0] + (s & 3u) + 8u * (s >> 2);   // acc_row(s, h) with 4 h folded into e[0]
    if (score >= __builtin_bit_cast(float, e[2]) && row < p.n_rows) {
      const uint32_t list = e[1] * RF_CAND_SHARDS + (blockIdx.x & (RF_CAND_SHARDS - 1));
      const uint32_t slot = atomicAdd(&p.cand_cnt[list], 1u);
      if (slot < p.cap) p.cand[(size_t)list * p.cap + slot] = make_uint2(row, __builtin_bit_cast(uint32_t, score));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  st.cnt = 0;
}
template <int CAP_E, class P>
__device__ __forceinline__ void w64_append(const f32x16& a, unsigned long long mask, float th, uint32_t row0, uint32_t q,
                                           int lane, W16Stage& st, const P& p) {
  const uint32_t n = (uint32_t)__popcll(mask);
  const bool ok = (mask >> lane) & 1ull;
  if (st.cnt + n <= (uint32_t)CAP_E) {
    if (ok) {
      // inline-asm stores (nothing here may draw a compiler wait on the corpus ring)
      const uint32_t e = st.base_addr + (st.cnt + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))) * (W64_ENTRY_WORDS * 4);
      const u32x4 hdr = {row0 + 4u * (uint32_t)(lane >> 5), q, __builtin_bit_cast(uint32_t, th), 0u};
      const f32x4 s0 = {a[0], a[1], a[2], a[3]}, s1 = {a[4], a[5], a[6], a[7]}, s2 = {a[8], a[9], a[10], a[11]},
                  s3 = {a[12], a[13], a[14], a[15]};
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32\n\t"
                   "ds_write_b128 %0, %4 offset:48\n\tds_write_b128 %0, %5 offset:64"
                   :: "v"(e), "v"(hdr), "v"(s0), "v"(s1), "v"(s2), "v"(s3) : "memory");
    }
    st.cnt += n;
  } else if (ok) {   // staging full within one phase (adversarial duplicates): flag the query
    atomicAdd(&p.cand_cnt[q * RF_CAND_SHARDS + (blockIdx.x & (RF_CAND_SHARDS - 1))], p.cap + 1u);
  }
}

template <int MODE, int DBG>
__global__ void __launch_bounds__(256, 1) k_scan_w64(WideParams p) {
  constexpr int NW = 4;
  constexpr int NP = WL_FRAGS / NW;                               // LDS-DMA pieces per wave and phase (12)
  constexpr int CAP_E = WL_STAGE_WORDS / NW / W64_ENTRY_WORDS;    // emit staging entries per wave: 38
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* slots = (u32x4*)smem_raw;
  uint32_t* stage = (uint32_t*)(slots + WL_SLOTS * WL_FRAGS * 64);
  u32x4* const dump = (u32x4*)(stage + WL_STAGE_WORDS);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const uint32_t G = gridDim.x;
  const uint32_t cnt = (p.n_work > blockIdx.x) ? (p.n_work - blockIdx.x + G - 1) / G : 0u;
  if (cnt == 0) return;  // workgroup-uniform
  const uint32_t nblk = (p.n_rows + 31u) >> 5;

  // A phase is 48 fragments (two blocks, contiguous in HBM); wave w brings fragments 12 w .. 12 w + 11.  Past the
  // end of the stream the pieces re-read the corpus' last block into the dump area, and so does the second block
  // of an odd tail (its rows are masked by row1): the same 12 pieces per wave in every phase.
  const uint32_t slots_s = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)slots;
  const uint32_t dump_s = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)dump;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  struct Pieces {
    const char* src;    // fragment 0 of the phase's first block (wave-uniform)
    uint32_t dst;       // LDS byte address of fragment 0 of the slot, or of the dump area
    uint32_t dstep;     // 1024 | 0
    uint32_t cut;       // fragments >= cut come from 24 fragments further back
  };
  auto pieces_of = [&](uint32_t ph) __attribute__((always_inline)) {
    const bool live = ph < cnt;
    uint32_t b = (blockIdx.x + ph * G) * p.bstride * WL_PB;
    Pieces pc;
    pc.cut = (live && b + 1u < nblk) ? 2u * WIDE_KS : (uint32_t)WIDE_KS;
    b = (live && b < nblk) ? b : nblk - 1u;
    pc.src = (const char*)p.corpus + (size_t)b * (WIDE_KS * RF_FRAG_BYTES);
    pc.dst = live ? slots_s + (ph % WL_SLOTS) * (uint32_t)(WL_FRAGS * RF_FRAG_BYTES) : dump_s;
    pc.dstep = live ? (uint32_t)RF_FRAG_BYTES : 0u;
    return pc;
  };
  auto issue_piece = [&](const Pieces& pc, int j) __attribute__((always_inline)) {
    const uint32_t f = (uint32_t)(wave * NP + j);
    const uint32_t fs = (DBG & 1) ? 0u : (f >= pc.cut ? f - (uint32_t)WIDE_KS : f);   // DBG 1: every piece re-reads one cached KiB
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt"
                 :: "s"(pc.dst + f * pc.dstep), "v"(lane_off), "s"(pc.src + (size_t)fs * RF_FRAG_BYTES) : "memory");
  };
  {
    const Pieces p0 = pieces_of(0), p1 = pieces_of(1);
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(p0, j);
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(p1, j);
  }

  // this wave's 64 queries (two blocks of 32) as B operands of the 32x32x16 MFMA, resident for the whole sweep
  u32x4 qf[2][WIDE_KS];
  float th[2];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int qi = wave * 64 + jb * 32 + c;
    const int qc = qi < p.B ? qi : p.B - 1;   // unconditional loads (no branch per fragment)
#pragma unroll
    for (int kk = 0; kk < WIDE_KS; ++kk)
      qf[jb][kk] = *(const u32x4*)(p.q + (size_t)qc * (WIDE_KS * 16) + kk * 16 + h * 8);
    th[jb] = (MODE == MODE_EMIT) ? p.thr[qc] : 0.f;
    if (qi >= p.B || (DBG & 1)) th[jb] = INFINITY;
  }
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int qi = wave * 64 + jb * 32 + c;
#pragma unroll
    for (int kk = 0; kk < WIDE_KS; ++kk) {
      u32x4 v = qf[jb][kk];
      if (qi >= p.B) v = u32x4{0u, 0u, 0u, 0u};
      asm volatile("" : "+a"(v));   // resident in the accumulator half: the vector half stays free for accumulators and operands
      qf[jb][kk] = v;
    }
  }
  float pm[2] = {-INFINITY, -INFINITY};
  W16Stage st;
  st.cnt = 0;
  st.base = stage + wave * (WL_STAGE_WORDS / NW);
  st.base_addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)st.base;

  // [phase parity][block of the phase][query block]: lane = query c, register i = row acc_row(i, h) of the block.
  // FOUR accumulation chains rotate (the phase's two blocks advance together, k-step by k-step): a dependent MFMA
  // with vector-register C / D issued two slots behind its predecessor waits for the write-back (the encoder's QKV
  // phase, two such chains: 49 cycles per MFMA against 38) -- at distance four it does not.  Both blocks therefore
  // finish with the phase's last MFMAs, and their filter rides under the NEXT phase's MFMAs: two accumulator sets.
  f32x16 acc[2][2][2];
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a_][b][jb][r] = -INFINITY;
  // The filter of one 32-row block as 18 single operations (per query block: 8 v_max3 over the 16 rows, then the
  // compare against the query's threshold / the running sample maximum), one per fragment of the stream, so that
  // each rides in the shadow of an MFMA pair.  The block's accumulators were finished at least four MFMAs earlier.
  float fm[2];
  unsigned long long hit[2];
  auto filter_slot = [&](auto Tc, f32x16 (&a)[2]) __attribute__((always_inline)) {
    constexpr int T = decltype(Tc)::value, jb = T / 9, w = T % 9;
    if constexpr (w == 0) fm[jb] = vmax3(a[jb][0], a[jb][1], a[jb][1]);
    else if constexpr (w < 8) fm[jb] = vmax3(fm[jb], a[jb][2 * w], a[jb][2 * w + 1]);
    else if constexpr (MODE == MODE_SAMPLE) pm[jb] = vmax3(pm[jb], fm[jb], fm[jb]);
    else hit[jb] = __builtin_amdgcn_fcmpf(fm[jb], th[jb], 3 /* FCMP_OGE */);
  };
  auto mask_tail = [&](f32x16 (&a)[2], uint32_t row0) __attribute__((always_inline)) {
    if (MODE == MODE_SAMPLE && row0 + 32u > p.n_rows) {   // wave-uniform: the corpus' last block / a block past the end
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (row0 + acc_row(r, h) >= p.n_rows) a[jb][r] = -INFINITY;
    }
  };
  auto take_hits = [&](f32x16 (&a)[2], uint32_t row0) __attribute__((always_inline)) {
    if (MODE == MODE_EMIT && (hit[0] | hit[1]) != 0ull) {
      const uint32_t q0 = (uint32_t)(wave * 64 + c);
      if (hit[0] != 0ull) w64_append<CAP_E>(a[0], hit[0], th[0], row0, q0, lane, st, p);
      if (hit[1] != 0ull) w64_append<CAP_E>(a[1], hit[1], th[1], row0, q0 + 32u, lane, st, p);
    }
  };
  uint32_t row0_prev = p.n_rows, row1_prev = p.n_rows;
  uint64_t t_wait = 0, t_bar = 0, t_c0 = 0, t_r0 = 0;
  if (DBG & 4) {
    t_c0 = __builtin_amdgcn_s_memtime();
    t_r0 = __builtin_amdgcn_s_memrealtime();
  }

  auto phase = [&](uint32_t ph, auto parc) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value;
    uint64_t ts0 = 0;
    if (DBG & 4) ts0 = __builtin_amdgcn_s_memtime();
    // my pieces of phase ph have landed (my 12 pieces of phase ph + 1 may stay in flight) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    uint64_t ts1 = 0;
    if (DBG & 4) ts1 = __builtin_amdgcn_s_memtime();
    // ... and after the barrier everybody's have, and everybody has consumed phase ph - 1
    __builtin_amdgcn_s_barrier();
    if (DBG & 4) {
      t_wait += ts1 - ts0;
      t_bar += __builtin_amdgcn_s_memtime() - ts1;
    }
    const Pieces nxt = pieces_of(ph + 2);  // goes into the slot phase ph - 1 has just vacated
    const uint32_t sa = slots_s + (ph % WL_SLOTS) * (uint32_t)(WL_FRAGS * RF_FRAG_BYTES) + lane_off;
    const uint32_t b0 = (blockIdx.x + ph * G) * p.bstride * WL_PB;
    const uint32_t row0 = (b0 < nblk) ? b0 * 32u : p.n_rows;
    const uint32_t row1 = (b0 + 1u < nblk) ? (b0 + 1u) * 32u : p.n_rows;
    // the one flush site of the loop (the append path never flushes)
    if (MODE == MODE_EMIT && st.cnt >= (uint32_t)CAP_E / 4) w64_flush(st, p, lane);
    // item n of the phase: block n & 1, k-step n >> 1 (fragment 24 (n & 1) + (n >> 1) of the slot); two MFMAs per item
    run_step<WL_FRAGS>([](int n) constexpr { return (n & 1) * WIDE_KS + (n >> 1); }, sa,
                       [&](auto Nc, const rf_u32x4& af) __attribute__((always_inline)) {
      constexpr int n = decltype(Nc)::value, blk = n & 1, kk = n >> 1;
      if constexpr (kk == 0) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[PAR][blk][0]) : "v"(af), "a"(qf[0][kk]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[PAR][blk][1]) : "v"(af), "a"(qf[1][kk]));
      } else {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[PAR][blk][0]) : "v"(af), "a"(qf[0][kk]));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[PAR][blk][1]) : "v"(af), "a"(qf[1][kk]));
      }
      if constexpr (!(DBG & 2)) {
        // the previous phase's two blocks (the other accumulator set) are filtered under items 2..19 and 20..37
        if constexpr (n == 2) mask_tail(acc[PAR ^ 1][0], row0_prev);
        if constexpr (n >= 2 && n < 20) filter_slot(std::integral_constant<int, n - 2>{}, acc[PAR ^ 1][0]);
        if constexpr (n == 20) {
          take_hits(acc[PAR ^ 1][0], row0_prev);
          mask_tail(acc[PAR ^ 1][1], row1_prev);
        }
        if constexpr (n >= 20 && n < 38) filter_slot(std::integral_constant<int, n - 20>{}, acc[PAR ^ 1][1]);
        if constexpr (n == 38) take_hits(acc[PAR ^ 1][1], row1_prev);
      }
      if constexpr ((n & 3) == 3 && !(DBG & 8)) issue_piece(nxt, n >> 2);
      __builtin_amdgcn_sched_barrier(0);
    });
    row0_prev = row0;
    row1_prev = row1;
  };
  for (uint32_t ph = 0; ph < cnt; ++ph) {
    if (ph & 1u) phase(ph, std::integral_constant<int, 1>{});
    else phase(ph, std::integral_constant<int, 0>{});
  }
  if (!(DBG & 2)) {   // the last phase's blocks: their MFMAs have only just been issued -- pad the XDL write hazard by hand
    auto tail = [&](f32x16 (&a)[2][2]) __attribute__((always_inline)) {
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]));
      mask_tail(a[0], row0_prev);
      static_for<0, 18>([&](auto Tc) __attribute__((always_inline)) { filter_slot(Tc, a[0]); });
      take_hits(a[0], row0_prev);
      mask_tail(a[1], row1_prev);
      static_for<0, 18>([&](auto Tc) __attribute__((always_inline)) { filter_slot(Tc, a[1]); });
      take_hits(a[1], row1_prev);
    };
    if ((cnt - 1u) & 1u) tail(acc[1]);
    else tail(acc[0]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dump pieces must land before the LDS is handed on
  if ((DBG & 4) && p.pmax && lane == 0) {   // diagnostic run only (tools/bench_wide.py --dbg 4)
    float* o = p.pmax + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = (float)(__builtin_amdgcn_s_memtime() - t_c0);
    o[1] = (float)(__builtin_amdgcn_s_memrealtime() - t_r0);
    o[2] = (float)t_wait;   // cycles in the vmcnt wait
    o[3] = (float)cnt;
    o[4] = (float)t_bar;    // cycles in the barrier
  }
  if (MODE == MODE_EMIT) {
    if (st.cnt > 0) w64_flush(st, p, lane);
  } else {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      const int qi = wave * 64 + jb * 32 + c;
      const float m = fmaxf(pm[jb], __shfl_xor(pm[jb], 32));
      if (h == 0 && qi < p.B) p.pmax[(size_t)qi * p.P + blockIdx.x] = m;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------
static size_t wide_lds_bytes() {
  return (size_t)WL_SLOTS * WL_FRAGS * RF_FRAG_BYTES + (size_t)WL_STAGE_WORDS * 4 + (size_t)RF_FRAG_BYTES;   // slots, emit staging, dump area
}
#define W16_NE 12   // LDS-DMA pieces per phase of each of waves 0-3 (the product's choice; see k_scan_w16)
template <int MODE, int DBG, int NE = W16_NE>
static int launch_w16(const WideParams& p, int grid, hipStream_t st) {
  auto kern = k_scan_w16<MODE, DBG, NE>;
  static rf_lds_attr attr;   // per instantiation, per device
  RF_HIP(rf_ensure_lds(attr, (const void*)kern, wide_lds_bytes()));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), wide_lds_bytes(), st, p);
  RF_HIP(hipGetLastError());
  return RF_OK;
}
#ifdef RF_EXPERIMENTS
template <int MODE, int AUX, int NW, int ABL = 0, int PIN = 0>
static int launch_ldsdma(const WideParams& p, int grid, hipStream_t st) {
  auto kern = k_scan_ldsdma<MODE, AUX, NW, ABL, PIN>;
  static rf_lds_attr attr;
  RF_HIP(rf_ensure_lds(attr, (const void*)kern, wide_lds_bytes()));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), wide_lds_bytes(), st, p);
  RF_HIP(hipGetLastError());
  return RF_OK;
}
#endif
template <int MODE, int DBG>
static int launch_w64(const WideParams& p, int grid, hipStream_t st) {
  auto kern = k_scan_w64<MODE, DBG>;
  static rf_lds_attr attr;   // per instantiation, per device
  RF_HIP(rf_ensure_lds(attr, (const void*)kern, wide_lds_bytes()));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), wide_lds_bytes(), st, p);
  RF_HIP(hipGetLastError());
  return RF_OK;
}
template <int MODE>
static int dispatch_ldsdma(const WideParams& p, int grid, hipStream_t st) {
#ifdef RF_EXPERIMENTS
  if (rf_knob_wide_form == 1) {   // rf_set_tuning("wide_form", 1): four waves x 64 queries (round 3)
    if ((rf_knob_wide_dbg & 63) == 4) return launch_w64<MODE, 4>(p, grid, st);
    if ((rf_knob_wide_dbg & 63) == 2 && MODE == MODE_EMIT) return launch_w64<MODE, 2>(p, grid, st);
    if ((rf_knob_wide_dbg & 63) == 8 && MODE == MODE_EMIT) return launch_w64<MODE, 8>(p, grid, st);
    return launch_w64<MODE, 0>(p, grid, st);
  }
  // rf_set_tuning("wide_dbg", bits): 1 = every DMA piece re-reads one cached KiB, 2 = no filters, 4 = clock
  // stamps, 8 = no LDS-DMA in the loop (1 | 2 | 8: results wrong); 64 = the round-1 kernel (32x32x16 MFMA)
  if (rf_knob_wide_dbg & 64) return launch_ldsdma<MODE, 2, 8>(p, grid, st);
  if (MODE == MODE_EMIT) {
    switch (rf_knob_wide_dbg & 63) {
      case 20: return launch_w16<MODE_EMIT, 20>(p, grid, st);   // stamps, static priorities (no progress-based s_setprio)
      case 16: return launch_w16<MODE_EMIT, 16>(p, grid, st);   // static priorities
      case 33: return launch_w16<MODE_EMIT, 33>(p, grid, st);   // no operand waits, cached-KiB DMA, no hits
      case 37: return launch_w16<MODE_EMIT, 37>(p, grid, st);   // the same with stamps
      case 62: return launch_w16<MODE_EMIT, 65>(p, grid, st);   // (wide_dbg 62) no operand reads at all, cached-KiB DMA, no hits
      case 63: return launch_w16<MODE_EMIT, 69>(p, grid, st);   // (wide_dbg 63) the same with stamps
      case 61: return launch_w16<MODE_EMIT, 77>(p, grid, st);   // (wide_dbg 61) no reads, no DMA, stamps: MFMAs + filters + barrier only
#define W16_DBG(x) case x: return launch_w16<MODE_EMIT, x>(p, grid, st);
      W16_DBG(1) W16_DBG(4) W16_DBG(8) W16_DBG(5)
#undef W16_DBG
      default: break;
    }
    switch (rf_knob_wide_ne) {   // rf_set_tuning("wide_ne", n): the DMA split between the wave halves; + 100: with clock stamps
#define W16_NEV(n) case n: return launch_w16<MODE_EMIT, 0, n>(p, grid, st); case 100 + n: return launch_w16<MODE_EMIT, 4, n>(p, grid, st);
      W16_NEV(6) W16_NEV(8) W16_NEV(9) W16_NEV(10) W16_NEV(11)
#undef W16_NEV
      default: break;
    }
  }
#else
  if (rf_knob_wide_form == 1) return launch_w64<MODE, 0>(p, grid, st);
#endif
  return launch_w16<MODE, 0>(p, grid, st);
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
  p.dbg = (uint32_t)(rf_knob_wide_dbg & 7);   // read by the round-1 kernel only (experiments build); k_scan_w16 takes DBG as a template
  if (rf_knob_wide_dbg & 4) p.pmax = ws.pmax;   // stamp buffer of the diagnostic run
  return dispatch_ldsdma<MODE_EMIT>(p, grid, st);
}
