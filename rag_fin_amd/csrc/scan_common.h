// Device helpers shared by scan.hip (sample / emit kernels) and scan_fused.hip.
#pragma once
#include "rf_internal.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum { MODE_SAMPLE = 0, MODE_EMIT = 1 };

#define SCAP 64  // per-wave LDS staging entries (>= 64: one ballot round can add 64)

struct ScanParams {
  const uint4* corpus;   // tiled
  const _Float16* q;     // row-major [B, dim]
  int B;
  uint32_t n_rows;
  uint32_t n_work;       // work items (blocks) for this launch
  uint32_t bstride;      // corpus block index = work index * bstride
  const float* thr;      // [64]
  uint32_t* cand_cnt;    // [64][RF_CAND_SHARDS]
  uint2* cand;           // [64][RF_CAND_SHARDS][cap]
  uint32_t cap;
  float* pmax;           // [64][P]
  int P;
};

#ifndef RF_RING24
#define RF_RING24 24
#endif
template <int KS>
struct RingOf {
  static constexpr int R = (KS == 24) ? RF_RING24 : ((KS < 24) ? KS : 16);  // must divide KS
};

__device__ __forceinline__ u32x4 ld_frag(const uint4* p) {
  return __builtin_nontemporal_load((const u32x4*)p);
}

__device__ __forceinline__ float max16(const f32x16& a) {
  float m0 = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
  float m1 = fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7]));
  float m2 = fmaxf(fmaxf(a[8], a[9]), fmaxf(a[10], a[11]));
  float m3 = fmaxf(fmaxf(a[12], a[13]), fmaxf(a[14], a[15]));
  return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

// row of accumulator register i within the 32-row block (lane half h)
__device__ __forceinline__ uint32_t acc_row(int i, int h) {
  return (uint32_t)((i & 3) + 8 * (i >> 2) + 4 * h);
}

struct EmitState {
  uint32_t* s_row;     // [SCAP] per wave (LDS)
  float* s_score;      // [SCAP]
  uint32_t* s_q;       // [SCAP]
  uint32_t cnt;        // wave-uniform
  uint32_t q_base;     // query index of this wave's column 0 (wide sweep: 32 * wave)
};

template <class P>
__device__ __forceinline__ void emit_flush(EmitState& es, const P& p, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  for (uint32_t i = lane; i < es.cnt; i += 64) {
    const uint32_t q = es.s_q[i];
    // RF_CAND_SHARDS counters per query: same-address atomics serialise (~12 ns
    // each), and most waves flush together at the end of the scan
    const uint32_t list = q * RF_CAND_SHARDS + (blockIdx.x & (RF_CAND_SHARDS - 1));
    const uint32_t slot = atomicAdd(&p.cand_cnt[list], 1u);
    if (slot < p.cap)
      p.cand[(size_t)list * p.cap + slot] =
          make_uint2(es.s_row[i], __builtin_bit_cast(uint32_t, es.s_score[i]));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  es.cnt = 0;
}

// Slow path of the filter (some lane holds a score >= its query's threshold).
// Branch-free build of a per-lane 32-bit hit mask (bit jb*16+i), then a wave loop
// that retires each lane's lowest set bit per iteration: the usual case (one or two
// hits in the whole wave) costs one iteration instead of 32 ballot+branch rounds.
template <int JB, class P>
__device__ __forceinline__ void emit_slow(const f32x16 (&acc)[JB], const float (&th)[JB],
                                          uint32_t row0, int lane, EmitState& es, const P& p) {
  const int h = lane >> 5;
  const uint32_t lim = p.n_rows - row0;  // rows of this block that exist (>= 32 except the last block)
  uint32_t bits = 0u;
#pragma unroll
  for (int jb = 0; jb < JB; ++jb)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      bits |= ((acc[jb][i] >= th[jb]) && (acc_row(i, h) < lim)) ? (1u << (jb * 16 + i)) : 0u;
  unsigned long long mask;
  while ((mask = __ballot(bits != 0u)) != 0ull) {
    const bool pass = bits != 0u;
    const int b = __ffs((int)bits) - 1;  // -1 when !pass (unused then)
    float s = 0.f;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int i = 0; i < 16; ++i) s = (b == jb * 16 + i) ? acc[jb][i] : s;
    const uint32_t n = (uint32_t)__popcll(mask);
    if (es.cnt + n > SCAP) emit_flush(es, p, lane);
    if (pass) {
      const int i = b & 15;
      const uint32_t slot = es.cnt + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      es.s_row[slot] = row0 + (uint32_t)((i & 3) + 8 * (i >> 2) + 4 * h);
      es.s_score[slot] = s;
      es.s_q[slot] = es.q_base + (uint32_t)((b >> 4) * 32 + (lane & 31));
    }
    es.cnt += n;
    bits &= bits - 1u;
  }
}

// Slow path, second form (wide sweep): one ballot per accumulator register instead of a
// per-lane bit mask and a 32-way select chain.  A hit is rare (~0.3 per wave and block at the
// default sample size) and almost always a single (query, row) pair, so the cost that matters
// is the scan for it: per register one compare whose SGPR-pair result IS the ballot, one scalar
// test, and the append only behind a taken branch.  `jb_hit` (wave-uniform) says which of the
// query blocks needs scanning at all.
// The staging area holds CAP entries and is flushed by the CALLER at one place per phase (an
// inlined flush per append, or even per query block, multiplies the code and made hipcc spill
// in the MFMA loop).  An append that does not fit -- more hits within one phase than the room
// left at its start: duplicate-heavy or otherwise adversarial data -- is not stored; instead its query's candidate counter is pushed past the
// list capacity, which the merge reports as RF_FLAG_CAND_OVERFLOW, and the caller answers that
// query through the exhaustive path.
template <int JB, int CAP, class P>
__device__ __forceinline__ void emit_scan(const f32x16 (&acc)[JB], const float (&th)[JB],
                                          const unsigned long long (&jb_hit)[JB], uint32_t row0,
                                          int lane, EmitState& es, const P& p) {
  const int h = lane >> 5;
  const uint32_t lim = p.n_rows - row0;  // rows of this block that exist (0 for a block past the end)
#pragma unroll
  for (int jb = 0; jb < JB; ++jb) {
    if (jb_hit[jb] == 0ull) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bool ok = (acc[jb][i] >= th[jb]) && (acc_row(i, h) < lim);
      const unsigned long long mask = __ballot(ok);
      if (mask != 0ull) {
        const uint32_t n = (uint32_t)__popcll(mask);
        const uint32_t q = es.q_base + (uint32_t)(jb * 32 + (lane & 31));
        if (es.cnt + n <= (uint32_t)CAP) {
          if (ok) {
            const uint32_t slot = es.cnt + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            es.s_row[slot] = row0 + acc_row(i, h);
            es.s_score[slot] = acc[jb][i];
            es.s_q[slot] = q;
          }
          es.cnt += n;
        } else if (ok) {
          atomicAdd(&p.cand_cnt[q * RF_CAND_SHARDS + (blockIdx.x & (RF_CAND_SHARDS - 1))], p.cap + 1u);
        }
      }
    }
  }
}
