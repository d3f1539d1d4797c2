// Single-launch scan: threshold estimation and candidate emission in ONE sweep.
//
// The two-kernel form (scan.hip: MODE_SAMPLE -> k_threshold -> MODE_EMIT) reads a
// 48 MB sample a second time and pays two extra kernel boundaries (~20 us of a
// ~150 us step).  Here every wave scores its FIRST block, keeps the 32x32
// accumulators in registers and contributes to a per-query lower bound of the
// final k-th best score; a grid-wide hand-off publishes the bound; the wave
// then filters the block it is holding and streams on.  Nothing is re-read.
//
//   bound:   workgroup g's maximum over its 4 x 32 first rows goes, per query, into
//            group (g mod 64) with a device-scope atomic max.  Group maxima belong to
//            distinct rows, so the k-th largest of the 64 group maxima is <= the
//            k-th best score of the whole corpus.
//   hand-off: there is NO barrier and no signal word.  Any subset of the group
//            maxima is a valid bound (each is the score of a real row), so a consumer
//            reads whatever has arrived (agent-scope atomic max on the producer side,
//            agent-scope atomic loads on the consumer side -- "atomics both sides",
//            cdna_hip_programming.md G16) and only needs k non-empty groups; it re-reads,
//            bounded, until it has them.  Candidate SETS may differ from run to run,
//            the merged result cannot: every set contains the exact top-k.  Under a
//            saturating stream each dependent global access costs several us of queueing,
//            which is why the counter/poll form (4 dependent hops) was dropped.
//   latency: while wave 0 of a workgroup derives the 64 thresholds (~5 us), the register
//            rings of all four waves already hold the requests for their second
//            blocks, so HBM keeps streaming.  (Holding a second accumulator set to
//            run block 2's MFMAs meanwhile was tried: the extra 32 VGPRs spill in the
//            prologue, and every scratch reload drains the 24 ring loads in front of
//            it -- 25 us slower.)
//   safety:  the re-read is bounded.  If too few workgroups are resident to fill k
//            groups the wave gives up, uses thr = -inf (every row a candidate), the candidate
//            lists overflow, rf_search flags the queries and the caller's
//            exhaustive path answers them: slow, never wrong, never hung.
#include "scan_common.h"

struct FusedParams {
  const uint4* corpus;
  const _Float16* q;
  int B;
  int k;
  uint32_t n_rows;
  uint32_t n_blocks;
  int use_sample;              // 0: small corpus, thr = -inf without any hand-off
  uint32_t* gmax;              // [64][RF_MAX_K] ordered-uint group maxima (zero = empty)
  uint32_t* bar;               // [0] arrivals  [1] give-up marker
  const uint32_t* max_norm2;
  float* eps_out;              // [64]
  uint32_t* cand_cnt;          // [64][RF_CAND_SHARDS]
  uint2* cand;                 // [64][RF_CAND_SHARDS][cap]
  uint32_t cap;
};

#define FUSED_POLL_LIMIT (1u << 18)

template <int KS, int JB, bool LAST>
__device__ __forceinline__ void mfma_block(u32x4 (&ring)[RingOf<KS>::R], const uint4* cur,
                                           const uint4* nxt, const u32x4* smemQ, int lane,
                                           f32x16 (&acc)[JB]) {
  constexpr int R = RingOf<KS>::R;
  // keep the query-fragment LDS reads inside the block (see scan.hip)
  asm volatile("" ::: "memory");
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
    if (kk + R < KS) {
      ring[kk % R] = ld_frag(cur + (kk + R) * 64);
    } else if (!LAST) {
      ring[kk % R] = ld_frag(nxt + (kk + R - KS) * 64);
    }
  }
}

// k-th largest (with multiplicity) of NG group maxima (RF_MAX_K stored, folded
// pairwise when NG == 32); 0 when fewer than k groups are non-empty.
template <int NG>
__device__ __forceinline__ uint32_t kth_group_max(const uint32_t* g_ptr, int k) {
  uint32_t v[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    v[g] = __hip_atomic_load(g_ptr + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (NG < RF_MAX_K) {
      const uint32_t o = __hip_atomic_load(g_ptr + g + NG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v[g] = o > v[g] ? o : v[g];
    }
  }
  uint32_t kth = 0u;
  int remaining = k;
  while (remaining > 0) {  // destructive: retire the current maximum each round
    uint32_t m = 0u;
#pragma unroll
    for (int g = 0; g < NG; ++g) m = v[g] > m ? v[g] : m;
    kth = m;
    if (m == 0u) break;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (v[g] == m) {
        v[g] = 0u;
        --remaining;
      }
  }
  return kth;
}

template <int JB>
__device__ __forceinline__ void filter_block(const f32x16 (&acc)[JB], const float (&th)[JB],
                                             uint32_t row0, int lane, EmitState& es,
                                             const FusedParams& p) {
  bool hit = false;
#pragma unroll
  for (int jb = 0; jb < JB; ++jb) hit |= (max16(acc[jb]) >= th[jb]);
  if (__ballot(hit) != 0ull) emit_slow<JB>(acc, th, row0, lane, es, p);
}

template <int KS, int JB, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 2) k_scan_fused(FusedParams p) {
  constexpr int R = RingOf<KS>::R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* smemQ = (u32x4*)smem_raw;                                   // JB*KS*64 uint4
  unsigned char* tail = smem_raw + (size_t)JB * KS * RF_FRAG_BYTES;
  uint32_t* stage = (uint32_t*)tail;                                 // 3 * WAVES * SCAP words
  float* red = (float*)(stage + 3 * WAVES * SCAP);                   // [WAVES*2][64]
  float* s_thr = red + WAVES * 2 * 64;                               // [64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dim = KS * 16;

  for (int idx = tid; idx < JB * KS * 64; idx += WAVES * 64) {
    const int l = idx & 63;
    const int kk = (idx >> 6) % KS;
    const int jb = idx / (64 * KS);
    const int qi = jb * 32 + (l & 31);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (qi < p.B) v = *(const u32x4*)(p.q + (size_t)qi * dim + kk * 16 + (l >> 5) * 8);
    smemQ[idx] = v;
  }
  EmitState es;
  es.cnt = 0;
  es.q_base = 0;
  es.s_row = stage + wave * SCAP;
  es.s_score = (float*)(stage + WAVES * SCAP) + wave * SCAP;
  es.s_q = stage + 2 * WAVES * SCAP + wave * SCAP;
  __syncthreads();

  const uint32_t W = gridDim.x * WAVES;
  const uint32_t gw = blockIdx.x * WAVES + wave;
  const uint32_t cnt = (p.n_blocks > gw) ? (p.n_blocks - gw + W - 1) / W : 0u;
  const size_t bstep = (size_t)KS * 64;
  auto blk = [&](uint32_t b) { return p.corpus + (size_t)b * bstep + lane; };

  u32x4 ring[R];
  if (cnt > 0) {
    const uint4* src = blk(gw);
#pragma unroll
    for (int s = 0; s < R; ++s) ring[s] = ld_frag(src + s * 64);
  }

  f32x16 acc1[JB];
  // ---- first block: MFMAs only, accumulators stay in registers ---------------------
  if (cnt == 1) mfma_block<KS, JB, true>(ring, blk(gw), blk(gw), smemQ, lane, acc1);
  else if (cnt >= 2) mfma_block<KS, JB, false>(ring, blk(gw), blk(gw + W), smemQ, lane, acc1);

  if (p.use_sample) {
    const int h = lane >> 5;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
      float m = -INFINITY;
      if (cnt >= 1) {
        const uint32_t row0 = gw * 32u;
        if (row0 + 32u > p.n_rows) {  // the corpus' ragged last block: ignore pad rows
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (row0 + acc_row(i, h) < p.n_rows) m = fmaxf(m, acc1[jb][i]);
        } else {
          m = max16(acc1[jb]);
        }
      }
      red[(wave * 2 + h) * 64 + jb * 32 + (lane & 31)] = m;
    }
    __syncthreads();
    if (wave == 0) {
      if (lane < JB * 32 && lane < p.B) {
        float m = -INFINITY;
        for (int s = 0; s < WAVES * 2; ++s) m = fmaxf(m, red[s * 64 + lane]);
        if (m > -INFINITY)
          __hip_atomic_fetch_max(&p.gmax[lane * RF_MAX_K + (int)(blockIdx.x % RF_MAX_K)],
                                 rf_f2ord(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // fire and forget: no signal, no drain (see "hand-off" in the header)
    }
  }

  // ---- thresholds: wave 0, lane = query ----------------------------------------------
  if (wave == 0) {
    // ||q||^2 from the LDS fragment image: query `lane` = (jb, j), both lane halves of a fragment
    float nq2 = 0.f;
    if (lane < JB * 32) {
      const int jb = lane >> 5, j = lane & 31;
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const half8 v = __builtin_bit_cast(half8, smemQ[(jb * KS + kk) * 64 + j + 32 * hh]);
#pragma unroll
          for (int e = 0; e < 8; ++e) nq2 = fmaf((float)v[e], (float)v[e], nq2);
        }
    }
    const float cmax2 = __builtin_bit_cast(float, *p.max_norm2);
    const float eps = 1.25f * (float)dim * 1.1920929e-7f * sqrtf(nq2) * sqrtf(cmax2);
    float t = (lane >= p.B) ? INFINITY : -INFINITY;   // padding query: never a candidate
    if (p.use_sample && lane < p.B) {
      // k-th largest (with multiplicity) of this query's group maxima: 32 groups
      // (pairs folded) when k <= 16, all 64 otherwise.  Whatever subset of workgroups
      // has contributed so far gives a valid bound; it only has to cover k groups, so
      // re-read (bounded) until it does.
      uint32_t kth = 0u;
      for (uint32_t attempt = 0; attempt < FUSED_POLL_LIMIT && kth == 0u; ++attempt) {
        if (attempt) __builtin_amdgcn_s_sleep(32);
        kth = p.k <= 16 ? kth_group_max<32>(p.gmax + lane * RF_MAX_K, p.k)
                        : kth_group_max<RF_MAX_K>(p.gmax + lane * RF_MAX_K, p.k);
      }
      if (kth != 0u) t = rf_ord2f(kth) - 2.f * eps;
      else if (lane == 0) __hip_atomic_store(&p.bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_thr[lane] = t;
    if (blockIdx.x == 0) p.eps_out[lane] = eps;
  }
  __syncthreads();
  float th[JB];
#pragma unroll
  for (int jb = 0; jb < JB; ++jb) th[jb] = s_thr[jb * 32 + (lane & 31)];

  // ---- filter the held block, then stream the rest ---------------------------------------
  if (cnt >= 1) filter_block<JB>(acc1, th, gw * 32u, lane, es, p);
  if (cnt >= 2) {
    uint32_t w = gw + W;
    for (uint32_t i = 1; i + 1 < cnt; ++i, w += W) {
      mfma_block<KS, JB, false>(ring, blk(w), blk(w + W), smemQ, lane, acc1);
      filter_block<JB>(acc1, th, w * 32u, lane, es, p);
    }
    mfma_block<KS, JB, true>(ring, blk(w), blk(w), smemQ, lane, acc1);
    filter_block<JB>(acc1, th, w * 32u, lane, es, p);
  }
  if (es.cnt > 0) emit_flush(es, p, lane);
}

// ---- host side --------------------------------------------------------------------------
template <int KS, int JB, int WAVES>
static int launch_fused(const FusedParams& p, int grid, hipStream_t st) {
  const size_t lds = (size_t)JB * KS * RF_FRAG_BYTES + (size_t)3 * WAVES * SCAP * 4 +
                     (size_t)WAVES * 2 * 64 * 4 + 64 * 4;
  auto kern = k_scan_fused<KS, JB, WAVES>;
  static bool attr_done = false;
  if (!attr_done) {
    RF_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, p);
  RF_HIP(hipGetLastError());
  return RF_OK;
}

int rf_launch_fused(const rf_index* ix, const void* q, int B, int JB, int k, const rf_workspace& ws,
                    hipStream_t st) {
  const int KS = ix->KS;
  const uint32_t nblk = (uint32_t)((ix->size + 31) / 32);
  const int WAVES = KS >= 48 ? 8 : 4;
  const int per_cu = KS >= 48 ? 1 : 2;
  // every workgroup must be resident for the hand-off: <= per_cu workgroups per CU
  int grid = ix->num_cus * per_cu;
  const uint32_t need = (nblk + WAVES - 1) / WAVES;
  if ((uint32_t)grid > need) grid = (int)need;
  if (grid < 1) grid = 1;
  FusedParams p{};
  p.corpus = ix->tiles;
  p.q = (const _Float16*)q;
  p.B = B;
  p.k = k;
  p.n_rows = (uint32_t)ix->size;
  p.n_blocks = nblk;
  p.use_sample = ix->size > RF_SMALL_ROWS ? 1 : 0;
  p.gmax = ws.gmax;
  p.bar = ws.bar;
  p.max_norm2 = ix->max_norm2;
  p.eps_out = ws.eps;
  p.cand_cnt = ws.cand_cnt;
  p.cand = ws.cand;
  p.cap = RF_SHARD_CAP;
#define RF_CASE(ks, waves)                                                 \
  case ks:                                                                 \
    return JB == 1 ? launch_fused<ks, 1, waves>(p, grid, st)               \
                   : launch_fused<ks, 2, waves>(p, grid, st);
  switch (KS) {
    RF_CASE(4, 4)
    RF_CASE(8, 4)
    RF_CASE(16, 4)
    RF_CASE(24, 4)
    RF_CASE(32, 4)
    RF_CASE(48, 8)
    RF_CASE(64, 8)
    default:
      break;
  }
#undef RF_CASE
  rf_set_error("no fused scan kernel for dim %d", ix->dim);
  return RF_ERR_UNSUPPORTED;
}
