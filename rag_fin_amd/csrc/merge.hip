// Threshold selection, candidate merge + exact rescoring, exhaustive fallback
// and cross-shard merge.  Together with scan.hip this is the GPU restatement of
// what Milvus returns for Collection.search(..., COSINE, top_k)
// (vector_rag_mcp/main.py:51-70): the k best rows by descending score.
//
// Ranking contract (mirrored by oracle/search.py): the score of (query, row)
// is the fp64 value p = ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)) where chain
// p_j = sum over d = j, j+8, j+16, ... (ascending) of q[d]*c[d], one fused
// multiply-add per step (products of two fp16 values are exact in fp64); rows
// are ranked by (score descending, row id ascending).
// The MFMA scan only nominates candidates; every returned score comes from the
// fp64 chain, so ids and ranks are bit-reproducible on the CPU.
#include "rf_internal.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Wave-wide reductions on the DPP network (row_shr 1/2/4/8, row_bcast 15/31):
// six VALU-rate steps instead of six LDS-crossbar shuffles.  The full result
// lands in lane 63 and is broadcast with readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_keep(uint32_t v) {
  // lanes without a valid source (or in a masked row) keep their own value
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ float wave_max_f(float v) {
#define RF_STEP(ctrl, rm) \
  v = fmaxf(v, __builtin_bit_cast(float, dpp_keep<ctrl, rm>(__builtin_bit_cast(uint32_t, v))));
  RF_STEP(0x111, 0xF) RF_STEP(0x112, 0xF) RF_STEP(0x114, 0xF) RF_STEP(0x118, 0xF)
  RF_STEP(0x142, 0xA) RF_STEP(0x143, 0xC)
#undef RF_STEP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#define RF_STEP(ctrl, rm)                                                           \
  {                                                                                 \
    const uint32_t lo = dpp_keep<ctrl, rm>((uint32_t)v);                            \
    const uint32_t hi = dpp_keep<ctrl, rm>((uint32_t)(v >> 32));                    \
    const unsigned long long w = ((unsigned long long)hi << 32) | lo;               \
    v = w > v ? w : v;                                                              \
  }
  RF_STEP(0x111, 0xF) RF_STEP(0x112, 0xF) RF_STEP(0x114, 0xF) RF_STEP(0x118, 0xF)
  RF_STEP(0x142, 0xA) RF_STEP(0x143, 0xC)
#undef RF_STEP
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// fp64 ranking score of the contract: eight interleaved chains (chain j takes
// dims d = j mod 8, ascending, one fma per step) combined by a pairwise tree.
__device__ __forceinline__ double tree8(const double (&p)[8]) {
  return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}
__device__ __forceinline__ double exact_dot(const _Float16* __restrict__ qrow,
                                            const uint4* __restrict__ tiles, int64_t row, int KS) {
  double p[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = 0.0;
  for (int c = 0; c < 2 * KS; ++c) {
    const uint4 cv = tiles[rf_chunk_index(row, c, KS)];
    const uint4 qv = *(const uint4*)(qrow + 8 * c);
    const half8 ch = __builtin_bit_cast(half8, cv);
    const half8 qh = __builtin_bit_cast(half8, qv);
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = fma((double)qh[j], (double)ch[j], p[j]);
  }
  return tree8(p);
}

// (score desc, row asc): is (s1, r1) ranked strictly before (s2, r2)?
__device__ __forceinline__ bool ranks_before(double s1, int64_t r1, double s2, int64_t r2) {
  return (s1 > s2) || (s1 == s2 && r1 < r2);
}

// ---- emit threshold -------------------------------------------------------------
// One wave per query.  eps bounds |MFMA fp32 score - exact score|:
// dim * 2^-23 * ||q|| * max||c||  (fp16 products are exact in fp32; the bound
// covers dim roundings of the fp32 accumulator at one ulp each).  The scan
// keeps every row whose MFMA score is >= kth - 2*eps, which provably contains
// every row of the exact top-k ("why the result is exact").
__global__ void __launch_bounds__(64) k_threshold(const _Float16* __restrict__ q, int B, int dim,
                                                  int k, const float* __restrict__ pmax, int P,
                                                  const uint32_t* __restrict__ max_norm2,
                                                  float* __restrict__ thr, float* __restrict__ eps_out,
                                                  uint32_t* __restrict__ cand_cnt) {
  const int qi = blockIdx.x;
  const int lane = threadIdx.x;
  float s = 0.f;
  if (qi < B)
    for (int d = lane; d < dim; d += 64) {
      const float x = (float)q[(size_t)qi * dim + d];
      s = fmaf(x, x, s);
    }
  s = wave_sum_f(s);
  const float cmax2 = __builtin_bit_cast(float, *max_norm2);
  const float eps = 1.25f * (float)dim * 1.1920929e-7f * sqrtf(s) * sqrtf(cmax2);

  float t;
  if (qi >= B) {
    t = INFINITY;
  } else if (P <= 0) {
    t = -INFINITY;
  } else {
    float v[RF_SAMPLE_WGS / 64];
#pragma unroll
    for (int i = 0; i < RF_SAMPLE_WGS / 64; ++i) {
      const int j = lane + 64 * i;
      v[i] = (j < P) ? pmax[(size_t)qi * P + j] : -INFINITY;
    }
    float kth = -INFINITY;
    for (int r = 0; r < k; ++r) {
      float m = v[0];
#pragma unroll
      for (int i = 1; i < RF_SAMPLE_WGS / 64; ++i) m = fmaxf(m, v[i]);
      const float wm = wave_max_f(m);
      kth = wm;
      if (wm == -INFINITY) break;
      const unsigned long long who = __ballot(m == wm);
      const int winner = __ffsll((long long)who) - 1;
      if (lane == winner) {
        bool done = false;
#pragma unroll
        for (int i = 0; i < RF_SAMPLE_WGS / 64; ++i)
          if (!done && v[i] == wm) {
            v[i] = -INFINITY;
            done = true;
          }
      }
    }
    t = (kth == -INFINITY) ? -INFINITY : kth - 2.f * eps;
  }
  if (lane == 0) {
    thr[qi] = t;
    eps_out[qi] = eps;
  }
  if (lane < RF_CAND_SHARDS) cand_cnt[qi * RF_CAND_SHARDS + lane] = 0u;
}

// ---- candidate merge + exact rescoring --------------------------------------------
// One workgroup per query:
//   1. find the k-th largest candidate by (MFMA score, row): rank counting over
//      LDS-broadcast keys when there are <= 1024 candidates (the usual ~150-300),
//      extraction rounds otherwise (small corpora, where every row is a candidate);
//   2. compact the rescoring set R = {MFMA score >= kth - 2 eps} (k + a few rows);
//   3. stage the rows of R in LDS with all 256 threads (one HBM latency instead
//      of one per 16-byte chunk), then one thread per row runs the fp64 chain;
//   4. rank R by (exact desc, row asc) and write the top-k.
#define MERGE_THREADS 256
#define MERGE_PER_THREAD (RF_CAND_CAP / MERGE_THREADS)
#define MERGE_RANK_MAX 1024
#define MERGE_STAGE_ROWS 32

__device__ __forceinline__ unsigned long long cand_key(uint2 e) {
  // larger key <=> (higher score, then lower row)
  return ((unsigned long long)rf_f2ord(__builtin_bit_cast(float, e.y)) << 32) |
         (unsigned long long)(0xFFFFFFFFu - e.x);
}
__device__ __forceinline__ float key_score(unsigned long long key) {
  return rf_ord2f((uint32_t)(key >> 32));
}
__device__ __forceinline__ uint32_t key_row(unsigned long long key) {
  return 0xFFFFFFFFu - (uint32_t)key;
}

static size_t merge_lds_bytes(int dim) {
  return (size_t)MERGE_RANK_MAX * 8 + (MERGE_THREADS / 64) * RF_MAX_K * 8 + RF_RESCORE_CAP * 8 +
         RF_RESCORE_CAP * 4 + 64 + (size_t)dim * 8 + (size_t)MERGE_STAGE_ROWS * (dim * 2 + 16);
}

__global__ void __launch_bounds__(MERGE_THREADS) k_merge(
    const _Float16* __restrict__ q, int dim, int KS, const uint4* __restrict__ tiles, int k,
    int64_t id_base, const uint32_t* __restrict__ cand_cnt, const uint2* __restrict__ cand,
    uint32_t cap, const float* __restrict__ eps_in, float* __restrict__ scores,
    int64_t* __restrict__ ids, double* __restrict__ exact, uint32_t* __restrict__ flags,
    uint32_t* __restrict__ cand_cnt_rw, uint32_t n_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* skeys = (unsigned long long*)lds;                       // [1024]
  unsigned long long* wtop = skeys + MERGE_RANK_MAX;                          // [4][RF_MAX_K]
  double* r_exact = (double*)(wtop + (MERGE_THREADS / 64) * RF_MAX_K);        // [RESCORE_CAP]
  uint32_t* r_row = (uint32_t*)(r_exact + RF_RESCORE_CAP);                    // [RESCORE_CAP]
  uint32_t* r_cnt = r_row + RF_RESCORE_CAP;                                   // misc: 16 words
  float* t_cut = (float*)(r_cnt + 1);
  double* qd = (double*)(r_cnt + 16);                                         // [dim]
  uint4* srows = (uint4*)(qd + dim);                                          // [32][2 KS + 1]
  const int srow_stride = 2 * KS + 1;

  const int qi = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  bool bad_row = false;
  // candidate lists of this query: RF_CAND_SHARDS lists of up to `cap` entries;
  // global candidate index g -> (list s, entry g - off[s])
  uint32_t off[RF_CAND_SHARDS + 1];
  uint32_t fl = 0u;
  off[0] = 0u;
#pragma unroll
  for (int s = 0; s < RF_CAND_SHARDS; ++s) {
    const uint32_t n = cand_cnt[qi * RF_CAND_SHARDS + s];
    if (n > cap) fl = RF_FLAG_CAND_OVERFLOW;
    off[s + 1] = off[s] + (n < cap ? n : cap);
  }
  const uint32_t total = off[RF_CAND_SHARDS];
  if (total > RF_CAND_CAP) fl = RF_FLAG_CAND_OVERFLOW;
  const uint32_t c = total < RF_CAND_CAP ? total : RF_CAND_CAP;
  const int kk = (uint32_t)k < c ? k : (int)c;
  const bool have_cut = kk > 0 && (uint32_t)k <= c;  // fewer than k candidates: rescore all
  const uint2* lists = cand + (size_t)qi * RF_CAND_SHARDS * cap;
  auto cand_at = [&](uint32_t g) -> uint2 {
    int s = 0;
#pragma unroll
    for (int t = 1; t < RF_CAND_SHARDS; ++t) s += g >= off[t] ? 1 : 0;
    uint2 e = lists[(size_t)s * cap + (g - off[s])];
    // a row id past the corpus cannot come from the sweep (k_threshold zeroes the counters of
    // every search); should one appear (a caller sharing one workspace between concurrent
    // searches), never let it reach the row gather -- make it the worst candidate and flag
    if (e.x >= n_rows) {
      e.x = 0u;
      e.y = 0xFF800000u;  // -inf
      bad_row = true;
    }
    return e;
  };
  const float eps2 = 2.f * eps_in[qi];
  if (tid == 0) {
    *r_cnt = 0u;
    *t_cut = -INFINITY;
  }
  for (int d = tid; d < dim; d += MERGE_THREADS) ((_Float16*)qd)[d] = q[(size_t)qi * dim + d];

  if (c <= MERGE_RANK_MAX) {
    // ---- pass 1a: rank counting ------------------------------------------------
    unsigned long long key[MERGE_RANK_MAX / MERGE_THREADS];
#pragma unroll
    for (int i = 0; i < MERGE_RANK_MAX / MERGE_THREADS; ++i) {
      const uint32_t idx = (uint32_t)tid + MERGE_THREADS * i;
      key[i] = idx < c ? cand_key(cand_at(idx)) : 0ull;
      skeys[idx] = key[i];  // zero padding never outranks a real key
    }
    __syncthreads();
    if (have_cut) {
      uint32_t rank[MERGE_RANK_MAX / MERGE_THREADS];
#pragma unroll
      for (int i = 0; i < MERGE_RANK_MAX / MERGE_THREADS; ++i) rank[i] = 0;
      for (uint32_t j = 0; j < c; j += 8) {
        unsigned long long o[8];  // same addresses in every lane: LDS broadcast, 8 in flight
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = skeys[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < MERGE_RANK_MAX / MERGE_THREADS; ++i) rank[i] += o[u] > key[i] ? 1u : 0u;
      }
#pragma unroll
      for (int i = 0; i < MERGE_RANK_MAX / MERGE_THREADS; ++i)
        if (key[i] != 0ull && rank[i] == (uint32_t)(kk - 1)) *t_cut = key_score(key[i]) - eps2;
    }
    __syncthreads();
    // ---- pass 2a: compact R from the LDS keys -----------------------------------
    const float cut = *t_cut;
#pragma unroll
    for (int i = 0; i < MERGE_RANK_MAX / MERGE_THREADS; ++i) {
      if (key[i] != 0ull && key_score(key[i]) >= cut) {
        const uint32_t slot = atomicAdd(r_cnt, 1u);
        if (slot < RF_RESCORE_CAP) r_row[slot] = key_row(key[i]);
      }
    }
  } else {
    // ---- pass 1b: extraction rounds (per wave, then across waves) ---------------
    unsigned long long key[MERGE_PER_THREAD];
#pragma unroll
    for (int i = 0; i < MERGE_PER_THREAD; ++i) {
      const uint32_t idx = (uint32_t)tid + MERGE_THREADS * i;
      key[i] = idx < c ? cand_key(cand_at(idx)) : 0ull;
    }
    for (int r = 0; r < kk; ++r) {
      unsigned long long m = key[0];
#pragma unroll
      for (int i = 1; i < MERGE_PER_THREAD; ++i) m = key[i] > m ? key[i] : m;
      const unsigned long long wm = wave_max_u64(m);
      if (lane == 0) wtop[wave * RF_MAX_K + r] = wm;
      if (wm != 0ull && m == wm) {
#pragma unroll
        for (int i = 0; i < MERGE_PER_THREAD; ++i)
          if (key[i] == wm) key[i] = 0ull;
      }
    }
    __syncthreads();
    if (wave == 0) {
      constexpr int NV = (MERGE_THREADS / 64 * RF_MAX_K + 63) / 64;
      unsigned long long v[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;  // j -> (wave j / kk, slot j % kk)
        v[i] = (kk > 0 && j < (MERGE_THREADS / 64) * kk) ? wtop[(j / kk) * RF_MAX_K + (j % kk)] : 0ull;
      }
      unsigned long long kth = 0ull;
      for (int r = 0; r < kk; ++r) {
        unsigned long long m = v[0];
#pragma unroll
        for (int i = 1; i < NV; ++i) m = v[i] > m ? v[i] : m;
        const unsigned long long wm = wave_max_u64(m);
        kth = wm;
        if (wm != 0ull && m == wm) {
#pragma unroll
          for (int i = 0; i < NV; ++i)
            if (v[i] == wm) v[i] = 0ull;
        }
      }
      if (lane == 0 && have_cut) *t_cut = key_score(kth) - eps2;
    }
    __syncthreads();
    // ---- pass 2b: compact R from global ------------------------------------------
    const float cut = *t_cut;
    for (uint32_t idx = tid; idx < c; idx += MERGE_THREADS) {
      const uint2 e = cand_at(idx);
      if (__builtin_bit_cast(float, e.y) >= cut) {
        const uint32_t slot = atomicAdd(r_cnt, 1u);
        if (slot < RF_RESCORE_CAP) r_row[slot] = e.x;
      }
    }
  }
  __syncthreads();
  uint32_t R = *r_cnt;
  if (R > RF_RESCORE_CAP) {
    fl |= RF_FLAG_TIE_OVERFLOW;
    R = RF_RESCORE_CAP;
  }

  // ---- pass 3: exact fp64 scores, rows staged through LDS ---------------------------
  // 8 lanes per candidate: lane j owns chain j (dims j, j+8, ...); xor-shuffle tree.
  const int chunks = 2 * KS;
  const _Float16* qh = (const _Float16*)qd;  // query row as fp16 in LDS
  for (uint32_t base = 0; base < R; base += MERGE_STAGE_ROWS) {
    const uint32_t nb = (R - base) < MERGE_STAGE_ROWS ? (R - base) : MERGE_STAGE_ROWS;
    for (uint32_t idx = tid; idx < nb * (uint32_t)chunks; idx += MERGE_THREADS) {
      const uint32_t r = idx / chunks, ch = idx % chunks;
      srows[r * srow_stride + ch] = tiles[rf_chunk_index((int64_t)r_row[base + r], (int)ch, KS)];
    }
    __syncthreads();
    {
      const uint32_t r = (uint32_t)tid >> 3;   // 32 candidates x 8 lanes = 256 threads
      const int j = tid & 7;
      const _Float16* row = (const _Float16*)(srows + (r < nb ? r : 0) * srow_stride);
      double acc = 0.0;
#pragma unroll 8
      for (int ch = 0; ch < chunks; ++ch)
        acc = fma((double)qh[8 * ch + j], (double)row[8 * ch + j], acc);
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      acc += __shfl_xor(acc, 4);
      if (j == 0 && r < nb) r_exact[base + r] = acc;
    }
    __syncthreads();
  }

  // ---- pass 4: rank by (exact desc, row asc) and write --------------------------------
  for (uint32_t i = tid; i < R; i += MERGE_THREADS) {
    const double s = r_exact[i];
    const uint32_t row = r_row[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < R; ++j) rank += ranks_before(r_exact[j], r_row[j], s, row) ? 1u : 0u;
    if (rank < (uint32_t)k) {
      const size_t o = (size_t)qi * k + rank;
      scores[o] = (float)s;
      ids[o] = (int64_t)row + id_base;
      if (exact) exact[o] = s;
    }
  }
  const uint32_t filled = R < (uint32_t)k ? R : (uint32_t)k;
  for (uint32_t j = filled + tid; j < (uint32_t)k; j += MERGE_THREADS) {
    const size_t o = (size_t)qi * k + j;
    scores[o] = -INFINITY;
    ids[o] = -1;
    if (exact) exact[o] = -INFINITY;
  }
  if (flags && __syncthreads_or(bad_row ? 1 : 0)) fl |= RF_FLAG_CAND_OVERFLOW;
  if (tid == 0 && flags) flags[qi] = fl;
  // leave the counters zero (every thread of this workgroup read them before the barriers
  // above); k_threshold zeroes them again at the start of every search, so a search never
  // depends on what an earlier one -- or an error return mid-pipeline -- left behind
  if (tid < RF_CAND_SHARDS) cand_cnt_rw[qi * RF_CAND_SHARDS + tid] = 0u;
}
// ---- exhaustive exact path ----------------------------------------------------------
// Workgroup-level running top-k list (sorted, in LDS) updated 256 entries at a
// time by rank counting.  Used for queries the fused path could not prove
// (flags != 0) and as an on-device cross-check in the tests.
#define EX_THREADS 256
struct ExList {
  double s[RF_MAX_K];
  int64_t r[RF_MAX_K];
  int n;
};

__device__ void exlist_update(ExList* L, double* t_s, int64_t* t_r, int* t_n, int k, double s,
                              int64_t row, bool valid, int tid) {
  // does my entry beat the current k-th?
  bool pass = valid;
  if (valid && L->n >= k) pass = ranks_before(s, row, L->s[k - 1], L->r[k - 1]);
  if (__syncthreads_or(pass ? 1 : 0) == 0) return;
  if (tid == 0) *t_n = L->n;
  __syncthreads();
  if (tid < L->n) {
    t_s[tid] = L->s[tid];
    t_r[tid] = L->r[tid];
  }
  if (pass) {
    const int slot = atomicAdd(t_n, 1);
    t_s[slot] = s;
    t_r[slot] = row;
  }
  __syncthreads();
  const int m = *t_n;
  for (int i = tid; i < m; i += EX_THREADS) {
    const double si = t_s[i];
    const int64_t ri = t_r[i];
    int rank = 0;
    for (int j = 0; j < m; ++j) rank += ranks_before(t_s[j], t_r[j], si, ri) ? 1 : 0;
    if (rank < k) {
      L->s[rank] = si;
      L->r[rank] = ri;
    }
  }
  if (tid == 0) L->n = m < k ? m : k;
  __syncthreads();
}

__global__ void __launch_bounds__(EX_THREADS) k_exhaustive_scan(
    const _Float16* __restrict__ q, int dim, int KS, const uint4* __restrict__ tiles, int64_t n_rows,
    int k, double* __restrict__ out_s, int64_t* __restrict__ out_r,
    const double* __restrict__ after_s, const int64_t* __restrict__ after_r, int64_t id_base) {
  __shared__ ExList L;
  __shared__ double t_s[RF_MAX_K + EX_THREADS];
  __shared__ int64_t t_r[RF_MAX_K + EX_THREADS];
  __shared__ int t_n;
  const int tid = threadIdx.x;
  const int qi = blockIdx.y;
  if (tid == 0) L.n = 0;
  __syncthreads();
  const _Float16* qrow = q + (size_t)qi * dim;
  // paging: only rows ranked strictly AFTER (after_s, after_r) are eligible
  const bool paged = after_s != nullptr;
  const double bs = paged ? after_s[qi] : 0.0;
  const int64_t br = paged ? after_r[qi] - id_base : 0;
  const int64_t step = (int64_t)gridDim.x * EX_THREADS;
  for (int64_t base = (int64_t)blockIdx.x * EX_THREADS; base < n_rows; base += step) {
    const int64_t row = base + tid;
    bool valid = row < n_rows;
    const double s = valid ? exact_dot(qrow, tiles, row, KS) : 0.0;
    if (paged) valid = valid && ranks_before(bs, br, s, row);
    exlist_update(&L, t_s, t_r, &t_n, k, s, row, valid, tid);
  }
  const size_t o = ((size_t)qi * gridDim.x + blockIdx.x) * RF_MAX_K;
  if (tid < RF_MAX_K) {
    out_s[o + tid] = tid < L.n ? L.s[tid] : -INFINITY;
    out_r[o + tid] = tid < L.n ? L.r[tid] : -1;
  }
}

__global__ void __launch_bounds__(EX_THREADS) k_exhaustive_final(
    const double* __restrict__ in_s, const int64_t* __restrict__ in_r, int lists, int k,
    int64_t id_base, float* __restrict__ scores, int64_t* __restrict__ ids,
    double* __restrict__ exact) {
  __shared__ ExList L;
  __shared__ double t_s[RF_MAX_K + EX_THREADS];
  __shared__ int64_t t_r[RF_MAX_K + EX_THREADS];
  __shared__ int t_n;
  const int tid = threadIdx.x;
  const int qi = blockIdx.x;
  if (tid == 0) L.n = 0;
  __syncthreads();
  const size_t base = (size_t)qi * lists * RF_MAX_K;
  const int total = lists * RF_MAX_K;
  for (int b = 0; b < total; b += EX_THREADS) {
    const int i = b + tid;
    const bool valid = i < total && in_r[base + i] >= 0;
    const double s = valid ? in_s[base + i] : 0.0;
    const int64_t row = valid ? in_r[base + i] : 0;
    exlist_update(&L, t_s, t_r, &t_n, k, s, row, valid, tid);
  }
  for (int j = tid; j < k; j += EX_THREADS) {
    const size_t o = (size_t)qi * k + j;
    if (j < L.n) {
      scores[o] = (float)L.s[j];
      ids[o] = L.r[j] + id_base;
      if (exact) exact[o] = L.s[j];
    } else {
      scores[o] = -INFINITY;
      ids[o] = -1;
      if (exact) exact[o] = -INFINITY;
    }
  }
}

// ---- cross-shard merge ---------------------------------------------------------------
// in [W][B][k] (exact fp64, global id) -> out [B][k].  Ids are global and unique,
// so ranking by (score desc, id asc) reproduces the single-device order bit for bit.
__global__ void __launch_bounds__(EX_THREADS) k_merge_shards(
    const double* __restrict__ in_s, const int64_t* __restrict__ in_r, size_t sstride,
    int W, int B, int k, float* __restrict__ scores, int64_t* __restrict__ ids,
    const uint32_t* __restrict__ flags_in, size_t fstride, uint32_t* __restrict__ flags_out) {
  // entry (shard w, query qi, rank j) sits at w * sstride + qi * k + j in both arrays; shard w's
  // per-query exactness flags (optional) at flags_in[w * fstride + qi]: the merged answer of a
  // query is proven exact only if EVERY shard's local answer was, so the output flag is their OR
  const int qi = blockIdx.x;
  const int tid = threadIdx.x;
  const int m = W * k;
  const size_t qoff = (size_t)qi * k;
  const size_t ooff = (size_t)qi * k;
  if (flags_out && tid == 0) {
    uint32_t f = 0u;
    if (flags_in)
      for (int w = 0; w < W; ++w) f |= flags_in[(size_t)w * fstride + qi];
    flags_out[qi] = f;
  }
  for (int i = tid; i < m; i += EX_THREADS) {
    const int w = i / k, j = i % k;
    const size_t src = (size_t)w * sstride + qoff + j;
    const int64_t ri = in_r[src];
    if (ri < 0) continue;
    const double si = in_s[src];
    int rank = 0;
    for (int i2 = 0; i2 < m; ++i2) {
      const size_t s2 = (size_t)(i2 / k) * sstride + qoff + (i2 % k);
      const int64_t r2 = in_r[s2];
      if (r2 >= 0 && ranks_before(in_s[s2], r2, si, ri)) ++rank;
    }
    if (rank < k) {
      scores[ooff + rank] = (float)si;
      ids[ooff + rank] = ri;
    }
  }
  // number of valid entries decides how many tail slots stay empty
  __shared__ int n_valid;
  if (tid == 0) n_valid = 0;
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < m; i += EX_THREADS)
    mine += in_r[(size_t)(i / k) * sstride + qoff + (i % k)] >= 0 ? 1 : 0;
  if (mine) atomicAdd(&n_valid, mine);
  __syncthreads();
  for (int j = n_valid + tid; j < k; j += EX_THREADS) {
    scores[ooff + j] = -INFINITY;
    ids[ooff + j] = -1;
  }
}

// ---- host side --------------------------------------------------------------------------
int rf_launch_threshold(const rf_index* ix, const void* q, int B, int k, int P,
                        const rf_workspace& ws, hipStream_t st) {
  // one wave per query slot of the sweep (64, or up to RF_QWIDE for a wide sweep)
  hipLaunchKernelGGL(k_threshold, dim3(B > RF_QCHUNK ? RF_QWIDE : RF_QCHUNK), dim3(64), 0, st, (const _Float16*)q, B, ix->dim,
                     k, ws.pmax, P, ix->max_norm2, ws.thr, ws.eps, ws.cand_cnt);
  RF_HIP(hipGetLastError());
  return RF_OK;
}

int rf_launch_merge(const rf_index* ix, const void* q, int B, int k, int64_t id_base,
                    const rf_workspace& ws, float* scores, int64_t* ids, double* exact,
                    uint32_t* flags, hipStream_t st) {
  const size_t lds = merge_lds_bytes(ix->dim);
  static rf_lds_attr lds_attr;
  RF_HIP(rf_ensure_lds(lds_attr, (const void*)k_merge, lds));
  hipLaunchKernelGGL(k_merge, dim3(B), dim3(MERGE_THREADS), lds, st, (const _Float16*)q, ix->dim,
                     ix->KS, ix->tiles, k, id_base, ws.cand_cnt, ws.cand, (uint32_t)RF_SHARD_CAP,
                     ws.eps, scores, ids, exact, flags, ws.cand_cnt, (uint32_t)ix->size);
  RF_HIP(hipGetLastError());
  return RF_OK;
}

int rf_launch_exhaustive(const rf_index* ix, const void* q, int B, int k, int64_t id_base,
                         const rf_workspace& ws, float* scores, int64_t* ids, double* exact,
                         const double* after_s, const int64_t* after_r, hipStream_t st) {
  int64_t need = (ix->size + EX_THREADS - 1) / EX_THREADS;
  int lists = (int)(need < RF_EX_WGS ? (need < 1 ? 1 : need) : RF_EX_WGS);
  hipLaunchKernelGGL(k_exhaustive_scan, dim3(lists, B), dim3(EX_THREADS), 0, st,
                     (const _Float16*)q, ix->dim, ix->KS, ix->tiles, ix->size, k, ws.ex_score,
                     ws.ex_row, after_s, after_r, id_base);
  hipLaunchKernelGGL(k_exhaustive_final, dim3(B), dim3(EX_THREADS), 0, st, ws.ex_score, ws.ex_row,
                     lists, k, id_base, scores, ids, exact);
  RF_HIP(hipGetLastError());
  return RF_OK;
}

int rf_launch_merge_shards(const double* exact, const int64_t* ids, size_t shard_stride, int W, int B, int k,
                           float* scores_out, int64_t* ids_out, const uint32_t* flags_in, size_t flag_stride,
                           uint32_t* flags_out, hipStream_t st) {
  hipLaunchKernelGGL(k_merge_shards, dim3(B), dim3(EX_THREADS), 0, st, exact, ids, shard_stride, W, B, k,
                     scores_out, ids_out, flags_in, flag_stride, flags_out);
  RF_HIP(hipGetLastError());
  return RF_OK;
}
