// Sentence-embedder forward pass (BERT encoder + mean-pool + L2-normalise) for the
// all-MiniLM-L6-v2 family (hidden 384, 12 heads x 32), hand-written for gfx950.
// Stands in for SentenceTransformer('all-MiniLM-L6-v2').encode(...)
// (vector_rag_mcp/main.py:41,50; retrieve.py:14,27; "chunking_storing (1).py":8,380).
//
// Data layout
//   * tokens are PACKED: only the lens[b] valid tokens of each sequence occupy
//     rows of the activation matrices (M = sum lens), so padding costs nothing and
//     needs no attention mask; the matrices themselves are fragment-tiled (toff());
//   * every Linear weight [out, in] is stored once in the same 32-row x 16-k MFMA
//     fragment tiling as the corpus (rf_internal.h): one 1-KiB wave load = one
//     A operand of v_mfma_f32_32x32x16_f16.
//
// Kernels
//   k_tok_offsets   lens -> packed row offsets
//   k_embed_ln      word + position + type embedding gather, LayerNorm        (K1)
//   the Linears     Y^T = W X^T on the matrix cores with the epilogues fused -- bias (QKV, K2) | bias + GELU
//                   (K5) | bias + residual + LayerNorm (K4, K6) -- in one of four forms chosen by the batch's
//                   token slots (launch_linear):
//     k_linear_small  <= 1024 slots (queries): output features spread over the chip, LayerNorm as k_ln_rows
//     k_linear        < 8192: a workgroup owns 32 or 64 tokens x 384 features, operands straight from L1/L2
//     k_linear_dma    >= 8192, QKV and FFN1: the wave's 32 tokens resident in registers, weights through an
//                     LDS-DMA ring, the epilogue of block b deferred into the MFMA stream of block b + 1
//     k_gemm_tile     >= 8192, out-projection and FFN2 (LayerNorm epilogues): both operands through a 4-slot
//                     LDS-DMA ring, 128 tokens x 384 features per workgroup
//   k_attention_mfma softmax(Q K^T / sqrt(32)) V per (sequence, head) on the matrix cores
//                   for T <= 256 (K3); k_attention = vector-ALU online-softmax form
//                   kept for longer sequences
//   k_qkv_attn_one  a single sequence of <= 32 tokens: K2 + K3 of a head in one launch
//   k_pool_norm     mean over the sequence, L2-normalise                      (K7)
#include "encoder_internal.h"
#include <mutex>
#include <new>
#include <stdlib.h>
#include <vector>

struct rf_encoder {
  rf_encoder_config cfg;
  rf_encoder_weights w;      // row-major originals (embeddings, biases, LN)
  int device;
  const uint4* qkv_t;        // tiled [L][3H/32][H/16][64]
  const uint4* ao_t;         // tiled [L][H/32][H/16][64]
  const uint4* ff1_t;        // tiled [L][I/32][H/16][64]
  const uint4* ff2_t;        // tiled [L][H/32][I/16][64]
  const uint4* post_t;       // per-layer packs of k_post_block [L][PB_PACK_FRAGS][64] (encoder_post.hip)
  // small batches: the ~45 launches of a forward as ONE hipGraph per (shape, buffers) -- a
  // 12-token query is launch-bound (5 us of host time per launch against 2-5 us of kernel)
  struct Graph {
    int B, T;
    const void *ids, *lens;
    void *o16, *o32, *ws;
    int tuning_gen;        // rf_set_tuning generation the graph was captured under
    hipGraphExec_t exec;   // nullptr: seen once (the plain run also sets the kernels' attributes)
    bool dead;             // capture failed for this key: stay on plain launches
    hipEvent_t done;       // recorded behind the last launch of exec: an evicted exec is destroyed only once it has run
  };
  mutable std::vector<Graph> graphs;            // least recently used first
  mutable std::vector<Graph> retired;           // evicted while their last launch may still be in flight
  mutable hipStream_t cap_stream = nullptr;
  mutable std::mutex mu;
};

static bool cfg_supported(const rf_encoder_config* c) {
  return c && c->hidden == HID && c->heads > 0 && c->hidden / c->heads == HEAD_DIM &&
         c->hidden % c->heads == 0 && c->intermediate == 4 * HID &&
         c->layers > 0 && c->vocab_size > 0 && c->max_position > 0 && c->type_vocab > 0;
}

static size_t layer_weight_elems(const rf_encoder_config* c) {
  return (size_t)3 * HID * HID + (size_t)HID * HID + (size_t)2 * c->intermediate * HID;
}

extern "C" size_t rf_encoder_storage_bytes(const rf_encoder_config* cfg) {
  if (!cfg_supported(cfg)) return 0;
  return (size_t)cfg->layers * (layer_weight_elems(cfg) + rf_post_pack_elems()) * sizeof(_Float16);
}

extern "C" int rf_encoder_create(rf_encoder_t** out, const rf_encoder_config* cfg,
                                 const rf_encoder_weights* w, void* storage_dev,
                                 size_t storage_bytes, int device, void* stream) {
  if (!out || !cfg || !w || !storage_dev) {
    rf_set_error("rf_encoder_create: null argument");
    return RF_ERR_INVALID;
  }
  *out = nullptr;
  if (!cfg_supported(cfg)) {
    rf_set_error("rf_encoder_create: unsupported config (need hidden 384, head_dim 32, "
                 "intermediate 1536); got hidden=%d heads=%d intermediate=%d",
                 cfg->hidden, cfg->heads, cfg->intermediate);
    return RF_ERR_UNSUPPORTED;
  }
  if (storage_bytes < rf_encoder_storage_bytes(cfg) || ((uintptr_t)storage_dev & 15)) {
    rf_set_error("rf_encoder_create: storage too small or misaligned");
    return RF_ERR_CAPACITY;
  }
  const void* const* ptrs = (const void* const*)w;
  for (size_t i = 0; i < sizeof(rf_encoder_weights) / sizeof(void*); ++i)
    if (!ptrs[i] || ((uintptr_t)ptrs[i] & 15)) {
      rf_set_error("rf_encoder_create: weight pointer %zu null or not 16-byte aligned", i);
      return RF_ERR_INVALID;
    }
  int rc = rf_device_check(device);
  if (rc != RF_OK) return rc;
  rf_encoder* e = new (std::nothrow) rf_encoder();
  if (!e) {
    rf_set_error("out of host memory");
    return RF_ERR_INVALID;
  }
  e->cfg = *cfg;
  e->w = *w;
  e->device = device;
  hipStream_t st = (hipStream_t)stream;
  const int L = cfg->layers, I = cfg->intermediate;
  _Float16* base = (_Float16*)storage_dev;
  _Float16* qkv = base;
  _Float16* ao = qkv + (size_t)L * 3 * HID * HID;
  _Float16* ff1 = ao + (size_t)L * HID * HID;
  _Float16* ff2 = ff1 + (size_t)L * I * HID;
  _Float16* post = ff2 + (size_t)L * HID * I;
  // [L*out, in] row-major -> tiled; out is a multiple of 32 so layers tile independently
  rf_launch_tile_rows(w->qkv_w, (uint4*)qkv, 0, (int64_t)L * 3 * HID, HID / 16, st);
  rf_launch_tile_rows(w->ao_w, (uint4*)ao, 0, (int64_t)L * HID, HID / 16, st);
  rf_launch_tile_rows(w->ff1_w, (uint4*)ff1, 0, (int64_t)L * I, HID / 16, st);
  rf_launch_tile_rows(w->ff2_w, (uint4*)ff2, 0, (int64_t)L * HID, I / 16, st);
  rf_launch_post_pack_build(w, post, L, st);
  RF_HIP(hipGetLastError());
  e->qkv_t = (const uint4*)qkv;
  e->ao_t = (const uint4*)ao;
  e->ff1_t = (const uint4*)ff1;
  e->ff2_t = (const uint4*)ff2;
  e->post_t = (const uint4*)post;
  *out = e;
  return RF_OK;
}

extern "C" int rf_encoder_destroy(rf_encoder_t* enc) {
  if (enc) {
    for (auto* v : {&enc->graphs, &enc->retired})
      for (auto& g : *v) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.done) (void)hipEventDestroy(g.done);
      }
    if (enc->cap_stream) (void)hipStreamDestroy(enc->cap_stream);
  }
  delete enc;
  return RF_OK;
}

#define SM_MAX_TOK 1024   // token slots (B * T) up to which the small-batch GEMM path is taken
// ---- workspace ---------------------------------------------------------------------
struct EncWs {
  int32_t* tok_off;   // [B + 1]
  _Float16* x;        // [Mpad, 384]
  _Float16* y;        // [Mpad, 384]
  _Float16* qkv;      // [Mpad, 1152]
  _Float16* ctx;      // [Mpad, 384]
  _Float16* ff;       // [Mpad, I]
  float* pre;         // [min(Mpad, SM_MAX_TOK), 384] fp32: pre-LayerNorm sums of the small-batch path
};

static size_t enc_carve(unsigned char* base, int B, int T, int I, EncWs* ws) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    unsigned char* p = base ? base + off : nullptr;
    off = (off + bytes + 255) / 256 * 256;
    return p;
  };
  const size_t Mpad = ((size_t)B * T + 255) / 256 * 256;   // k_linear_dma reads whole 128- / 256-token tiles
  int32_t* tok = (int32_t*)take(((size_t)B + 1) * 4);
  _Float16* x = (_Float16*)take(Mpad * HID * 2);
  _Float16* y = (_Float16*)take(Mpad * HID * 2);
  _Float16* qkv = (_Float16*)take(Mpad * 3 * HID * 2);
  _Float16* ctx = (_Float16*)take(Mpad * HID * 2);
  _Float16* ff = (_Float16*)take(Mpad * (size_t)I * 2);
  float* pre = (float*)take((Mpad < SM_MAX_TOK ? Mpad : (size_t)SM_MAX_TOK) * HID * 4);
  if (ws) *ws = EncWs{tok, x, y, qkv, ctx, ff, pre};
  return off;
}

extern "C" size_t rf_encode_workspace_bytes(const rf_encoder_t* enc, int B, int T) {
  if (!enc || B <= 0 || T <= 0) return 0;
  return enc_carve(nullptr, B, T, enc->cfg.intermediate, nullptr);
}

// ---- kernels -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ void __launch_bounds__(256) k_tok_offsets(const int32_t* __restrict__ lens, int B, int T,
                                                     int32_t* __restrict__ tok_off) {
  // exclusive scan of clamp(lens, 0, T) by one workgroup
  __shared__ int32_t part[256];
  const int tid = threadIdx.x;
  const int per = (B + 255) / 256;
  int32_t s = 0;
  for (int i = 0; i < per; ++i) {
    const int b = tid * per + i;
    if (b < B) s += min(max(lens[b], 0), T);
  }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int32_t run = 0;
    for (int i = 0; i < 256; ++i) {
      const int32_t v = part[i];
      part[i] = run;
      run += v;
    }
    tok_off[B] = run;
  }
  __syncthreads();
  int32_t run = part[tid];
  for (int i = 0; i < per; ++i) {
    const int b = tid * per + i;
    if (b < B) {
      tok_off[b] = run;
      run += min(max(lens[b], 0), T);
    }
  }
}

// One workgroup per 32 consecutive positions of one sequence, lane = (position c, feature half h); its four
// waves take six of the 24 16-feature groups each.  The wave's stores are runs of whole 16-byte slots of the
// tiled activations (a token's slot of fragment f is next to its neighbour's: 512 contiguous bytes per half,
// split at most once by a token-block boundary) instead of 48 slots 1 KiB apart per token; every load of a
// wave (18 x 16 bytes per lane) is in flight at once and its 48 sums stay in registers; the LayerNorm sums are
// lane-local plus one xor-32 exchange and one trip through LDS between the four waves.  The per-sequence
// scalars (length, packed offset) are wave-uniform scalar loads.  (Round-2 history: one wave per token, four
// tokens per wave in a row, each a chain of three dependent vector loads: 49 us per 64 k-token batch; one wave
// per 32 positions with two sweeps over the rows: 56 us -- 8 waves per CU cannot hide the latency.)
__global__ void __launch_bounds__(256) k_embed_ln(
    const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
    int32_t* __restrict__ tok_off, int B, int T, int vocab, const _Float16* __restrict__ word,
    const _Float16* __restrict__ pos, const _Float16* __restrict__ type, const _Float16* __restrict__ g,
    const _Float16* __restrict__ b, float eps, _Float16* __restrict__ out) {
  __shared__ float red[2][4][32];
  constexpr int FW = HID / 16 / 4;                             // feature groups per wave (6)
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int cpr = (T + 31) >> 5;                               // 32-position chunks per sequence row
  const int bi = blockIdx.x / cpr, p0 = (blockIdx.x % cpr) * 32;
  const int len = min(max(lens[bi], 0), T);
  // ONE sequence (a query): the packed offsets are {0, len} -- written here, so that the launch of k_tok_offsets
  // (a dependent kernel boundary, ~4.5 us of the query's latency) is not needed
  if (B == 1 && blockIdx.x == 0 && threadIdx.x == 0) {
    tok_off[0] = 0;
    tok_off[1] = len;
  }
  if (p0 >= len) return;                                       // workgroup-uniform
  const int p = p0 + c;
  const bool live = p < len;
  int id = live ? ids[(size_t)bi * T + p] : 0;
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const int f0 = 16 * FW * wave + 8 * h;                       // the lane's first feature
  const _Float16* wrow = word + (size_t)id * HID + f0;
  const _Float16* prow = pos + (size_t)(live ? p : 0) * HID + f0;
  const _Float16* trow = type + f0;
  half8 a[FW], cc[FW], d[FW];
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    a[f] = *(const half8*)(wrow + 16 * f);
    cc[f] = *(const half8*)(prow + 16 * f);
    d[f] = *(const half8*)(trow + 16 * f);
  }
  float v[FW][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int f = 0; f < FW; ++f)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[f][j] = (float)a[f][j] + (float)cc[f][j] + (float)d[f][j];
      s1 += v[f][j];
      s2 = fmaf(v[f][j], v[f][j], s2);
    }
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  if (h == 0) {
    red[0][wave][c] = s1;
    red[1][wave][c] = s2;
  }
  // the scale and shift are not needed before the statistics: their round trip sits under the exchange
  half8 gg[FW], bb[FW];
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    gg[f] = *(const half8*)(g + f0 + 16 * f);
    bb[f] = *(const half8*)(b + f0 + 16 * f);
  }
  __syncthreads();
  const float t1 = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
  const float t2 = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  const float mu = t1 * (1.f / HID);
  // E[v^2] - mu^2 in fp32 over 384 values of order 1 with |mu| << 1: the cancellation is ~1e-6 relative
  const float rstd = rsqrtf(fmaxf(t2 * (1.f / HID) - mu * mu, 0.f) + eps);
  const int token = (B == 1 ? 0 : tok_off[bi]) + p;
  _Float16* orow = out + ((size_t)(token >> 5) * (HID / 16) * 64 + (size_t)h * 32 + (token & 31)) * 8 + (size_t)(FW * wave) * 512;
#pragma unroll
  for (int f = 0; f < FW; ++f) {
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (_Float16)((v[f][j] - mu) * rstd * (float)gg[f][j] + (float)bb[f][j]);
    if (live) *(half8*)(orow + (size_t)f * 512) = o;
  }
}

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RES_LN = 2 };

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 output
// step): one v_rcp, one v_exp and six fma instead of libm erff's ~50 instructions --
// the GELU epilogue of the 1536-wide FFN GEMM evaluates it 50 M times per layer call.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float y = fmaf(t, 1.061405429f, -1.453152027f);
  y = fmaf(t, y, 1.421413741f);
  y = fmaf(t, y, -0.284496736f);
  y = fmaf(t, y, 0.254829592f);
  y *= t;
  const float r = 1.0f - y * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float y) {
  return 0.5f * y * (1.f + erf_fast(y * 0.70710678118654752f));
}
// The same arithmetic on two elements per lane with v_pk_fma_f32 / v_pk_mul_f32 (full rate on
// two floats): 12 packed ops + 2 rcp + 2 exp per PAIR instead of ~14 + 2 per element.  The
// GELU epilogue of the 1536-wide FFN GEMM is VALU-bound (50 M evaluations per layer call).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 y) {
  // erf(x) = x P(t), t = 2 x^2 / c^2 - 1, x clamped to +-c = 3.2 (erf(3.2) = 1 - 6e-6): a degree-10
  // minimax fit, max |error| 4.8e-6 in fp32 Horner form (gelu: 2e-5 absolute, far below the fp16
  // the result is stored in).  No transcendental: the rcp + exp2 form it replaces (Abramowitz-
  // Stegun 7.1.26) spent 4 quarter-rate instructions per pair and made the FFN1 epilogue as long
  // as the tile's MFMAs; this one is 13 packed FMAs/multiplies + 2 clamps per pair.
  constexpr float C = 3.2f;
  const f32x2 x = y * 0.70710678118654752f;
  f32x2 xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -C, C);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -C, C);
  const f32x2 xs = xc * (2.0f / (C * C));
  const f32x2 t = __builtin_elementwise_fma(xc, xs, f32x2{-1.f, -1.f});
#define RF_P2(v) f32x2{v, v}
  f32x2 p = __builtin_elementwise_fma(t, RF_P2(2.395397033e-03f), RF_P2(-6.752740320e-03f));
  p = __builtin_elementwise_fma(t, p, RF_P2(9.277549144e-03f));
  p = __builtin_elementwise_fma(t, p, RF_P2(-1.580625450e-02f));
  p = __builtin_elementwise_fma(t, p, RF_P2(3.215588812e-02f));
  p = __builtin_elementwise_fma(t, p, RF_P2(-5.433418336e-02f));
  p = __builtin_elementwise_fma(t, p, RF_P2(8.094794964e-02f));
  p = __builtin_elementwise_fma(t, p, RF_P2(-1.137384297e-01f));
  p = __builtin_elementwise_fma(t, p, RF_P2(1.543205805e-01f));
  p = __builtin_elementwise_fma(t, p, RF_P2(-2.173020031e-01f));
  p = __builtin_elementwise_fma(t, p, RF_P2(4.413347567e-01f));
#undef RF_P2
  const f32x2 r = xc * p;   // erf(x), |r| < 1
  const f32x2 hy = y * 0.5f;
  return __builtin_elementwise_fma(hy, r, hy);
}

// Y[tokens, N] = X[tokens, K] W^T + b with W in the fragment tiling.  Workgroup =
// 32*NTB tokens x 384 features (grid.y picks the 384-feature group); wave w owns
// features 96 w .. +96 for all NTB token blocks: 3*NTB accumulators, each weight
// fragment feeds NTB MFMAs.  A operand = weight fragment (one contiguous 1-KiB wave
// load), B operand = 32 token rows read straight from the row-major activation
// (16 B per lane; the four waves' identical reads hit L1).  No LDS and no barrier in
// the k-loop: staging X through LDS with a barrier per 64-k stage was measured 25-75 %
// SLOWER here (12-24 MFMAs between barriers cannot hide them at this occupancy).
// The accumulator holds the TOKEN on the lane and FEATURES in registers: 4 consecutive
// features per register quad -> 8-byte stores, and lane-local LayerNorm partial sums.
// The k-loop is latency-bound unless several k-steps are in flight (a k-step is only
// 3*NTB MFMAs = 96-192 matrix cycles against ~1500 cycles to L2): both operands run
// through a 4-deep REGISTER RING -- fragment set d is re-armed with k-step kk+4 right
// after k-step kk has consumed it, with unconditional loads so hipcc can count them
// (20 KB in flight per wave).  KS = K/16 is a template parameter (24 or 96).
// Copying the workgroup's X tile to LDS once (the four waves need the same B
// fragments) was measured SLOWER in three forms (row-major + barrier per stage, whole
// tile up front, tiled copy + this W ring: QKV 41 -> 51 us, FFN1 66 -> 80 us): the
// redundant per-wave fragment loads are L1 hits and cost less than the LDS round trip.
template <int EPI, int NTB, int KS>   // NTB = 32-token blocks per workgroup
__global__ void __launch_bounds__(256) k_linear(
    const _Float16* __restrict__ X, int K, const uint4* __restrict__ Wt,
    const _Float16* __restrict__ bias, _Float16* __restrict__ out, int ldo,
    const int32_t* __restrict__ m_ptr, const _Float16* __restrict__ res,
    const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta, float eps) {
  constexpr int LIN_TOK = 32 * NTB;
  __shared__ float red[2][4][LIN_TOK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int t0 = blockIdx.x * LIN_TOK;
  const int M = *m_ptr;
  if (t0 >= M) return;  // whole workgroup: no barrier is skipped by a subset
  const int fgroup = blockIdx.y * 384 + wave * 96;  // first feature of this wave
  const uint4* wbase = Wt + (size_t)(fgroup / 32) * KS * 64 + lane;

  f32x16 acc[NTB][3];
#pragma unroll
  for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
    for (int fb = 0; fb < 3; ++fb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[tb][fb][i] = 0.f;

  {
    constexpr int D = 4;
    static_assert(KS % D == 0, "ring depth must divide the k-steps");
    const _Float16* xfrag = X + ((size_t)(t0 >> 5) * KS * 64 + lane) * 8;   // fragment (block, kk) = 1 KiB
    half8 xr[D][NTB];
    uint4 wr[D][3];
    auto arm = [&](int d, int kk) {
#pragma unroll
      for (int tb = 0; tb < NTB; ++tb) xr[d][tb] = *(const half8*)(xfrag + ((size_t)tb * KS + kk) * 512);
#pragma unroll
      for (int fb = 0; fb < 3; ++fb) wr[d][fb] = wbase[((size_t)fb * KS + kk) * 64];
    };
    auto consume = [&](int d) {
#pragma unroll
      for (int fb = 0; fb < 3; ++fb)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb)
          acc[tb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wr[d][fb]),
                                                               xr[d][tb], acc[tb][fb], 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d) arm(d, d);
    __builtin_amdgcn_sched_barrier(0);
    for (int kk0 = 0; kk0 < KS - D; kk0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        consume(d);
        arm(d, kk0 + d + D);
        // pin the re-arm here: left free, hipcc sinks the loads to just before their
        // use four k-steps later (98 VGPRs, nothing in flight, no speed-up)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) consume(d);
  }

  // acc[tb][fb][4 g + j] = Y[t0 + 32 tb + c][fgroup + 32 fb + 8 g + 4 h + j]
#pragma unroll
  for (int tb = 0; tb < NTB; ++tb) {
    const int token = t0 + tb * 32 + c;
    float v[3][16];
#pragma unroll
    for (int fb = 0; fb < 3; ++fb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = fgroup + 32 * fb + 8 * g + 4 * h;
        const half4 bv = *(const half4*)(bias + f);
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            f32x2 y;
            y[0] = acc[tb][fb][4 * g + j] + (float)bv[j];
            y[1] = acc[tb][fb][4 * g + j + 1] + (float)bv[j + 1];
            y = gelu_erf2(y);
            v[fb][4 * g + j] = y[0];
            v[fb][4 * g + j + 1] = y[1];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[fb][4 * g + j] = acc[tb][fb][4 * g + j] + (float)bv[j];
        }
        if (EPI == EPI_BIAS_RES_LN) {
          const half4 rv = *(const half4*)(res + toff(token, f, HID / 16));
#pragma unroll
          for (int j = 0; j < 4; ++j) v[fb][4 * g + j] += (float)rv[j];
        }
      }

    float mu = 0.f, rstd = 1.f;
    if (EPI == EPI_BIAS_RES_LN) {
      // LayerNorm over the 384 features of a token: 48 per lane, x2 lane halves, x4 waves
      float s = 0.f;
#pragma unroll
      for (int fb = 0; fb < 3; ++fb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[fb][i];
      s += __shfl_xor(s, 32);
      if (h == 0) red[0][wave][tb * 32 + c] = s;
      __syncthreads();
      const int tc = tb * 32 + c;
      mu = (red[0][0][tc] + red[0][1][tc] + red[0][2][tc] + red[0][3][tc]) * (1.f / HID);
      float q = 0.f;
#pragma unroll
      for (int fb = 0; fb < 3; ++fb)
#pragma unroll
        for (int i = 0; i < 16; ++i) q += (v[fb][i] - mu) * (v[fb][i] - mu);
      q += __shfl_xor(q, 32);
      if (h == 0) red[1][wave][tc] = q;
      __syncthreads();
      rstd = rsqrtf((red[1][0][tc] + red[1][1][tc] + red[1][2][tc] + red[1][3][tc]) * (1.f / HID) + eps);
    }

    if (token < M) {
#pragma unroll
      for (int fb = 0; fb < 3; ++fb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f = fgroup + 32 * fb + 8 * g + 4 * h;
          half4 o;
          if (EPI == EPI_BIAS_RES_LN) {
            const half4 gv = *(const half4*)(gamma + f);
            const half4 be = *(const half4*)(beta + f);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              o[j] = (_Float16)((v[fb][4 * g + j] - mu) * rstd * (float)gv[j] + (float)be[j]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (_Float16)v[fb][4 * g + j];
          }
          *(half4*)(out + toff(token, f, ldo / 16)) = o;
        }
    }
  }
}

// ---- K = 384 GEMMs with plain epilogues (QKV, FFN1): weights through an LDS-DMA ring -------
// k_linear above loads BOTH operands of every k-step straight from L1/L2 into registers, per
// wave: 5 KiB of fragment loads per 6 MFMAs, and the texture-addresser path, not the matrix
// pipe, sets its pace (~1.7 x the MFMA time).  This form is the batch-256 corpus sweep
// (scan_wide.hip) with the roles renamed:
//   * a wave keeps ITS 32 tokens' activations resident as 24 B-operand fragments (96 registers,
//     pinned to the accumulator half of the register file) for the whole kernel;
//   * the weight matrix streams through a 3-slot LDS ring by LDS-DMA, two 32-feature blocks
//     (48 KiB) per phase, loads two phases ahead behind a counted vmcnt and a raw s_barrier;
//     all eight waves of the workgroup read it from LDS (inline-asm ds_read_b128, lds_ring.h),
//     so a weight byte crosses the load path once per 128 tokens instead of once per wave;
//   * workgroup = 128 tokens x ALL N features: waves w and w + 4 share a token block and a
//     SIMD and take the even / odd feature block of each phase, so one's epilogue (bias, GELU,
//     fp16 stores) runs under the other's MFMAs;
//   * the bias vector sits in LDS and is read with inline-asm ds_read_b64 (an ordinary global
//     load in the loop would make hipcc wait vmcnt(0) and drain the ring).
#define LD_TOK 128
#define LD_WAVES 8
#define LD_SLOTS 3
#define LD_FRAGS 48                       // fragments per phase: 2 feature blocks x 24 k-steps
#define LD_PW (LD_FRAGS / LD_WAVES)       // LDS-DMA pieces per wave and phase (6)
// WIDE = 0: 128 tokens per workgroup, waves w / w + 4 share a token block and split each phase's
// two feature blocks; WIDE = 1: 256 tokens per workgroup, every wave owns a token block and takes
// BOTH feature blocks of a phase (the per-wave fixed costs -- six LDS-DMA issues, a barrier --
// are paid once per 48 MFMAs instead of once per 24, and a weight byte serves 256 tokens).
// GELU for the deferred epilogue below: plain (unpacked) fp32 only.  Measured on gfx950
// (tools/ubench/mfma_valu.hip): v_fma_f32 issues under the MFMAs of either wave of the SIMD at
// almost no cost (8 MFMA + 32 v_fma_f32 in one wave: 276 cycles against 264 for the MFMAs alone),
// v_pk_fma_f32 does not (484 cycles) -- the packed form is only worth its two lanes where no MFMA
// is in flight.  h(y) = y (1/2 + yc Q(t)), yc = y clamped to +-3.2 sqrt 2, t = yc^2 / 3.2^2 - 1:
// the erf polynomial of gelu_erf2 with 1/(2 sqrt 2) folded into its coefficients; 16 VALU
// operations per value (bias add included), same error (2e-5 absolute at the clamp).
__device__ __forceinline__ float gelu_erf_s(float y) {
  constexpr float CP = 4.52548360824585f;   // 3.2 sqrt 2
  const float yc = __builtin_amdgcn_fmed3f(y, -CP, CP);
  const float t = __builtin_fmaf(yc * yc, 0.09765625f, -1.f);
  float p = __builtin_fmaf(t, 8.469007444e-04f, -2.387454268e-03f);
  p = __builtin_fmaf(t, p, 3.280109027e-03f);
  p = __builtin_fmaf(t, p, -5.588355009e-03f);
  p = __builtin_fmaf(t, p, 1.136882324e-02f);
  p = __builtin_fmaf(t, p, -1.921003498e-02f);
  p = __builtin_fmaf(t, p, 2.861942165e-02f);
  p = __builtin_fmaf(t, p, -4.021260887e-02f);
  p = __builtin_fmaf(t, p, 5.456056446e-02f);
  p = __builtin_fmaf(t, p, -7.682786137e-02f);
  p = __builtin_fmaf(t, p, 1.560353935e-01f);
  return y * __builtin_fmaf(yc, p, 0.5f);
}

// ABL: ablations for tools/bench_encode.py --linear-dbg (experiments build only; results wrong):
// 1 = every LDS-DMA piece re-reads one cached KiB, 2 = no LDS-DMA in the loop, 4 = no epilogue,
// 8 = every workgroup stores into one L2-resident window, 16 = no LDS fragment reads, 32 = no MFMAs.
template <int EPI, int WIDE, int ABL>
__global__ void __launch_bounds__(LD_WAVES * 64, 1) RF_NO_PACKED_FP32 k_linear_dma(
    const _Float16* __restrict__ X, const uint4* __restrict__ Wt, const _Float16* __restrict__ bias,
    _Float16* __restrict__ out, int N, const int32_t* __restrict__ m_ptr, float* __restrict__ dbg) {
  constexpr int KS = HID / 16;   // 24
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // diagnostic run only (dbg != nullptr, tools/bench_encode.py --stamps): clock stamps per wave
  const uint64_t ts_entry = dbg ? __builtin_amdgcn_s_memtime() : 0;
  const uint64_t tr_entry = dbg ? __builtin_amdgcn_s_memrealtime() : 0;
  uint64_t ts_loop = 0, t_wait = 0;
  rf_u32x4* slots = (rf_u32x4*)smem_raw;                      // [3][48 * 64]
  rf_u32x4* const dump = slots + LD_SLOTS * LD_FRAGS * 64;    // 1 KiB
  float* bias_l = (float*)(dump + 64);                        // [N] as fp32
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  const int t0 = blockIdx.x * (WIDE ? 2 * LD_TOK : LD_TOK);
  const int tb = WIDE ? wave : (wave & 3);
  const uint32_t n_ph = (uint32_t)N / 64u;
  const uint32_t nblk = (uint32_t)N / 32u;

  struct Pieces {
    const uint4* src;
    rf_u32x4* dst;
    int dstep, sstep;
  };
  auto pieces_of = [&](uint32_t ph) __attribute__((always_inline)) {
    const bool live = ph < n_ph;
    uint32_t b = 2u * ph + (uint32_t)(wave >> 2);
    b = live ? b : nblk - 1u;
    Pieces pc;
    pc.src = Wt + ((size_t)b * KS + (wave & 3) * LD_PW) * 64 + lane;
    pc.dst = live ? slots + ((ph % LD_SLOTS) * LD_FRAGS + wave * LD_PW) * 64 : dump;
    pc.dstep = live ? 64 : 0;
    pc.sstep = 64;
    if (ABL & 1) {
      pc.src = Wt + lane;
      pc.sstep = 0;
    }
    return pc;
  };
  auto issue_piece = [&](const Pieces& pc, int j) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pc.src + j * pc.sstep),
                                     (__attribute__((address_space(3))) void*)(pc.dst + j * pc.dstep), 16, 0, 0);
  };
  // Prologue order = dependency order: the activations (needed by the first MFMA), then the first
  // two phases of the weight stream, and only then the token count -- every read above is legal
  // for any workgroup of the grid (buffers are padded to whole tiles), so the early exit of the
  // workgroups past the packed token count does not have to sit in front of them.
  rf_u32x4 xf[KS];   // this wave's 32 tokens as B-operand fragments (fragment (token block, kk) = 1 KiB)
  {
    const _Float16* xfrag = X + (((size_t)(t0 >> 5) + tb) * KS * 64 + lane) * 8;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) xf[kk] = *(const rf_u32x4*)(xfrag + (size_t)kk * 512);
  }
  {
    const Pieces p0 = pieces_of(0), p1 = pieces_of(1);
#pragma unroll
    for (int j = 0; j < LD_PW; ++j) issue_piece(p0, j);
#pragma unroll
    for (int j = 0; j < LD_PW; ++j) issue_piece(p1, j);
  }
  const int M = *m_ptr;
  if (t0 >= M) {   // whole workgroup; its LDS-DMA pieces must land before the LDS is handed on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  for (int i = tid; i < N; i += LD_WAVES * 64) bias_l[i] = (float)bias[i];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    rf_u32x4 v = xf[kk];
    asm volatile("" : "+a"(v));
    xf[kk] = v;
  }
  __syncthreads();   // bias in LDS (this barrier also drains the first DMA pieces: needed at once anyway)

  // lane's 16 features of a block: 8 g + 4 h + j -> four 16-byte reads of the fp32 bias
  const uint32_t bias_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)bias_l + (uint32_t)h * 16u;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  struct Bias {
    f32x4v q[4];
  };
  auto bias_issue = [&](Bias& bq, uint32_t blk) __attribute__((always_inline)) {
    const uint32_t ba = bias_a + blk * 128u;
    // straight into the accumulator half of the register file: the four quads are the block's accumulator input
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=a"(bq.q[0]) : "v"(ba));
    asm volatile("ds_read_b128 %0, %1 offset:32" : "=a"(bq.q[1]) : "v"(ba));
    asm volatile("ds_read_b128 %0, %1 offset:64" : "=a"(bq.q[2]) : "v"(ba));
    asm volatile("ds_read_b128 %0, %1 offset:96" : "=a"(bq.q[3]) : "v"(ba));
  };
  auto bias_landed = [&](Bias& bq) __attribute__((always_inline)) {   // call after a wait that covers the four reads
    asm volatile("" : "+a"(bq.q[0]), "+a"(bq.q[1]), "+a"(bq.q[2]), "+a"(bq.q[3]));
  };
  // epilogue of one finished block: acc[4 g + j] = Y[token][32 blk + 8 g + 4 h + j], in four quads.  Four
  // values = four independent dependency chains: a dependent v_fma_f32 issues every ~8 cycles, four
  // interleaved chains every ~5.3 (tools/ubench/mfma_valu2.hip).  Stores are 16 bytes per lane: the 16-byte
  // slot (token, 8 features) of the tiled layout is split between lanes c (h = 0) and c + 32 (h = 1), so the
  // packed quads 2 m and 2 m + 1 go through v_permlane32_swap (cdna_hip_programming.md T21) and the wave's
  // 64 lanes then hold fragment 2 blk + m whole, lane-linear -- two global_store_dwordx4 per block instead of
  // four dwordx2 (the stores are issue-bound per instruction, not per byte).
  _Float16* const out_lane_w = out + (((ABL & 8) ? (size_t)tb : ((size_t)(t0 >> 5) + tb)) * (size_t)(N / 16) * 64 + lane) * 8;
  // (the bias is already in: it is the accumulator INPUT of the block's first MFMA -- one VALU operation per
  // output less in a kernel whose waves are bound by their VALU issue)
  auto epi_quad = [&](const f32x16& acc, uint32_t blk, int g, uint2& even) __attribute__((always_inline)) {
    if (ABL & 4) return;
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = acc[4 * g + j];
    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = gelu_erf_s(y[j]);
    }
    const half4 o = {(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
    const uint2 pk = __builtin_bit_cast(uint2, o);
    if (!(g & 1)) {
      even = pk;
      return;
    }
    const auto sx = __builtin_amdgcn_permlane32_swap(even.x, pk.x, false, false);
    const auto sy = __builtin_amdgcn_permlane32_swap(even.y, pk.y, false, false);
    // unconditional: rows >= M of the last tile exist (the workspace is padded to whole tiles) and
    // nobody reads them -- and a fixed 2 stores per block keeps the vmcnt arithmetic exact.
    *(uint4*)(out_lane_w + ((size_t)((ABL & 8) ? (blk & 1u) : blk) * 2 + (g >> 1)) * 512) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
  };
  auto epilogue = [&](const f32x16& acc, uint32_t blk) __attribute__((always_inline)) {
    uint2 even;
#pragma unroll
    for (int g = 0; g < 4; ++g) epi_quad(acc, blk, g, even);
  };
  // matrix part of a phase: this wave's feature block (par) of the slot x its 32 tokens.  In the WIDE
  // form the epilogue of the PREVIOUS block (prev, blk_prev) is cut into the same instruction stream:
  // after every sixth MFMA one quad of outputs (~70 plain VALU operations), fenced with sched_barrier so
  // that the compiler keeps the slices where they are -- the VALU work issues under the MFMAs (of this
  // wave and of the other wave of the SIMD) instead of in a VALU-only stretch between two MFMA stretches.
  auto mfma_part = [&](uint32_t ph, const Pieces& nxt, int par, bool with_dma, f32x16& acc, const f32x16* prev,
                       uint32_t blk_prev) __attribute__((always_inline)) {
    const uint32_t blk_cur = 2u * ph + (uint32_t)par;
    const rf_u32x4* slot = slots + ((ph % LD_SLOTS) * LD_FRAGS + par * KS) * 64 + lane;
    const uint32_t sa = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)slot;
    rf_u32x4 fa[2][WL_GRP];
    constexpr int NG = KS / WL_GRP;   // 6
    Bias bq;
    uint2 even;                           // the even quad's packed outputs, waiting for their odd partner
    bias_issue(bq, blk_cur);              // older than every fragment read of this part: covered by its first wait
    if (!(ABL & 16)) lds_read_group<0>(fa[0], sa);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int g = kk / WL_GRP, j = kk % WL_GRP;
      if (j == 0) {
        if (ABL & 16) {   // no fragment reads (the bias reads still need their wait)
          if (g == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (g + 1 < NG) {
          lds_read_group_dyn(fa[(g + 1) & 1], sa, (g + 1) * WL_GRP);
          lds_wait_group<WL_GRP>(fa[g & 1]);
        } else {
          lds_wait_group<0>(fa[g & 1]);
        }
        if (g == 0) bias_landed(bq);
        if (with_dma && g < LD_PW && !(ABL & 2)) issue_piece(nxt, g);   // one LDS-DMA piece per group of 4 MFMAs
      }
      const half8 a = __builtin_bit_cast(half8, fa[g & 1][j]);
      if (ABL & 32) {   // no MFMAs
        if (kk == 0) asm volatile("" : "=v"(acc));
      } else if (kk == 0) {
        // accumulator input = the block's bias in the accumulator's layout (register 4 q + j = feature 8 q + 4 h + j)
        const f32x16 b0 = {bq.q[0][0], bq.q[0][1], bq.q[0][2], bq.q[0][3], bq.q[1][0], bq.q[1][1], bq.q[1][2], bq.q[1][3],
                           bq.q[2][0], bq.q[2][1], bq.q[2][2], bq.q[2][3], bq.q[3][0], bq.q[3][1], bq.q[3][2], bq.q[3][3]};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, xf[kk]), b0, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, xf[kk]), acc, 0, 0, 0);
      }
      if (prev && kk % 6 == 5) {
        epi_quad(*prev, blk_prev, kk / 6, even);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (dbg) ts_loop = __builtin_amdgcn_s_memtime();
  if (WIDE) {
    // Block b's epilogue runs under block b + 1's MFMAs.  Before the loop there is no finished block:
    // the first part "finishes" an all-zero accumulator into block 0's place, which the real block 0
    // (stored later by the same lanes) overwrites -- no branch in the loop, and the VMEM count per
    // phase (6 pieces + 4 stores) stays what the counted wait below assumes.
    f32x16 acc0, acc1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (uint32_t ph = 0; ph < n_ph; ++ph) {
      uint64_t ts0 = 0;
      if (dbg) ts0 = __builtin_amdgcn_s_memtime();
      // the previous phase issued, after the barrier that follows my pieces of phase ph, exactly 6
      // pieces (phase ph+1) and 4 stores: those may stay in flight, everything older has landed
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LD_PW + 4) : "memory");
      __builtin_amdgcn_s_barrier();
      if (dbg) t_wait += __builtin_amdgcn_s_memtime() - ts0;
      const Pieces nxt = pieces_of(ph + 2);
      mfma_part(ph, nxt, 0, true, acc0, &acc1, ph ? 2u * ph - 1u : 0u);
      mfma_part(ph, nxt, 1, false, acc1, &acc0, 2u * ph);
    }
    epilogue(acc1, nblk - 1u);   // the last block's epilogue has no MFMAs to hide under
  } else {
    f32x16 acc;
    for (uint32_t ph = 0; ph < n_ph; ++ph) {
      uint64_t ts0 = 0;
      if (dbg) ts0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LD_PW + 2) : "memory");
      __builtin_amdgcn_s_barrier();
      if (dbg) t_wait += __builtin_amdgcn_s_memtime() - ts0;
      const Pieces nxt = pieces_of(ph + 2);
      // Tried and dropped: running the two waves of a SIMD (w and w + 4) through the phase in
      // OPPOSITE orders (one's MFMAs over the other's epilogue and DMA issues, pieces issued in a
      // burst): 5 300 cycles per phase instead of 4 300 -- a burst of six LDS-DMA issues costs more
      // than six issues spread over the MFMA groups.
      const int par = wave >> 2;
      mfma_part(ph, nxt, par, true, acc, nullptr, 0u);
      epilogue(acc, 2u * ph + (uint32_t)par);
    }
  }
  if (ABL & 4) asm volatile("" ::"v"(out_lane_w));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dump pieces must land before the LDS is handed on
  if (dbg && lane == 0) {
    float* o = dbg + ((size_t)blockIdx.x * LD_WAVES + wave) * 8;
    const uint64_t te = __builtin_amdgcn_s_memtime();
    o[0] = (float)(te - ts_entry);                                  // cycles, whole kernel
    o[1] = (float)(__builtin_amdgcn_s_memrealtime() - tr_entry);    // 100-MHz ticks, whole kernel
    o[2] = (float)(ts_loop - ts_entry);                             // cycles before the loop
    o[3] = (float)t_wait;                                           // cycles in vmcnt wait + barrier
    o[4] = (float)n_ph;
  }
}

// ---- tiled GEMM with both operands through an LDS-DMA ring (round 2) ----------------------------
// Y[128 tokens, 384 features] per workgroup, any K: the form for the GEMMs whose activations do
// not fit a wave's registers (FFN2: K = 1536) and for the LayerNorm epilogues, which need all 384
// features of a token inside one workgroup.  k_linear (above) loads both operands per wave
// straight from L1/L2 -- 5 KiB of fragment loads per 6 MFMAs per wave, texture-addresser-bound at
// ~1.7 x its MFMA time (round-1 counters: matrix pipe busy 28 % for FFN2, 13 % for the
// out-projection).  Here a K-step of 32 is brought ONCE per workgroup by LDS-DMA -- 4 token blocks
// and 12 feature blocks of two 1-KiB fragments each, 32 KiB, HBM/L2 order == LDS order because the
// fragment image is lane-linear -- into a 4-slot ring with three stages in flight behind a counted
// vmcnt and one raw s_barrier per stage; the 8 waves (2 token halves x 4 feature quarters) read
// their 4 + 6 operand tiles with the permuted, conflict-free ds_read_b128 of the batch-256 sweep
// (scan_wide.hip) and run 24 v_mfma_f32_16x16x32_f16 per stage: 10 LDS reads per 24 MFMAs.
// A = weights (16 features x 32 k), B = activations (16 tokens x 32 k): the accumulator holds the
// TOKEN on the lane and 4 consecutive FEATURES in registers, as in the other GEMMs, so bias, GELU,
// residual add, the LayerNorm statistics and the 8-byte stores into the tiled layout are lane-local
// plus one LDS exchange between the four feature quarters.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GT_TOK 128
#define GT_SLOTS 4
#define GT_STAGE_FRAGS 32                  // 1-KiB fragments per stage: 4 x 2 (activations) + 12 x 2 (weights)
template <int EPI, int KS, int DMODE>      // KS = K / 16 (24 | 96); DMODE: who issues the LDS-DMA pieces (below)
__global__ void __launch_bounds__(512, 1) k_gemm_tile(
    const _Float16* __restrict__ X, const uint4* __restrict__ Wt, const _Float16* __restrict__ bias,
    _Float16* __restrict__ out, int ldo, const int32_t* __restrict__ m_ptr, const _Float16* __restrict__ res,
    const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta, float eps, float* __restrict__ dbg) {
  constexpr int NST = KS / 2;              // stages (k-steps of 32)
  // diagnostic run only (dbg != nullptr, tools/bench_encode.py --stamps --stamp-epi 3 | 4): clock stamps per wave
  const uint64_t ts_entry = dbg ? __builtin_amdgcn_s_memtime() : 0;
  uint64_t ts_loop = 0, ts_epi = 0, t_wait = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  rf_u32x4* slots = (rf_u32x4*)smem_raw;                           // [GT_SLOTS][32 frags][64 lanes]
  rf_u32x4* const dump = slots + GT_SLOTS * GT_STAGE_FRAGS * 64;   // 1 KiB: pieces issued past the last stage
  float* red = (float*)(dump + 64);                               // [2][4 quarters][128 tokens] LayerNorm partial sums
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;                         // token half, feature quarter
  const int t0 = blockIdx.x * GT_TOK;
  const int n0 = blockIdx.y * 384;

  // LDS-DMA issue is stall-prone (60-185 cycles per piece): were every wave to issue its share at the
  // same point of every stage, both waves of a SIMD would stall together and the matrix pipe would idle.
  // The halves of the workgroup (waves 0-3 / 4-7: the two waves of each SIMD) therefore take TURNS: the
  // stage t is brought, three stages ahead, by half (t + 1) & 1 alone -- 8 pieces per wave, every other
  // stage -- so in every stage one wave per SIMD goes straight to its MFMAs.  Piece p = 8 w' + j
  // (w' = wave & 3): p < 8: activation block p >> 1, fragment p & 1; p >= 8: weight block (p - 8) >> 1.
  // A wave's outstanding pieces at the wait in front of stage t are then either {t, t + 2} (its half
  // brought t) or {t + 1}: `vmcnt(8)` is right for both.
  // DMODE 1: EVERY wave brings 4 pieces of every stage, one after each sixth MFMA of the stage before
  // (mfma_stage): 8 pieces in a burst ahead of the MFMAs cost the issuing wave ~1 200 cycles, during which the
  // stage's barrier holds everybody else -- spread, the stalls of a piece sit on top of six queued MFMAs.
  const int half = wave >> 2;
  constexpr int NP = DMODE == 1 ? 4 : 8;    // pieces per wave per issue
  const int p0 = DMODE == 1 ? wave * 4 : (wave & 3) * 8;
  // a piece's source = a wave-uniform base (scalar registers) + the lane's 16 bytes: nothing of it lives in
  // vector registers across the MFMAs it is issued between
  const char* const my_base = (p0 < 8) ? (const char*)X + ((size_t)((t0 >> 5) + (p0 >> 1)) * KS) * 1024
                                       : (const char*)Wt + ((size_t)((n0 >> 5) + ((p0 - 8) >> 1)) * KS) * 1024;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  auto issue_piece = [&](int s, int j) __attribute__((always_inline)) {   // j = 2 b + f: block b of mine, fragment f
    const bool live = s < NST;
    const char* src = my_base + ((size_t)(j >> 1) * KS + (size_t)(live ? 2 * s : 0) + (j & 1)) * 1024;
    rf_u32x4* dst = live ? slots + ((s % GT_SLOTS) * GT_STAGE_FRAGS + p0 + j) * 64 : dump;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane_off),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue_stage = [&](int s) __attribute__((always_inline)) {   // all of my pieces of stage s, live or not (uniform vmcnt arithmetic)
    if (DMODE == 2 && s >= 3) return;   // ablation (experiments build; results wrong): no LDS-DMA in the stage loop
    if (DMODE != 1 && ((s + 1) & 1) != half) return;
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(s, j);
  };
  issue_stage(0);
  issue_stage(1);
  issue_stage(2);
  const int M = *m_ptr;
  if (t0 >= M) {   // whole workgroup; its LDS-DMA pieces must land before the LDS is handed on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  f32x4 acc[6][4];   // [feature tile ft][token tile tt]: lane = token 16 tt + c16, features 16 ft + 4 g + j
#pragma unroll
  for (int ft = 0; ft < 6; ++ft)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[ft][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // operand tile addresses inside a stage: activation block b at fragment 2 b, weight block b at 8 + 2 b;
  // lane's bytes = ((g >> 1) * 64 + (g & 1) * 32 + 16 rg + c16) * 16 (see k_scan_w16)
  const uint32_t lane_a = (uint32_t)(((g >> 1) * 64 + (g & 1) * 32 + c16) * 16);
  const uint32_t xa = lane_a + (uint32_t)((2 * wm) * 2048);             // my first token block
  const uint32_t wa = lane_a + (uint32_t)(8 * 1024 + (3 * wn) * 2048);  // my first feature block
  // Software pipeline by one stage: after barrier(s) the operands of stage s are READ into one register set
  // while the MFMAs of stage s-1 run on the other -- the LDS latency (and the LDS-DMA issues of stage s+3)
  // sit under 24 MFMAs instead of in front of them.
  auto read_stage = [&](int s, rf_u32x4 (&xb)[4], rf_u32x4 (&wf)[6]) __attribute__((always_inline)) {
    const uint32_t sb = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(slots + (s % GT_SLOTS) * GT_STAGE_FRAGS * 64);
    // tile tt of my tokens = block tt >> 1, row group tt & 1; tile ft of my features likewise
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(xb[0]) : "v"(sb + xa));
    asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(xb[1]) : "v"(sb + xa));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(xb[2]) : "v"(sb + xa));
    asm volatile("ds_read_b128 %0, %1 offset:2304" : "=v"(xb[3]) : "v"(sb + xa));
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(wf[0]) : "v"(sb + wa));
    asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(wf[1]) : "v"(sb + wa));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(wf[2]) : "v"(sb + wa));
    asm volatile("ds_read_b128 %0, %1 offset:2304" : "=v"(wf[3]) : "v"(sb + wa));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(wf[4]) : "v"(sb + wa));
    asm volatile("ds_read_b128 %0, %1 offset:4352" : "=v"(wf[5]) : "v"(sb + wa));
  };
  // nxt (DMODE 1): the stage whose pieces this wave issues between the MFMAs
  auto mfma_stage = [&](rf_u32x4 (&xb)[4], rf_u32x4 (&wf)[6], int nxt) __attribute__((always_inline)) {
    if (DMODE != 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ft = 0; ft < 6; ++ft) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        acc[ft][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wf[ft]), __builtin_bit_cast(half8, xb[tt]),
                                                             acc[ft][tt], 0, 0, 0);
      if (DMODE == 1 && ft >= 1 && ft <= 4) {
        __builtin_amdgcn_sched_barrier(0);
        issue_piece(nxt, ft - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (DMODE != 1) __builtin_amdgcn_s_setprio(0);
  };
  auto sync_stage = [&]() __attribute__((always_inline)) {
    uint64_t ts0 = 0;
    if (dbg) ts0 = __builtin_amdgcn_s_memtime();
    // my pieces of the stage have landed (those of the next two stages may stay in flight) ...
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    // ... after the barrier everybody's have, and everybody has READ the previous stage (its slot is free)
    __builtin_amdgcn_s_barrier();
    if (dbg) t_wait += __builtin_amdgcn_s_memtime() - ts0;
  };
  rf_u32x4 xb0[4], wf0[6], xb1[4], wf1[6];
  static_assert(NST % 2 == 0, "stages are processed in pairs");
  if (dbg) ts_loop = __builtin_amdgcn_s_memtime();
  // vmcnt arithmetic of DMODE 1: the pieces of stage t + 3 go out under the MFMAs of stage t - 1, i.e. after the
  // barrier of stage t (the slot they overwrite, that of stage t - 1, has been read by everybody by then); at
  // the wait in front of stage t a wave has therefore issued, after its pieces of stage t, those of t + 1 and
  // t + 2 (4 each): vmcnt(8), the same constant as in the other mode.
  sync_stage();
  read_stage(0, xb0, wf0);
  issue_stage(3);
#pragma unroll 1
  for (int s = 1; s < NST; s += 2) {
    // odd stage s: read into set 1, compute stage s-1 from set 0
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb0[0]), "+v"(xb0[1]), "+v"(xb0[2]), "+v"(xb0[3]), "+v"(wf0[0]), "+v"(wf0[1]),
                 "+v"(wf0[2]), "+v"(wf0[3]), "+v"(wf0[4]), "+v"(wf0[5]));
    sync_stage();
    read_stage(s, xb1, wf1);
    if (DMODE != 1) issue_stage(s + 3);
    mfma_stage(xb0, wf0, s + 3);          // (DMODE 1) under the MFMAs of stage s-1: the pieces of stage s+3
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb1[0]), "+v"(xb1[1]), "+v"(xb1[2]), "+v"(xb1[3]), "+v"(wf1[0]), "+v"(wf1[1]),
                 "+v"(wf1[2]), "+v"(wf1[3]), "+v"(wf1[4]), "+v"(wf1[5]));
    if (s + 1 < NST) {   // even stage s+1: read into set 0, compute stage s from set 1
      sync_stage();
      read_stage(s + 1, xb0, wf0);
      if (DMODE != 1) issue_stage(s + 4);
    }
    mfma_stage(xb1, wf1, s + 4);          // (in the last trip: pieces past the last stage, into the dump slot)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dump pieces must land before the LDS is handed on
  if (dbg) ts_epi = __builtin_amdgcn_s_memtime();

  // ---- epilogue: acc[ft][tt][j] = Y[t0 + 64 wm + 16 tt + c16][n0 + 96 wn + 16 ft + 4 g + j] ----------------
  const int fbase = n0 + 96 * wn + 4 * g;
  float mu[4], rstd[4];
  // every global load of the epilogue -- bias, residual, LayerNorm scale and shift -- is issued HERE, ahead of
  // the first use: spread through the epilogue (the scale and shift behind the LayerNorm's barriers) they were
  // three to four memory round trips in a row, most of the epilogue's 15-17 k cycles (stamps)
  half4 bvs[6], gvs[6], bes[6];
  uint4 rrs[6][2];
#pragma unroll
  for (int ft = 0; ft < 6; ++ft) {
    bvs[ft] = *(const half4*)(bias + fbase + 16 * ft);
    if (EPI == EPI_BIAS_RES_LN) {
      gvs[ft] = *(const half4*)(gamma + fbase + 16 * ft);
      bes[ft] = *(const half4*)(beta + fbase + 16 * ft);
      // residual: ONE 16-byte lane-linear load per pair of token tiles (the fragment of token block
      // 2 wm + tp, feature group 6 wn + ft, whole); the half exchange of the stores below, backwards, follows
#pragma unroll
      for (int tp = 0; tp < 2; ++tp)
        rrs[ft][tp] = *(const uint4*)(res + ((size_t)((t0 >> 5) + 2 * wm + tp) * (HID / 16) + (n0 >> 4) + 6 * wn + ft) * 512 + lane * 8);
    }
  }
#pragma unroll
  for (int ft = 0; ft < 6; ++ft) {
    const half4 bv = bvs[ft];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      if (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          f32x2 y;
          y[0] = acc[ft][tt][j] + (float)bv[j];
          y[1] = acc[ft][tt][j + 1] + (float)bv[j + 1];
          y = gelu_erf2(y);
          acc[ft][tt][j] = y[0];
          acc[ft][tt][j + 1] = y[1];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ft][tt][j] += (float)bv[j];
      }
    }
    if (EPI == EPI_BIAS_RES_LN) {
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        const uint4 r = rrs[ft][tp];
        const auto sx = __builtin_amdgcn_permlane16_swap(r.x, r.z, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(r.y, r.w, false, false);
        const uint2 lo = make_uint2(sx[0], sy[0]), hi = make_uint2(sx[1], sy[1]);
        const half4 r0 = __builtin_bit_cast(half4, lo), r1 = __builtin_bit_cast(half4, hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[ft][2 * tp][j] += (float)r0[j];
          acc[ft][2 * tp + 1][j] += (float)r1[j];
        }
      }
    }
  }
  uint64_t ts_e1 = 0, ts_e2 = 0;
  if (dbg) {
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[5][3]));
    ts_e1 = __builtin_amdgcn_s_memtime();   // bias + residual added: the epilogue's loads have arrived
  }
  if (EPI == EPI_BIAS_RES_LN) {
    // LayerNorm over the 384 features of a token: 24 per lane, x 4 lane groups (g), x 4 feature quarters.  Sum and
    // sum of squares in ONE sweep and ONE exchange through LDS (variance = E[x^2] - mean^2 in fp32: the inputs
    // are residual-stream values of order 1 with |mean| << spread, the cancellation costs ~1e-6 relative): the
    // two-sweep form had a second barrier and a second LDS round trip on every workgroup's critical path.
    // (The cross-lane sums stay __shfl_xor: as v_permlane16/32_swap pair sums they measured 4 600 against 5 000
    // cycles for this block -- and the BUILTIN fed one value twice is folded by hipcc 7.2 into x + x, wrong
    // statistics with no diagnostic; only the inline-asm form is usable for that.)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int ft = 0; ft < 6; ++ft)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[ft][tt][j];
          s1 += v;
          s2 = fmaf(v, v, s2);
        }
      s1 += __shfl_xor(s1, 16);
      s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (g == 0) {
        red[(0 * 4 + wn) * GT_TOK + 64 * wm + 16 * tt + c16] = s1;
        red[(1 * 4 + wn) * GT_TOK + 64 * wm + 16 * tt + c16] = s2;
      }
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int tk = 64 * wm + 16 * tt + c16;
      const float t1 = (red[(0 * 4 + 0) * GT_TOK + tk] + red[(0 * 4 + 1) * GT_TOK + tk]) +
                       (red[(0 * 4 + 2) * GT_TOK + tk] + red[(0 * 4 + 3) * GT_TOK + tk]);
      const float t2 = (red[(1 * 4 + 0) * GT_TOK + tk] + red[(1 * 4 + 1) * GT_TOK + tk]) +
                       (red[(1 * 4 + 2) * GT_TOK + tk] + red[(1 * 4 + 3) * GT_TOK + tk]);
      mu[tt] = t1 * (1.f / HID);
      rstd[tt] = rsqrtf(fmaxf(t2 * (1.f / HID) - mu[tt] * mu[tt], 0.f) + eps);
    }
  }
  if (dbg) {
    asm volatile("" : "+v"(mu[0]), "+v"(rstd[3]));
    ts_e2 = __builtin_amdgcn_s_memtime();   // LayerNorm statistics known
  }
  // Stores: 16 bytes per lane.  A lane holds 4 consecutive features (8 bytes) of each token tile; the 16-byte slot
  // (token, 8 features) of the tiled layout is split between lanes g and g ^ 1.  v_permlane16_swap between the
  // packed values of tile 2 tp (vdst) and tile 2 tp + 1 (src) leaves lanes with g even holding the whole slot of
  // THEIR token of tile 2 tp and lanes with g odd that of tile 2 tp + 1 -- and the wave's 64 slots are then one
  // whole fragment, lane-linear: 12 global_store_dwordx4 per wave instead of 24 dwordx2 (cdna_hip_programming.md
  // T21; here worth little by itself -- the epilogue's 12-13 k cycles per tile are a memory round trip for the
  // residual (4-5 k), the LayerNorm statistics with their shuffles, LDS exchange and barrier (5 k) and the
  // normalise + store pass (3 k): stamps).
#pragma unroll
  for (int ft = 0; ft < 6; ++ft) {
    const half4 gv = gvs[ft], be = bes[ft];
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      uint2 pk[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int tt = 2 * tp + u;
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[ft][tt][j];
          if (EPI == EPI_BIAS_RES_LN) v = (v - mu[tt]) * rstd[tt] * (float)gv[j] + (float)be[j];
          o[j] = (_Float16)v;
        }
        pk[u] = __builtin_bit_cast(uint2, o);
      }
      const auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
      const int token = t0 + 64 * wm + 16 * (2 * tp + (g & 1)) + c16;   // the token whose slot this lane now holds
      if (token < M)
        *(uint4*)(out + ((size_t)((t0 >> 5) + 2 * wm + tp) * (ldo / 16) + (n0 >> 4) + 6 * wn + ft) * 512 + lane * 8) =
            make_uint4(sx[0], sy[0], sx[1], sy[1]);
    }
  }
  if (dbg && lane == 0) {
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (wg < 512) {   // the buffer holds 4096 waves x 8 floats
      float* d = dbg + (wg * 8 + wave) * 8;
      const uint64_t te = __builtin_amdgcn_s_memtime();
      d[0] = (float)(te - ts_entry);        // cycles, whole wave
      d[1] = (float)(ts_loop - ts_entry);   // prologue (first three stages' pieces, first wait)
      d[2] = (float)(ts_epi - ts_loop);     // the stage loop
      d[3] = (float)t_wait;                 // of it: vmcnt wait + barrier
      d[4] = (float)(te - ts_epi);          // epilogue
      d[5] = (float)NST;
      d[6] = (float)(ts_e1 - ts_epi);       // of the epilogue: until bias + residual are in
      d[7] = (float)(ts_e2 - ts_e1);        // ... the LayerNorm statistics (the rest: normalise + store)
    }
  }
}

// LayerNorm of ONE fp32 row [384] -> fp16 tiled activations, by one wave
__device__ __forceinline__ void ln_row(const float* __restrict__ pre, int token, int lane,
                                       const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta,
                                       float eps, _Float16* __restrict__ out) {
  // lane l < 48 owns features 8 l .. 8 l + 7
  float v[8];
  if (lane < 48) {
    const float4 a = *(const float4*)(pre + (size_t)token * HID + lane * 8);
    const float4 b = *(const float4*)(pre + (size_t)token * HID + lane * 8 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  const float mu = wave_sum(s) * (1.f / HID);
  float q = 0.f;
  if (lane < 48) {
#pragma unroll
    for (int j = 0; j < 8; ++j) q += (v[j] - mu) * (v[j] - mu);
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.f / HID) + eps);
  if (lane < 48) {
    const half8 gg = *(const half8*)(gamma + lane * 8);
    const half8 bb = *(const half8*)(beta + lane * 8);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (_Float16)((v[j] - mu) * rstd * (float)gg[j] + (float)bb[j]);
    *(half8*)(out + toff(token, lane * 8, HID / 16)) = o;
  }
}

// ---- small batches (a single query: the reference's only serving mode) -------------------
// With a handful of tokens the GEMMs above are latency chains: one 64/128-token tile means one
// workgroup per 384 (or all N) features streaming 0.3-1.2 MB of weights through ONE CU --
// 9-45 us per call, 620 us per 6-layer forward of a 12-token query.  k_linear_small spreads
// the OUTPUT FEATURES over the chip instead: a workgroup owns 32 features x 64 tokens, its four
// waves split the K range (6 or 24 k-steps each, every load issued before the first MFMA),
// partial sums meet in LDS.  N/32 = 36 / 12 / 48 / 12 workgroups per call, each reading 24-96 KB.
// The LayerNorm epilogues need all 384 features of a token, which no longer meet in one
// workgroup: the GEMM writes bias + residual sums as fp32 rows and k_ln_rows (one wave per
// token) normalises them -- same arithmetic (fp32 statistics of the fp32 sums), one more launch.
enum { EPI_PRE_LN = 3 };   // bias + residual -> fp32 [token][384] row-major scratch
#define SM_TOK 64
template <int EPI, int KS>
__global__ void __launch_bounds__(256) k_linear_small(
    const _Float16* __restrict__ X, const uint4* __restrict__ Wt, const _Float16* __restrict__ bias,
    _Float16* __restrict__ out, float* __restrict__ pre, int N, const int32_t* __restrict__ m_ptr,
    const _Float16* __restrict__ res) {
  constexpr int KW = KS / 4;   // k-steps per wave
  __shared__ float red[4][2][16][64];   // [wave][token block][register][lane]: 32 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int fblk = blockIdx.x;
  const int t0 = blockIdx.y * SM_TOK;
  // every load of this wave's K range goes out before anything else (KW = 6: 18 loads; 24: in 3 rounds of 8)
  const uint4* wsrc = Wt + ((size_t)fblk * KS + wave * KW) * 64 + lane;
  const _Float16* xsrc = X + (((size_t)(t0 >> 5) * KS + wave * KW) * 64 + lane) * 8;
  f32x16 acc[2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[tb][i] = 0.f;
  constexpr int RND = KW < 8 ? KW : 8;
#pragma unroll
  for (int r0 = 0; r0 < KW; r0 += RND) {
    uint4 wf[RND];
    half8 xf[RND][2];
#pragma unroll
    for (int r = 0; r < RND; ++r) {
      wf[r] = wsrc[(size_t)(r0 + r) * 64];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) xf[r][tb] = *(const half8*)(xsrc + ((size_t)tb * KS + r0 + r) * 512);
    }
#pragma unroll
    for (int r = 0; r < RND; ++r)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wf[r]), xf[r][tb], acc[tb], 0, 0, 0);
  }
  const int M = *m_ptr;
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][tb][i][lane] = acc[tb][i];
  __syncthreads();
  // wave w finishes token block w >> 1, registers 8 (w & 1) .. +8 (two groups of 4 features)
  const int tb = wave >> 1;
  const int token = t0 + tb * 32 + c;
#pragma unroll
  for (int gg = 0; gg < 2; ++gg) {
    const int g = 2 * (wave & 1) + gg;
    const int f = fblk * 32 + 8 * g + 4 * h;   // first of the lane's 4 consecutive features
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 4 * g + j;
      v[j] = (red[0][tb][i][lane] + red[1][tb][i][lane]) + (red[2][tb][i][lane] + red[3][tb][i][lane]);
    }
    const half4 bv = *(const half4*)(bias + f);
    if (EPI == EPI_PRE_LN) {
      if (token < M) {
        const half4 rv = *(const half4*)(res + toff(token, f, HID / 16));
        float4 o;
        o.x = v[0] + (float)bv[0] + (float)rv[0];
        o.y = v[1] + (float)bv[1] + (float)rv[1];
        o.z = v[2] + (float)bv[2] + (float)rv[2];
        o.w = v[3] + (float)bv[3] + (float)rv[3];
        *(float4*)(pre + (size_t)token * HID + f) = o;
      }
    } else {
      half4 o;
      if (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          f32x2 y;
          y[0] = v[j] + (float)bv[j];
          y[1] = v[j + 1] + (float)bv[j + 1];
          y = gelu_erf2(y);
          o[j] = (_Float16)y[0];
          o[j + 1] = (_Float16)y[1];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (_Float16)(v[j] + (float)bv[j]);
      }
      if (token < M) *(half4*)(out + toff(token, f, N / 16)) = o;
    }
  }
}

// ONE query (a single sequence of <= 32 tokens: the reference's serving mode): the QKV projection of a head
// and its attention in one launch, one workgroup per head.  The four waves split the K range of the three
// 32-feature blocks (Q, K, V of the head) as k_linear_small does, the partial sums meet in LDS, and wave 0
// finishes them straight into MFMA operands -- no activation leaves the workgroup:
//   Q, K:  A = weights, B = tokens  -> the accumulator holds the TOKEN on the lane and the head's dims in
//          registers; registers 8 s .. 8 s + 7 of both, as fp16, are the operands of S^T = K Q^T for k-step s
//          (the dims come in the accumulator's order in both: a dot product does not care);
//   V:     A = tokens, B = weights  -> the accumulator holds the DIM on the lane and tokens in registers:
//          V^T, whose registers 8 s .. 8 s + 7 are the A operand of O^T = V^T P in the key order of the
//          probability accumulator (16 s + 8 (e >> 2) + 4 h + (e & 3)).
// Same arithmetic as k_linear_small + k_attention_mfma (fp32 sums in the same order, bias, fp16 rounding of
// Q / K / V, fp32 softmax), six launches fewer per forward (~5 us each on the 45-launch dependency chain).
__global__ void __launch_bounds__(256) k_qkv_attn_one(const _Float16* __restrict__ X, const uint4* __restrict__ Wt,
                                                      const _Float16* __restrict__ bias,
                                                      const int32_t* __restrict__ m_ptr, _Float16* __restrict__ ctx) {
  constexpr int KS = HID / 16, KW = KS / 4;   // 24 k-steps, 6 per wave
  typedef float f32x4r __attribute__((ext_vector_type(4)));
  __shared__ f32x4r red[4][3][4][64];         // [wave][q | k | v][register quad][lane]: 48 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int head = blockIdx.x;
  constexpr int NHEADS = HID / HEAD_DIM;
  uint4 wf[3][KW];
  half8 xf[KW];
#pragma unroll
  for (int r = 0; r < KW; ++r) {
#pragma unroll
    for (int blk = 0; blk < 3; ++blk)
      wf[blk][r] = Wt[((size_t)(blk * NHEADS + head) * KS + wave * KW + r) * 64 + lane];
    xf[r] = *(const half8*)(X + ((size_t)(wave * KW + r) * 64 + lane) * 8);
  }
  f32x16 acc[3];
#pragma unroll
  for (int blk = 0; blk < 3; ++blk)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[blk][i] = 0.f;
#pragma unroll
  for (int r = 0; r < KW; ++r) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wf[0][r]), xf[r], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, wf[1][r]), xf[r], acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[r], __builtin_bit_cast(half8, wf[2][r]), acc[2], 0, 0, 0);
  }
#pragma unroll
  for (int blk = 0; blk < 3; ++blk)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      red[wave][blk][q][lane] = f32x4r{acc[blk][4 * q], acc[blk][4 * q + 1], acc[blk][4 * q + 2], acc[blk][4 * q + 3]};
  __syncthreads();
  if (wave != 0) return;
  int n = *m_ptr;
  n = n < 0 ? 0 : (n > 32 ? 32 : n);
  // full sums + bias -> fp16 operands.  Q / K: register 4 q + j = feature 8 q + 4 h + j; V^T: lane = feature c
  half8 qf[2], kf[2], vf[2];
  const float bvt = (float)bias[(2 * NHEADS + head) * HEAD_DIM + c];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4r sq = (red[0][0][q][lane] + red[1][0][q][lane]) + (red[2][0][q][lane] + red[3][0][q][lane]);
    f32x4r sk = (red[0][1][q][lane] + red[1][1][q][lane]) + (red[2][1][q][lane] + red[3][1][q][lane]);
    f32x4r sv = (red[0][2][q][lane] + red[1][2][q][lane]) + (red[2][2][q][lane] + red[3][2][q][lane]);
    const half4 bq = *(const half4*)(bias + head * HEAD_DIM + 8 * q + 4 * h);
    const half4 bk = *(const half4*)(bias + (NHEADS + head) * HEAD_DIM + 8 * q + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      qf[q >> 1][4 * (q & 1) + j] = (_Float16)(sq[j] + (float)bq[j]);
      kf[q >> 1][4 * (q & 1) + j] = (_Float16)(sk[j] + (float)bk[j]);
      // V^T register 4 q + j = token 8 q + 4 h + j: rows beyond the sequence are whatever the activation buffer
      // held (their probabilities are exactly 0 -- but 0 x NaN is not)
      vf[q >> 1][4 * (q & 1) + j] = (8 * q + 4 * h + j < n) ? (_Float16)(sv[j] + bvt) : (_Float16)0.f;
    }
  }
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[0], z, 0, 0, 0);
  sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[1], sc, 0, 0, 0);
  // lane = query c, register i = key 8 (i >> 2) + 4 h + (i & 3)
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if ((i & 3) + 8 * (i >> 2) + 4 * h >= n) sc[i] = -INFINITY;
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) m = fmaxf(m, sc[i]);
  m = fmaxf(m, __shfl_xor(m, 32));
  const float c2 = 0.17677669529663687f * 1.4426950408889634f;   // exp(scale (s - m)) = exp2(c2 s - c2 m)
  const float mc = m * c2;
  float l = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    sc[i] = __builtin_amdgcn_exp2f(fmaf(sc[i], c2, -mc));
    l += sc[i];
  }
  l += __shfl_xor(l, 32);
  f32x16 o = z;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    half8 pb;
#pragma unroll
    for (int j = 0; j < 8; ++j) pb[j] = (_Float16)sc[8 * s + j];
    o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s], pb, o, 0, 0, 0);
  }
  if (c < n) {
    const float inv = 1.f / l;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      half4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (_Float16)(o[4 * g + j] * inv);
      *(half4*)(ctx + toff(c, head * HEAD_DIM + 8 * g + 4 * h, HID / 16)) = t;
    }
  }
}

// LayerNorm of fp32 rows [token][384] -> fp16 tiled activations; one wave per token
// (the LayerNorm of the small-batch path.  Tried and dropped: the GEMM's last-arriving workgroup
// normalising the rows behind agent-scope fences and one atomic -- 250 vs 222 us per 12-token
// encode, the hand-off costs more than the ~5 us dependent launch it removes)
__global__ void __launch_bounds__(256) k_ln_rows(const float* __restrict__ pre, const int32_t* __restrict__ m_ptr,
                                                 const _Float16* __restrict__ gamma,
                                                 const _Float16* __restrict__ beta, float eps,
                                                 _Float16* __restrict__ out) {
  const int token = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (token >= *m_ptr) return;
  ln_row(pre, token, threadIdx.x & 63, gamma, beta, eps, out);
}

// Attention on the matrix cores for sequences of <= 256 tokens (the model's
// max_seq_length): one workgroup per (sequence, head); K rows and V^T in LDS.
//   S^T = K Q^T   A = 32 keys (LDS rows padded to 80 B: conflict-free b128 reads),
//                 B = 32 queries straight from HBM -> the accumulator holds the QUERY on
//                 the lane and KEYS in registers, so the softmax is lane-local plus one
//                 xor-32 shuffle;
//   O^T = V^T P   the probability accumulator IS the B operand of the second product
//                 (registers 8s..8s+7 -> fp16 = k-step s, rows in the order
//                 16s + 8(j>>2) + 4h + (j&3); cdna_hip_programming.md section 3), and V^T
//                 rows (stride T+4 halfs: conflict-free b64 reads) supply the A operand in
//                 the same key order.
__device__ __forceinline__ float att_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
#define ATT_MAX_KB 8
// KB = key blocks of 32 the instantiation holds scores for (2 | 4 | 6 | 8: T <= 64 | 128 | 192 | 256).  The
// scores of a (query block, head) item stay in registers between the two products -- 16 KB registers -- and
// the kernel is latency-bound (10 % matrix pipe busy), so short batches (the ingest buckets are sorted by
// length) take an instantiation with fewer registers and more waves per SIMD instead of the T = 256 one.
template <int KB, int WPS, int NH, int NHALF>   // NH = heads per workgroup (1 | 2); NHALF = groups the key blocks are taken in
__global__ void __launch_bounds__(256, WPS) k_attention_mfma(const _Float16* __restrict__ qkv,
                                                        const int32_t* __restrict__ tok_off,
                                                        _Float16* __restrict__ ctx, float* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // diagnostic run only (dbg != nullptr, tools/bench_encode.py --stamps --stamp-epi 2): clock stamps per wave
  const uint64_t ts_entry = dbg ? __builtin_amdgcn_s_memtime() : 0;
  uint64_t ts_stage = 0, t_qk = 0, t_exp = 0, t_pv = 0, t_out = 0;
  // a workgroup serves NH heads of one sequence: the (head, query block) items are dealt round-robin to the 4
  // waves.  NH = 1 is the product's choice: half the LDS per workgroup lets a third workgroup share the CU at
  // T = 256 (3 waves per SIMD), and the kernel's time is waiting -- staging latency, LDS and MFMA drains -- not
  // issue: +2 % on the whole encoder against NH = 2 (A/B in one process, tools/bench_encode.py --tune att_heads=).
  const int b = blockIdx.x, head0 = blockIdx.y * NH;
  const int r0 = tok_off[b];
  const int n = tok_off[b + 1] - r0;
  if (n <= 0) return;
  const int nkb = (n + 31) >> 5;
  // the LDS image has the INSTANTIATION's shape (TP = 32 KB rows, zero beyond the sequence): compile-time
  // strides, and the K Q^T products below need no per-block branch
  constexpr int TP = KB * 32;
  constexpr int vstride = TP + 4;
  constexpr size_t head_lds = (size_t)TP * 40 + (size_t)32 * vstride;    // halfs per head: K rows, then V^T
  _Float16* lds_h = (_Float16*)lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int n_items = NH * nkb;
  // the item's queries (B operand, straight from HBM) are fetched one item ahead: qn is issued before
  // the current item's products and softmax, so its latency is not on the wave's path
  auto load_q = [&](int hh, int qb, half8 (&q)[2]) __attribute__((always_inline)) {
    const int qrow = qb * 32 + c;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) q[s][e] = (_Float16)0.f;
      if (hh < NH && qrow < n)
        q[s] = *(const half8*)(qkv + toff(r0 + qrow, (head0 + hh) * HEAD_DIM + 16 * s + 8 * h, 3 * HID / 16));
    }
  };
  // items (head of the pair, query block) = hh * nkb + qb, dealt round-robin to the four waves; the pair is
  // stepped without a division
  auto advance = [&](int& hh, int& qb) __attribute__((always_inline)) {
    qb += 4;
    while (qb >= nkb && hh < NH) {
      qb -= nkb;
      ++hh;
    }
  };
  int hh_cur = 0, qb_cur = wave - 4;
  advance(hh_cur, qb_cur);
  half8 qf[2], qn[2];
  load_q(hh_cur, qb_cur, qf);
  {
    // Staging: thread = (16-byte part of a row, row mod 64).  ALL of a thread's global loads are issued
    // before the first LDS write (one row per loop trip, load -> wait -> write, paid the HBM latency up to
    // eight times in a row).
    constexpr int RI = (KB + 1) / 2;   // rows rl + 64 i, i < RI, cover the TP rows (KB = 1: the first 32 of them)
    const int part = tid & 3, rl = tid >> 2;
    half8 kreg[NH][RI], vreg[NH][RI];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const int row = rl + 64 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) { kreg[hh][i][e] = (_Float16)0.f; vreg[hh][i][e] = (_Float16)0.f; }
        if (row < n) {
          kreg[hh][i] = *(const half8*)(qkv + toff(r0 + row, HID + (head0 + hh) * HEAD_DIM + part * 8, 3 * HID / 16));
          vreg[hh][i] = *(const half8*)(qkv + toff(r0 + row, 2 * HID + (head0 + hh) * HEAD_DIM + part * 8, 3 * HID / 16));
        }
      }
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const int row = rl + 64 * i;
        if (row < TP) {
          _Float16* ks_w = lds_h + hh * head_lds;
          _Float16* vt_w = ks_w + (size_t)TP * 40;
          *(half8*)(ks_w + row * 40 + part * 8) = kreg[hh][i];
#pragma unroll
          for (int e = 0; e < 8; ++e) vt_w[(part * 8 + e) * vstride + row] = vreg[hh][i][e];
        }
      }
  }
  __syncthreads();
  if (dbg) ts_stage = __builtin_amdgcn_s_memtime();
  const float scale = 0.17677669529663687f;  // 1 / sqrt(32)
  while (hh_cur < NH) {
    uint64_t ta = 0, tb = 0;
    if (dbg) ta = __builtin_amdgcn_s_memtime();
    const int head = head0 + hh_cur;
    const int qb = qb_cur;
    const _Float16* ks = lds_h + hh_cur * head_lds;           // [TP][40]
    const _Float16* vt = ks + (size_t)TP * 40;               // [32][TP + 4]
    const int q0 = qb * 32;
    int hh_nxt = hh_cur, qb_nxt = qb_cur;
    advance(hh_nxt, qb_nxt);
    load_q(hh_nxt, qb_nxt, qn);
    // The key blocks are taken in NHALF groups of HB (online softmax: running maximum m_run, running row sum
    // l_run, the output accumulator rescaled when the maximum moves).  NHALF = 2 for the long instantiations:
    // the scores of a group, not of the whole sequence, stay in registers -- 64 instead of 128 at T = 256 -- and
    // a fourth workgroup fits the CU's registers (the kernel's time is waiting, not issue: DESIGN.md 4.5).
    // The groups are ALWAYS blocks 0-3 and 4-7 -- whatever instantiation a batch's width selects (the second
    // group of KB = 6 is blocks 4, 5) -- so a sequence's arithmetic does not depend on the batch it is encoded
    // in: the same text gives the same bits wherever it lands (tests/test_config4_gpu.py).
    constexpr int HB = NHALF == 1 ? KB : 4;
    static_assert(NHALF == 1 ? KB <= 4 : (KB > 4 && KB <= 8), "groups of four key blocks");
    const float c2 = scale * 1.4426950408889634f;   // exp(scale * (s - m)) = exp2(c2 * s - c2 * m)
    float m_run = -INFINITY;
    f32x2 l2 = {0.f, 0.f};                          // lane-partial row sum (the two halves of a query's keys meet at the end)
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    int lim = n - 4 * h;                 // the lane's keys 32 kb + 8 (i >> 2) + (i & 3) >= lim are padding
    asm volatile("" : "+v"(lim));        // opaque per item: as a loop invariant the 16 x KB compares were hoisted
                                         // into 128 live lane masks (and spilled)
#pragma unroll
    for (int hf = 0; hf < NHALF; ++hf) {
      const int kb0 = hf * HB;
      if (hf > 0 && kb0 >= nkb) break;   // wave-uniform: the sequence ends before this group
      f32x16 sc[HB];
      // S^T = K Q^T for EVERY key block of the group, back to back (rows beyond the sequence are zero in LDS): no
      // branch between the LDS reads and the products, so the reads are issued ahead and the products pipeline.
      // (Per block behind `kb < nkb` branches, with the running maximum inside, each block was a serial chain
      // read -> 2 MFMAs -> drain -> 8 v_max3 of ~430 cycles: stamps.)  Raw scores stay in the accumulator; the
      // 1/sqrt(32) scale is folded into the exponent below.
#pragma unroll
      for (int kk = 0; kk < HB; ++kk) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // (a block past the instantiation's last -- the second group of KB = 6 -- re-reads the last one: its
        // scores are masked and skipped below like those of any block the sequence does not reach)
        const int kb = kb0 + kk < KB ? kb0 + kk : KB - 1;
        const half8 a0 = *(const half8*)(ks + (kb * 32 + c) * 40 + 8 * h);
        const half8 a1 = *(const half8*)(ks + (kb * 32 + c) * 40 + 16 + 8 * h);
        sc[kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf[0], z, 0, 0, 0);
        sc[kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf[1], sc[kk], 0, 0, 0);
      }
      // att_max3 is inline asm, which hipcc's hazard recognizer does not see reading the MFMA result: the wait
      // states an XDL write needs before a VALU read (11 for the 8-pass 32x32x16) are supplied by hand -- ONE
      // block of nops that takes every accumulator as an in/out operand, so that it cannot be scheduled above
      // any of the products.
      if constexpr (HB == 1) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sc[0]));
      else if constexpr (HB == 2) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sc[0]), "+v"(sc[1]));
      else if constexpr (HB == 3) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]));
      else if constexpr (HB == 4) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]));
      else if constexpr (HB == 6)
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]), "+v"(sc[4]), "+v"(sc[5]));
      else
        asm volatile("s_nop 7\n\ts_nop 7"
                     : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]), "+v"(sc[4]), "+v"(sc[5]), "+v"(sc[6]), "+v"(sc[7]));
      // key >= n -> -inf: only in the block(s) that hold padding, behind a wave-uniform branch
#pragma unroll
      for (int kk = 0; kk < HB; ++kk) {
        const int kb = kb0 + kk;
        if ((kb + 1) * 32 > n) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (kb * 32 + (i & 3) + 8 * (i >> 2) >= lim) sc[kk][i] = -INFINITY;
        }
      }
      // maximum by v_max3_f32 (two new elements per instruction) over the blocks the sequence has
      float m = m_run;
#pragma unroll
      for (int kk = 0; kk < HB; ++kk) {
        if (kb0 + kk < nkb) {
#pragma unroll
          for (int i = 0; i < 16; i += 2) m = att_max3(m, sc[kk][i], sc[kk][i + 1]);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 32));
      if (dbg) {
        asm volatile("" : "+v"(m));
        tb = __builtin_amdgcn_s_memtime();
        t_qk += tb - ta;
        ta = tb;
      }
      const float mc = m * c2;
      if (hf > 0) {   // the maximum may have moved: what was accumulated so far is rescaled (exp2(-inf) = 0 never occurs: m_run is finite here)
        const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run, c2, -mc));
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] *= alpha;
        l2 *= f32x2{alpha, alpha};
      }
      m_run = m;
#pragma unroll
      for (int kk = 0; kk < HB; ++kk) {
        if (kb0 + kk < nkb) {
          // exponent argument and row sum on PAIRS (v_pk_fma_f32 / v_pk_add_f32: two floats per
          // lane at the single rate); v_exp_f32 is the transcendental pipe either way
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            f32x2 e;
            e[0] = sc[kk][i];
            e[1] = sc[kk][i + 1];
            e = __builtin_elementwise_fma(e, f32x2{c2, c2}, f32x2{-mc, -mc});
            f32x2 pr;
            pr[0] = __builtin_amdgcn_exp2f(e[0]);
            pr[1] = __builtin_amdgcn_exp2f(e[1]);
            sc[kk][i] = pr[0];
            sc[kk][i + 1] = pr[1];
            l2 += pr;
          }
        }
      }
      if (dbg) {
        asm volatile("" : "+v"(l2));
        tb = __builtin_amdgcn_s_memtime();
        t_exp += tb - ta;
        ta = tb;
      }
#pragma unroll
      for (int kk = 0; kk < HB; ++kk) {
        const int kb = kb0 + kk;
        if (kb < nkb) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            half8 pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (_Float16)sc[kk][8 * s + j];
            const _Float16* vrow = vt + c * vstride + kb * 32 + 16 * s + 4 * h;  // c = head dim here
            const half4 lo = *(const half4*)vrow;
            const half4 hi = *(const half4*)(vrow + 8);
            half8 a;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = lo[j]; a[4 + j] = hi[j]; }
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb, o, 0, 0, 0);
          }
        }
      }
    }
    float l = l2[0] + l2[1];
    l += __shfl_xor(l, 32);
    if (dbg) {
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(o));
      tb = __builtin_amdgcn_s_memtime();
      t_pv += tb - ta;
      ta = tb;
    }
    // the next item's queries are consumed HERE, ahead of the output stores: the wait the compiler puts in
    // front of their first use then covers the two loads only (issued ~4 000 cycles ago) -- behind the
    // conditional stores it is a vmcnt(0) that also waits for the stores (+2 % on the whole encoder)
    asm volatile("" : "+v"(qn[0]), "+v"(qn[1]));
    if (q0 + c < n) {
      const float inv = 1.f / l;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (_Float16)(o[4 * g + j] * inv);
        *(half4*)(ctx + toff(r0 + q0 + c, head * HEAD_DIM + 8 * g + 4 * h, HID / 16)) = t;
      }
    }
    qf[0] = qn[0];
    qf[1] = qn[1];
    hh_cur = hh_nxt;
    qb_cur = qb_nxt;
    if (dbg) t_out += __builtin_amdgcn_s_memtime() - ta;
  }
  if (dbg && lane == 0) {
    float* d = dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
    if ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) < 1024) {   // the buffer holds 4096 waves x 8 floats
      d[0] = (float)(__builtin_amdgcn_s_memtime() - ts_entry);   // cycles, whole wave
      d[1] = (float)(ts_stage - ts_entry);                       // staging + barrier
      d[2] = (float)t_qk;
      d[3] = (float)t_exp;
      d[4] = (float)t_pv;
      d[5] = (float)t_out;
      d[6] = (float)n_items;
      d[7] = (float)n;
    }
  }
}

// softmax(q k^T / sqrt(32)) v for one (sequence, head): K and V rows in LDS, one
// query per thread, online softmax in fp32.
__global__ void __launch_bounds__(256) k_attention(const _Float16* __restrict__ qkv,
                                                   const int32_t* __restrict__ tok_off,
                                                   _Float16* __restrict__ ctx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int b = blockIdx.x, head = blockIdx.y;
  const int r0 = tok_off[b];
  const int n = tok_off[b + 1] - r0;
  if (n <= 0) return;
  half8* ks = (half8*)lds;            // [n][4]
  half8* vs = ks + (size_t)n * 4;     // [n][4]
  const int tid = threadIdx.x;
  for (int i = tid; i < n * 4; i += 256) {
    const int row = i >> 2, part = i & 3;
    ks[i] = *(const half8*)(qkv + toff(r0 + row, HID + head * HEAD_DIM + part * 8, 3 * HID / 16));
    vs[i] = *(const half8*)(qkv + toff(r0 + row, 2 * HID + head * HEAD_DIM + part * 8, 3 * HID / 16));
  }
  __syncthreads();
  const float scale = 0.17677669529663687f;  // 1 / sqrt(32)
  for (int qi = tid; qi < n; qi += 256) {
    float q[HEAD_DIM];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const half8 t = *(const half8*)(qkv + toff(r0 + qi, head * HEAD_DIM + p * 8, 3 * HID / 16));
#pragma unroll
      for (int j = 0; j < 8; ++j) q[p * 8 + j] = (float)t[j] * scale;
    }
    float m = -INFINITY, l = 0.f;
    float o[HEAD_DIM];
#pragma unroll
    for (int d = 0; d < HEAD_DIM; ++d) o[d] = 0.f;
    for (int j = 0; j < n; ++j) {
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const half8 kv = ks[j * 4 + p];  // same address in every lane: LDS broadcast
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(q[p * 8 + e], (float)kv[e], s);
      }
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn);
      const float pj = __expf(s - mn);
      l = l * corr + pj;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const half8 vv = vs[j * 4 + p];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[p * 8 + e] = fmaf(pj, (float)vv[e], o[p * 8 + e] * corr);
      }
      m = mn;
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      half8 t;
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (_Float16)(o[p * 8 + e] * inv);
      *(half8*)(ctx + toff(r0 + qi, head * HEAD_DIM + p * 8, HID / 16)) = t;
    }
  }
}

// masked mean over the sequence (all packed rows are valid), clamp(count, 1e-9), then
// x / max(||x||, 1e-12).  One 4-wave workgroup per sequence.  The activations are fragment-tiled
// (32 tokens x 16 features = 1 KiB contiguous), so a wave reads whole fragments: wave w takes the
// feature steps kk = w, w + 4, ... of every token block the sequence touches, lane (t = l & 31,
// h = l >> 5) adds the 8 features 16 kk + 8 h .. of token 32 tb + t, and the 32 token lanes meet in
// a shuffle tree.  (Round 1 walked the rows one by one with 2-byte loads: 50 us per 64 k-token
// batch, as long as a GEMM.)
__global__ void __launch_bounds__(256) k_pool_norm(const _Float16* __restrict__ x,
                                                   const int32_t* __restrict__ tok_off,
                                                   _Float16* __restrict__ out16,
                                                   float* __restrict__ out32) {
  __shared__ float pooled[HID];
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int t = lane & 31, h = lane >> 5;
  const int r0 = tok_off[b];
  const int n = tok_off[b + 1] - r0;
  const float inv_cnt = 1.f / fmaxf((float)n, 1e-9f);
  for (int kk = wave; kk < HID / 16; kk += 4) {
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    // lane t sums the sequence's tokens t, t + 32, t + 64 ... (SEQUENCE-relative: the order of the additions, and
    // with it the bits of the result, do not depend on where the sequence sits in the packed stream); eight
    // tokens per lane and trip, every load issued before the first add -- one load per trip of a loop with a
    // divergent body was 48 HBM latencies in a row per wave (34 us per batch)
    for (int i0 = 0; i0 < n; i0 += 256) {
      half8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rel = i0 + 32 * u + t;
        const int tok = r0 + rel;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = (_Float16)0.f;
        if (rel < n) v[u] = *(const half8*)(x + (((size_t)(tok >> 5) * (HID / 16) + kk) * 64 + h * 32 + (tok & 31)) * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += (float)v[u][j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = a[j];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      if (t == 0) pooled[16 * kk + 8 * h + j] = v * inv_cnt;
    }
  }
  __syncthreads();
  float a0 = 0.f, a1 = 0.f;
  if (tid < HID / 2) {
    a0 = pooled[tid * 2];
    a1 = pooled[tid * 2 + 1];
  }
  const float s = wave_sum(a0 * a0 + a1 * a1);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float nrm = fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), 1e-12f);
  if (tid < HID / 2) {
    a0 /= nrm;
    a1 /= nrm;
    if (out16) {
      out16[(size_t)b * HID + tid * 2] = (_Float16)a0;
      out16[(size_t)b * HID + tid * 2 + 1] = (_Float16)a1;
    }
    if (out32) {
      out32[(size_t)b * HID + tid * 2] = a0;
      out32[(size_t)b * HID + tid * 2 + 1] = a1;
    }
  }
}

// ---- forward pass -----------------------------------------------------------------------
// the LayerNorm-fused GEMMs have only 384 output features (one feature group), so
// they take 32-token tiles to put twice as many workgroups on the chip
template <int EPI>
static void launch_linear(const _Float16* X, int K, const uint4* Wt, const _Float16* bias,
                          _Float16* out, int N, int tokens, const int32_t* m_ptr, const _Float16* res,
                          const _Float16* g, const _Float16* b, float eps, float* pre, hipStream_t st) {
  constexpr int NTB = 2;
  const int tiles = (tokens + 32 * NTB - 1) / (32 * NTB);
  if (tokens <= SM_MAX_TOK && rf_knob_linear_small) {
    // small batches: output features spread over the chip (k_linear_small), LayerNorm as its own launch
    const dim3 grid(N / 32, (tokens + SM_TOK - 1) / SM_TOK);
    constexpr int E = (EPI == EPI_BIAS_RES_LN) ? (int)EPI_PRE_LN : (int)EPI;
    if (K == 384)
      hipLaunchKernelGGL((k_linear_small<E, 24>), grid, dim3(256), 0, st, X, Wt, bias, out, pre, N, m_ptr, res);
    else
      hipLaunchKernelGGL((k_linear_small<E, 96>), grid, dim3(256), 0, st, X, Wt, bias, out, pre, N, m_ptr, res);
    if (EPI == EPI_BIAS_RES_LN)
      hipLaunchKernelGGL(k_ln_rows, dim3((tokens + 3) / 4), dim3(256), 0, st, pre, m_ptr, g, b, eps, out);
    return;
  }
  // bit per GEMM of a layer: 1 = FFN2 + LayerNorm (K 1536), 2 = out-projection + LayerNorm, 4 = QKV, 8 = FFN1
  const int gt_bit = (K != 384) ? 1 : (EPI == EPI_BIAS_RES_LN ? 2 : (EPI == EPI_BIAS ? 4 : 8));
  if ((rf_knob_gemm_tile & gt_bit) && tokens >= 8192) {
    const size_t lds = (size_t)GT_SLOTS * GT_STAGE_FRAGS * RF_FRAG_BYTES + RF_FRAG_BYTES + (size_t)2 * 4 * GT_TOK * 4;
    const dim3 grid((tokens + GT_TOK - 1) / GT_TOK, N / 384);
#define RF_GT_LAUNCH(KS_, D_)                                                                                   \
  do {                                                                                                         \
    static rf_lds_attr attr_;                                                                                  \
    (void)rf_ensure_lds(attr_, (const void*)k_gemm_tile<EPI, KS_, D_>, lds);                                   \
    hipLaunchKernelGGL((k_gemm_tile<EPI, KS_, D_>), grid, dim3(512), lds, st, X, Wt, bias, out, N, m_ptr, res, g, b, eps, gt_dbg); \
  } while (0)
    // clock stamps (experiments build): debug_epi 3 = the K = 1536 GEMM (FFN2), 4 = the K = 384 one (out-projection)
    float* const gt_dbg = (rf_knob_debug_epi == (K == 384 ? 4 : 3)) ? (float*)rf_debug_buffer : nullptr;
#ifdef RF_EXPERIMENTS
    if (rf_knob_gemm_tile_dma == 2) {
      if (K == 384) RF_GT_LAUNCH(24, 2);
      else RF_GT_LAUNCH(96, 2);
    } else
#endif
    if (rf_knob_gemm_tile_dma == 1) {
      if (K == 384) RF_GT_LAUNCH(24, 1);
      else RF_GT_LAUNCH(96, 1);
    } else {
      if (K == 384) RF_GT_LAUNCH(24, 0);
      else RF_GT_LAUNCH(96, 0);
    }
#undef RF_GT_LAUNCH
    return;
  }
  if (K == 384 && EPI != EPI_BIAS_RES_LN && rf_knob_linear_dma && tokens >= 8192) {
    const size_t lds = (size_t)LD_SLOTS * LD_FRAGS * RF_FRAG_BYTES + RF_FRAG_BYTES + (size_t)N * 4;   // ring + dump + fp32 bias
    constexpr int E = (EPI == EPI_BIAS_GELU ? EPI_BIAS_GELU : EPI_BIAS);
    // 256-token workgroups once they still fill the chip (>= 256 of them would need 64 k tokens;
    // from ~48 k the halved weight traffic and per-wave overhead outweigh the idle CUs)
    const bool wide = rf_knob_linear_dma == 2 || (rf_knob_linear_dma == 1 && tokens >= 49152);   // 3: never
    float* dbgp = (rf_knob_debug_epi == (int)EPI) ? (float*)rf_debug_buffer : nullptr;
    const dim3 grid_w((tokens + 2 * LD_TOK - 1) / (2 * LD_TOK)), grid_n((tokens + LD_TOK - 1) / LD_TOK), block(LD_WAVES * 64);
#define RF_LD_LAUNCH(W, A)                                                                                     \
  do {                                                                                                         \
    static rf_lds_attr attr_;   /* per instantiation, per device */                                            \
    (void)rf_ensure_lds(attr_, (const void*)k_linear_dma<E, W, A>, lds);                                       \
    hipLaunchKernelGGL((k_linear_dma<E, W, A>), (W) ? grid_w : grid_n, block, lds, st, X, Wt, bias, out, N, m_ptr, dbgp); \
  } while (0)
#ifdef RF_EXPERIMENTS
    if (wide && rf_knob_linear_dbg) {   // ablations of the wide form (results wrong)
      switch (rf_knob_linear_dbg) {
        case 1: RF_LD_LAUNCH(1, 1); return;
        case 2: RF_LD_LAUNCH(1, 2); return;
        case 4: RF_LD_LAUNCH(1, 4); return;
        case 8: RF_LD_LAUNCH(1, 8); return;
        case 12: RF_LD_LAUNCH(1, 12); return;
        case 16: RF_LD_LAUNCH(1, 16); return;
        case 32: RF_LD_LAUNCH(1, 32); return;
        case 48: RF_LD_LAUNCH(1, 48); return;
        case 50: RF_LD_LAUNCH(1, 50); return;
        default: break;
      }
    }
#endif
    if (wide) RF_LD_LAUNCH(1, 0);
    else RF_LD_LAUNCH(0, 0);
#undef RF_LD_LAUNCH
    return;
  }
  if (K == 384 && rf_knob_k384_ntb == 4 && tokens >= 8192)   // 128-token tiles
    hipLaunchKernelGGL((k_linear<EPI, 4, 24>), dim3((tokens + 127) / 128, N / 384), dim3(256), 0, st, X, K, Wt, bias,
                       out, N, m_ptr, res, g, b, eps);
  else if (K == 384)
    hipLaunchKernelGGL((k_linear<EPI, NTB, 24>), dim3(tiles, N / 384), dim3(256), 0, st, X, K, Wt, bias, out,
                       N, m_ptr, res, g, b, eps);
  else if (rf_knob_ffn2_ntb == 4 && tokens >= 8192)   // experiment: 128-token tiles for the K = 1536 GEMM
    hipLaunchKernelGGL((k_linear<EPI, 4, 96>), dim3((tokens + 127) / 128, N / 384), dim3(256), 0, st, X, K, Wt, bias,
                       out, N, m_ptr, res, g, b, eps);
  else   // K == 1536 (checked by rf_encoder_create: intermediate == 4 * hidden is the only other K)
    hipLaunchKernelGGL((k_linear<EPI, NTB, 96>), dim3(tiles, N / 384), dim3(256), 0, st, X, K, Wt, bias, out,
                       N, m_ptr, res, g, b, eps);
}

static int encode_enqueue(const rf_encoder_t* enc, const int32_t* ids_dev, const int32_t* lens_dev, int B, int T,
                          void* out_f16_dev, float* out_f32_dev, void* workspace_dev, hipStream_t st);

// First call with a (shape, buffers) key: plain launches.  Second: the same launch sequence is
// captured on a private stream, instantiated and replayed on the caller's stream; later calls
// replay.  Returns RF_ERR_UNSUPPORTED when the caller should use plain launches.
static int encode_graphed(const rf_encoder_t* enc, const int32_t* ids, const int32_t* lens, int B, int T, void* o16,
                          float* o32, void* wsp, hipStream_t st) {
  std::lock_guard<std::mutex> lock(enc->mu);
  // an exec leaves the cache through `retired`: destroyed once the event behind its last launch has fired
  // (hipGraphExecDestroy on an exec that is still running on another stream is undefined)
  auto retire = [&](rf_encoder::Graph& e) {
    if (e.exec) enc->retired.push_back(e);
    else if (e.done) (void)hipEventDestroy(e.done);
  };
  for (size_t i = 0; i < enc->retired.size();) {
    rf_encoder::Graph& r = enc->retired[i];
    if (!r.done || hipEventQuery(r.done) == hipSuccess) {
      (void)hipGraphExecDestroy(r.exec);
      if (r.done) (void)hipEventDestroy(r.done);
      enc->retired.erase(enc->retired.begin() + i);
    } else {
      ++i;
    }
  }
  (void)hipGetLastError();   // hipEventQuery's hipErrorNotReady is not an error of this call
  // graphs captured under other tuning settings (rf_set_tuning picks kernels) are dropped
  for (size_t i = 0; i < enc->graphs.size();) {
    if (enc->graphs[i].tuning_gen != rf_tuning_generation) {
      retire(enc->graphs[i]);
      enc->graphs.erase(enc->graphs.begin() + i);
    } else {
      ++i;
    }
  }
  rf_encoder::Graph* g = nullptr;
  for (size_t i = 0; i < enc->graphs.size(); ++i) {
    auto& e = enc->graphs[i];
    if (e.B == B && e.T == T && e.ids == ids && e.lens == lens && e.o16 == o16 && e.o32 == (void*)o32 && e.ws == wsp) {
      // least recently used first: a hit moves to the back
      rf_encoder::Graph hit = e;
      enc->graphs.erase(enc->graphs.begin() + i);
      enc->graphs.push_back(hit);
      g = &enc->graphs.back();
      break;
    }
  }
  if (!g) {
    if (enc->graphs.size() >= 32) {   // evict the least recently used
      retire(enc->graphs.front());
      enc->graphs.erase(enc->graphs.begin());
    }
    enc->graphs.push_back(rf_encoder::Graph{B, T, ids, lens, o16, (void*)o32, wsp, rf_tuning_generation, nullptr, false, nullptr});
    return RF_ERR_UNSUPPORTED;
  }
  if (g->dead) return RF_ERR_UNSUPPORTED;
  if (!g->exec) {
    if (!enc->cap_stream && hipStreamCreateWithFlags(&enc->cap_stream, hipStreamNonBlocking) != hipSuccess) {
      g->dead = true;
      return RF_ERR_UNSUPPORTED;
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(enc->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      g->dead = true;
      return RF_ERR_UNSUPPORTED;
    }
    const int rc = encode_enqueue(enc, ids, lens, B, T, o16, o32, wsp, enc->cap_stream);
    const hipError_t e1 = hipStreamEndCapture(enc->cap_stream, &graph);
    hipGraphExec_t exec = nullptr;
    if (rc != RF_OK || e1 != hipSuccess || !graph ||
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      g->dead = true;
      return RF_ERR_UNSUPPORTED;
    }
    (void)hipGraphDestroy(graph);
    g->exec = exec;
  }
  RF_HIP(hipGraphLaunch(g->exec, st));
  if (!g->done && hipEventCreateWithFlags(&g->done, hipEventDisableTiming) != hipSuccess) g->done = nullptr;
  if (g->done) RF_HIP(hipEventRecord(g->done, st));
  return RF_OK;
}

extern "C" int rf_encode(const rf_encoder_t* enc, const int32_t* ids_dev, const int32_t* lens_dev,
                         int B, int T, void* out_f16_dev, float* out_f32_dev, void* workspace_dev,
                         size_t workspace_bytes, void* stream) {
  if (!enc || !ids_dev || !lens_dev || !workspace_dev || (!out_f16_dev && !out_f32_dev)) {
    rf_set_error("rf_encode: null argument");
    return RF_ERR_INVALID;
  }
  if (B <= 0 || T <= 0 || T > enc->cfg.max_position) {
    rf_set_error("rf_encode: B=%d T=%d out of range (max_position %d)", B, T, enc->cfg.max_position);
    return RF_ERR_INVALID;
  }
  if ((size_t)T * 2 * HEAD_DIM * 2 > 160 * 1024) {
    rf_set_error("rf_encode: T=%d does not fit the attention kernel's LDS", T);
    return RF_ERR_UNSUPPORTED;
  }
  if (workspace_bytes < rf_encode_workspace_bytes(enc, B, T) || ((uintptr_t)workspace_dev & 15)) {
    rf_set_error("rf_encode: workspace too small or misaligned");
    return RF_ERR_CAPACITY;
  }
  hipStream_t st = (hipStream_t)stream;
  if ((size_t)B * T <= SM_MAX_TOK && rf_knob_encode_graph) {
    const int rc = encode_graphed(enc, ids_dev, lens_dev, B, T, out_f16_dev, out_f32_dev, workspace_dev, st);
    if (rc != RF_ERR_UNSUPPORTED) return rc;   // RF_ERR_UNSUPPORTED here = "take the plain path"
  }
  return encode_enqueue(enc, ids_dev, lens_dev, B, T, out_f16_dev, out_f32_dev, workspace_dev, st);
}

static int encode_enqueue(const rf_encoder_t* enc, const int32_t* ids_dev, const int32_t* lens_dev, int B, int T,
                          void* out_f16_dev, float* out_f32_dev, void* workspace_dev, hipStream_t st) {
  const rf_encoder_config& c = enc->cfg;
  const rf_encoder_weights& w = enc->w;
  const int I = c.intermediate, L = c.layers;
  EncWs ws;
  enc_carve((unsigned char*)workspace_dev, B, T, I, &ws);
  const int tiles = B * T;   // token slots; launch_linear turns them into tiles
  const int32_t* m_ptr = ws.tok_off + B;

  if (B > 1) hipLaunchKernelGGL(k_tok_offsets, dim3(1), dim3(256), 0, st, lens_dev, B, T, ws.tok_off);   // B == 1: k_embed_ln writes them
  {
    const int chunks = B * ((T + 31) / 32);   // one workgroup per 32 positions of a row
    hipLaunchKernelGGL(k_embed_ln, dim3(chunks), dim3(256), 0, st, ids_dev, lens_dev, ws.tok_off, B, T,
                       c.vocab_size, (const _Float16*)w.word_emb, (const _Float16*)w.pos_emb,
                       (const _Float16*)w.type_emb, (const _Float16*)w.emb_ln_g,
                       (const _Float16*)w.emb_ln_b, c.ln_eps, ws.x);
  }
  const size_t attn_lds = (size_t)T * 2 * HEAD_DIM * 2;
  static rf_lds_attr attn_attr;
  if (T > 32 * ATT_MAX_KB) RF_HIP(rf_ensure_lds(attn_attr, (const void*)k_attention, attn_lds));
  // instantiation by the batch's width: <key blocks, minimum waves per SIMD the register budget must allow, heads
  // per workgroup>
  const int att_kb = T <= 32 ? 1 : (T <= 64 ? 2 : (T <= 128 ? 4 : (T <= 192 ? 6 : 8)));   // a query is one key block
  const int att_nh = rf_knob_att_heads;
  const size_t att_tp = (size_t)att_kb * 32;   // the kernel's LDS image has the instantiation's shape
  const size_t mfma_lds = att_nh * (att_tp * 80 + (size_t)32 * (att_tp + 4) * 2);  // 37 KB per head at T = 256
#define RF_ATT_CASE(KB_, W2, W1, NHF)                                                                                \
  case KB_: {                                                                                                    \
    static rf_lds_attr a2_, a1_;   /* per instantiation, per device */                                           \
    if (att_nh == 2) {                                                                                           \
      RF_HIP(rf_ensure_lds(a2_, (const void*)k_attention_mfma<KB_, W2, 2, NHF>, mfma_lds));                           \
    } else {                                                                                                     \
      RF_HIP(rf_ensure_lds(a1_, (const void*)k_attention_mfma<KB_, W1, 1, NHF>, mfma_lds));                           \
    }                                                                                                            \
  } break;
  if (T <= 32 * ATT_MAX_KB) {
    switch (att_kb) {
      RF_ATT_CASE(1, 4, 6, 1)
      RF_ATT_CASE(2, 4, 6, 1)
      RF_ATT_CASE(4, 3, 4, 1)
      RF_ATT_CASE(6, 3, 4, 2)
      RF_ATT_CASE(8, 3, 4, 2)
      default: break;
    }
  }
#undef RF_ATT_CASE
  float* const att_dbg = (rf_knob_debug_epi == 2) ? (float*)rf_debug_buffer : nullptr;   // clock stamps (experiments build)
  const bool one_query = rf_knob_one_query && B == 1 && T <= 32 && rf_knob_linear_small;
  _Float16* x = ws.x;
  _Float16* y = ws.y;
  for (int l = 0; l < L; ++l) {
    const uint4* qkv_t = enc->qkv_t + (size_t)l * 3 * HID * HID / 8;
    const uint4* ao_t = enc->ao_t + (size_t)l * HID * HID / 8;
    const uint4* ff1_t = enc->ff1_t + (size_t)l * I * HID / 8;
    const uint4* ff2_t = enc->ff2_t + (size_t)l * HID * I / 8;
    if (one_query) {   // a single sequence of <= 32 tokens: QKV projection and attention of a head in one launch
      hipLaunchKernelGGL(k_qkv_attn_one, dim3(c.heads), dim3(256), 0, st, x, qkv_t,
                         (const _Float16*)w.qkv_b + (size_t)l * 3 * HID, m_ptr, ws.ctx);
    } else {
    if (!(l > 0 && rf_knob_post_block && rf_knob_post_qkv && tiles >= 8192))   // else: written by the previous layer's k_post_block
    launch_linear<EPI_BIAS>(x, HID, qkv_t, (const _Float16*)w.qkv_b + (size_t)l * 3 * HID, ws.qkv,
                            3 * HID, tiles, m_ptr, nullptr, nullptr, nullptr, 0.f, ws.pre, st);
    if (T <= 32 * ATT_MAX_KB) {
      const dim3 ag(B, c.heads / att_nh);
#define RF_ATT_CASE(KB_, W2, W1, NHF)                                                                                        \
  case KB_:                                                                                                              \
    if (att_nh == 2)                                                                                                     \
      hipLaunchKernelGGL((k_attention_mfma<KB_, W2, 2, NHF>), ag, dim3(256), mfma_lds, st, ws.qkv, ws.tok_off, ws.ctx, att_dbg); \
    else                                                                                                                 \
      hipLaunchKernelGGL((k_attention_mfma<KB_, W1, 1, NHF>), ag, dim3(256), mfma_lds, st, ws.qkv, ws.tok_off, ws.ctx, att_dbg); \
    break;
      switch (att_kb) {
        RF_ATT_CASE(1, 4, 6, 1)
        RF_ATT_CASE(2, 4, 6, 1)
        RF_ATT_CASE(4, 3, 4, 1)
        RF_ATT_CASE(6, 3, 4, 2)
        RF_ATT_CASE(8, 3, 4, 2)
        default: break;
      }
#undef RF_ATT_CASE
    }
    else
      hipLaunchKernelGGL(k_attention, dim3(B, c.heads), dim3(256), attn_lds, st, ws.qkv, ws.tok_off,
                         ws.ctx);
    }
    if (rf_knob_post_block && tiles >= 8192) {
      // out-projection + LayerNorm, FFN1 + GELU, FFN2 + LayerNorm in one launch (encoder_post.hip): x -> y
      rf_post_args pa;
      pa.ctx = ws.ctx;
      pa.res = x;
      pa.out = y;
      pa.qkv_out = (rf_knob_post_qkv && l + 1 < L) ? ws.qkv : nullptr;   // the next layer's Q | K | V in the same launch
      pa.pack = enc->post_t + (size_t)l * rf_post_pack_elems() / 8;
      pa.eps = c.ln_eps;
      pa.m_ptr = m_ptr;
      pa.dbg = (rf_knob_debug_epi == 5) ? (float*)rf_debug_buffer : nullptr;
      pa.abl = rf_knob_post_dbg;
      const int prc = rf_launch_post_block(pa, tiles, st);
      if (prc != RF_OK) return prc;
      _Float16* t_ = x;
      x = y;
      y = t_;
      continue;
    }
    launch_linear<EPI_BIAS_RES_LN>(ws.ctx, HID, ao_t, (const _Float16*)w.ao_b + (size_t)l * HID, y, HID,
                                   tiles, m_ptr, x, (const _Float16*)w.ln1_g + (size_t)l * HID,
                                   (const _Float16*)w.ln1_b + (size_t)l * HID, c.ln_eps, ws.pre, st);
    launch_linear<EPI_BIAS_GELU>(y, HID, ff1_t, (const _Float16*)w.ff1_b + (size_t)l * I, ws.ff, I,
                                 tiles, m_ptr, nullptr, nullptr, nullptr, 0.f, ws.pre, st);
    launch_linear<EPI_BIAS_RES_LN>(ws.ff, I, ff2_t, (const _Float16*)w.ff2_b + (size_t)l * HID, x, HID,
                                   tiles, m_ptr, y, (const _Float16*)w.ln2_g + (size_t)l * HID,
                                   (const _Float16*)w.ln2_b + (size_t)l * HID, c.ln_eps, ws.pre, st);
  }
  hipLaunchKernelGGL(k_pool_norm, dim3(B), dim3(256), 0, st, x, ws.tok_off, (_Float16*)out_f16_dev,
                     out_f32_dev);
  RF_HIP(hipGetLastError());
  return RF_OK;
}
