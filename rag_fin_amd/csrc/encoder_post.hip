// The post-attention half of an encoder layer in ONE launch (round 3):
//
//   y  = LayerNorm(ctx Wo^T + bo + x)                       out-projection (K4)
//   h  = GELU(y W1^T + b1)                                   FFN1 (K5)
//   x' = LayerNorm(h W2^T + b2 + y)                          FFN2 (K6)
//
// (BertSelfOutput, BertIntermediate, BertOutput of the model behind SentenceTransformer.encode:
// vector_rag_mcp/main.py:50, "chunking_storing (1).py":379-380.)  As three launches the 64 k-token
// batch wrote y (50 MB) and h (201 MB) to HBM and read them back (0.68 GB of the layer's 1.12 GB
// were such round trips), paid three prologues / drains and two LayerNorm exchanges through LDS.
//
// Here a WAVE owns 32 tokens x ALL features, one wave per SIMD (4 waves = 128 tokens per
// workgroup, up to 512 registers per lane):
//   * every GEMM is  acc^T[feature][token] += W-fragment (A, from LDS) x activation-fragment (B,
//     registers): the 32x32 accumulator holds the TOKEN on the lane and 16 features in registers --
//     registers 8 s .. 8 s + 7 of lane (c, h) are the features 32 b + 8 (2 s + (e >> 2)) + 4 h + (e & 3);
//   * converted to fp16 these eight registers ARE a B operand of the next GEMM for a k-step whose
//     sixteen k indices are that permutation of the block's features.  A dot product does not care
//     in which order k runs as long as both operands agree, so the weights of the consuming GEMM are
//     stored once with their k axis permuted the same way (rf_launch_post_stream_build):  y feeds
//     FFN1 and h feeds FFN2 register to register, no LDS exchange, no transposition;
//   * a token's 384 features sit in lanes c and c + 32 of one wave: the LayerNorm statistics are
//     lane-local sums plus one xor-32 exchange -- no LDS, no barrier;
//   * the weights (Wo: 288 KiB, W1 + W2: 2.25 MiB per layer, L2-resident) stream through a 3-slot
//     LDS ring by LDS-DMA, 48 KiB per step, two steps ahead behind a counted vmcnt and one raw
//     s_barrier per step; all four waves read every fragment (one ds_read_b128 per MFMA).
//   * MLP step i = { FFN2 of block i - 2 (24 MFMAs), FFN1 of block i (24 MFMAs) } with the GELU of
//     block i - 1 cut into 136 single operations placed in the MFMA gaps by a cost table (gelu_sched): the slot of step i
//     holds W1[block i] and the W2 columns of block i - 2.
// Registers: acc2 192 + activations 96 + two FFN1 accumulators 32 + two h operands 16 + LDS read
// groups 32 + GELU temporaries ~16.
#include "encoder_internal.h"
#include <type_traits>

#define PB_TOK 128
#define PB_WAVES 4
#define PB_SLOTS 3
#define PB_PW (PB_FRAGS / PB_WAVES)   // LDS-DMA pieces per wave and step (12)
#define PB_STEPS (PB_STEPS_A + PB_STEPS_B)
#define PB_PARAM_FLOATS (4 * HID + 6 * HID)   // b1, then bo, g1, be1, b2, g2, be2
#define PB_LDS_BYTES ((size_t)PB_SLOTS * PB_FRAGS * 1024 + (size_t)PB_PARAM_FRAGS * 1024)

// ---- the per-layer pack ---------------------------------------------------------------------------
// pack[l][fragment F][lane][16 bytes]:
//   F < 16:         parameters as fp32: b1 [1536], bo, g1, be1, b2, g2, be2 [384 each] (15 KiB), then zeros
//   16 <= F < 304:  out-projection weights, standard fragment tiling, in ring order [step S 0..5][k-step 4 S + j, j 0..3][block 0..11]
//   304 <= F:       the MLP stream [step i 0..49][fragment f 0..47], lane = 32 h + r:
//     f < 24  (i < 48):  W1[32 i + r][perm(f, h, e)]                          FFN1 block i, k-step f
//     f >= 24 (i >= 2):  W2[32 ob + r][32 (i - 2) + perm(s, h, e)], ob = (f - 24) >> 1, s = (f - 24) & 1
//   then [QKV of layer l + 1: 18 steps x 48 fragments, k-permuted][its bias as fp32, 8 fragments] (zeros in the last layer)
//   perm(q, h, e) = 32 (q >> 1) + 16 (q & 1) + 8 (e >> 2) + 4 h + (e & 3): the feature that register
//   8 (q & 1) + e of a 32x32 accumulator of block q >> 1 holds in lane half h.  Unused fragments are zero.
struct PackSrc {
  const _Float16 *ao_w, *ff1_w, *ff2_w, *ff1_b, *ao_b, *ln1_g, *ln1_b, *ff2_b, *ln2_g, *ln2_b, *qkv_w, *qkv_b;
};
__global__ void __launch_bounds__(256) k_post_pack_build(const PackSrc w, uint4* __restrict__ pack, int L) {
  constexpr int I = 4 * HID;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)L * PB_PACK_FRAGS * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const int F = (int)((idx >> 6) % PB_PACK_FRAGS);
  const int l = (int)(idx / ((size_t)64 * PB_PACK_FRAGS));
  const int r = lane & 31, h = lane >> 5;
  if (F < 16) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    const int j = (F * 64 + lane) * 4;   // first of the lane's four floats
    if (j < I) {
      const _Float16* p = w.ff1_b + (size_t)l * I + j;
      o = make_float4((float)p[0], (float)p[1], (float)p[2], (float)p[3]);
    } else if (j < PB_PARAM_FLOATS) {
      const int v = (j - I) / HID, k = (j - I) % HID;
      const _Float16* base = v == 0 ? w.ao_b : v == 1 ? w.ln1_g : v == 2 ? w.ln1_b : v == 3 ? w.ff2_b : v == 4 ? w.ln2_g : w.ln2_b;
      const _Float16* p = base + (size_t)l * HID + k;
      o = make_float4((float)p[0], (float)p[1], (float)p[2], (float)p[3]);
    }
    pack[idx] = __builtin_bit_cast(uint4, o);
    return;
  }
  half8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (_Float16)0.f;
  if (F < 16 + PB_STEPS_A * PB_FRAGS) {
    // step S = k-steps 4 S .. 4 S + 3 of ALL 12 feature blocks: fragment (F - 16) % 48 = 12 j + b
    const int S = (F - 16) / PB_FRAGS, q = (F - 16) % PB_FRAGS;
    const int b = q % 12, kk = 4 * S + q / 12;
    const _Float16* src = w.ao_w + ((size_t)l * HID + 32 * b + r) * HID + 16 * kk + 8 * h;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = src[e];
  } else if (F >= 16 + PB_RING_FRAGS) {
    // QKV bias of layer l + 1 as fp32
    float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int j = ((F - 16 - PB_RING_FRAGS) * 64 + lane) * 4;
    if (l + 1 < L && j < 3 * HID) {
      const _Float16* p = w.qkv_b + (size_t)(l + 1) * 3 * HID + j;
      o4 = make_float4((float)p[0], (float)p[1], (float)p[2], (float)p[3]);
    }
    pack[idx] = __builtin_bit_cast(uint4, o4);
    return;
  } else if (F >= 16 + (PB_STEPS_A + PB_STEPS_B) * PB_FRAGS) {
    // QKV weights of layer l + 1, k-permuted (B operand = x' as the accumulator leaves it): step t = blocks 2 t, 2 t + 1,
    // fragment m of the step = block 2 t + (m & 1), k-step m >> 1 (the two accumulation chains alternate)
    const int q = F - 16 - (PB_STEPS_A + PB_STEPS_B) * PB_FRAGS;
    const int t = q / PB_FRAGS, m = q % PB_FRAGS;
    const int blk = 2 * t + (m & 1), ks = m >> 1;
    if (l + 1 < L) {
      const _Float16* src = w.qkv_w + ((size_t)(l + 1) * 3 * HID + 32 * blk + r) * HID + 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = src[e];
        o[4 + e] = src[8 + e];
      }
    }
  } else {
    const int i = (F - 16 - PB_STEPS_A * PB_FRAGS) / PB_FRAGS, f = (F - 16 - PB_STEPS_A * PB_FRAGS) % PB_FRAGS;
    const _Float16* src = nullptr;
    if (f < 24) {
      if (i < 48) src = w.ff1_w + ((size_t)l * I + 32 * i + r) * HID + 32 * (f >> 1) + 16 * (f & 1) + 4 * h;
    } else if (i >= 2) {
      const int m = f - 24;
      src = w.ff2_w + ((size_t)l * HID + 32 * (m >> 1) + r) * I + 32 * (i - 2) + 16 * (m & 1) + 4 * h;
    }
    if (src) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = src[e];
        o[4 + e] = src[8 + e];
      }
    }
  }
  pack[idx] = __builtin_bit_cast(uint4, o);
}

void rf_launch_post_pack_build(const rf_encoder_weights* w, void* pack, int L, hipStream_t st) {
  const size_t total = (size_t)L * PB_PACK_FRAGS * 64;
  PackSrc ps;
  ps.ao_w = (const _Float16*)w->ao_w;
  ps.ff1_w = (const _Float16*)w->ff1_w;
  ps.ff2_w = (const _Float16*)w->ff2_w;
  ps.ff1_b = (const _Float16*)w->ff1_b;
  ps.ao_b = (const _Float16*)w->ao_b;
  ps.ln1_g = (const _Float16*)w->ln1_g;
  ps.ln1_b = (const _Float16*)w->ln1_b;
  ps.ff2_b = (const _Float16*)w->ff2_b;
  ps.ln2_g = (const _Float16*)w->ln2_g;
  ps.ln2_b = (const _Float16*)w->ln2_b;
  ps.qkv_w = (const _Float16*)w->qkv_w;
  ps.qkv_b = (const _Float16*)w->qkv_b;
  hipLaunchKernelGGL(k_post_pack_build, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ps, (uint4*)pack, L);
}

// ---- GELU of one FFN1 block (16 values per lane), cut into 136 slots -------------------------------
// gelu(y) = y Phi(y) = max(y, 0) - |y| Phi(-|y|), and log2 Phi(-t) is so smooth that a degree-5 polynomial in
// t = |y| (weighted minimax fit on [0, 6], tools/fit_gelu.py) gives |error| <= 6.4e-7 over |y| <= 40 -- thirty
// times closer to the exact erf form than the degree-10 erf polynomial of the other GEMM paths (1.9e-5) -- in
// 7 plain operations + 1 v_exp_f32 per value instead of 15: the kernel's waves are bound by their VALU issue.
// The leading coefficient is negative, so beyond the fitted range the exponent runs to -inf, 2^P to 0 and the
// value to max(y, 0): no clamp.  |y| and -|y| are source modifiers (free).  Plain fp32 only (packed fp32 does
// not issue under MFMAs).  Slot O: quad O / 34 (accumulator registers 4 q .. 4 q + 3, four independent chains);
// within it 5 Horner stages x 4 values, the exponentials alternating with the (independent) max(y, 0), the
// final fma, then two fp16 pair conversions.
#define GELU_SLOTS 136
struct GeluTmp {
  float p[4], r[4];
};
__device__ constexpr float kGeluP[6] = {-1.000037670135498f, -1.1507878303527832f, -0.45999258756637573f,
                                        -0.051827218383550644f, 0.007084481883794069f, -0.0004732970555778593f};
template <int O>
__device__ __forceinline__ void gelu_slot(const f32x16& y, GeluTmp& g, uint32_t (&hw)[8]) {
  constexpr int quad = O / 34, w = O % 34;
  if constexpr (w < 20) {
    constexpr int st = w >> 2, v = w & 3;
    const float t = __builtin_fabsf(y[4 * quad + v]);
    if constexpr (st == 0) g.p[v] = __builtin_fmaf(kGeluP[5], t, kGeluP[4]);
    else g.p[v] = __builtin_fmaf(g.p[v], t, kGeluP[4 - st]);
  } else if constexpr (w < 28) {
    constexpr int v = (w - 20) >> 1;
    if constexpr (((w - 20) & 1) == 0) g.p[v] = __builtin_amdgcn_exp2f(g.p[v]);
    else asm("v_max_f32 %0, 0, %1" : "=v"(g.r[v]) : "v"(y[4 * quad + v]));   // max(y, 0) in ONE operation (fmaxf and fmed3f(y, 0, inf) both compile to a canonicalising v_max + the v_max)
  } else if constexpr (w < 32) {
    constexpr int v = w - 28;
    g.p[v] = __builtin_fmaf(-__builtin_fabsf(y[4 * quad + v]), g.p[v], g.r[v]);
  } else {
    constexpr int pr = w - 32;
    const half2v o = {(_Float16)g.p[2 * pr], (_Float16)g.p[2 * pr + 1]};
    hw[2 * quad + pr] = __builtin_bit_cast(uint32_t, o);
  }
}

// Which GELU slots ride in which MFMA gap of a full MLP step (48 gaps).  A gap hides about 24 cycles of issue beside
// its MFMA (MI355X_MICROARCH.md, "single-issue instructions hidden per gap"), and the gaps are not alike: an even gap
// also issues the two LDS reads of the pair three ahead (R cycles), every fourth one an LDS-DMA piece (s_mov m0 +
// s_nop + the load: D cycles).  A v_exp_f32 slot costs 8 cycles, every other slot 4.  The slots keep their order;
// the table gives each gap the slots that fit b - (its fixed cost), with the smallest b for which all 136 are placed
// -- the spread that overflows no gap if any does not.  (Uniform 2.83 slots per gap: the LDS-DMA gaps ran over.)
struct GeluSched {
  int start[49];
};
__host__ __device__ constexpr int gelu_slot_cost(int O) {
  const int w = O % 34;
  return (w >= 20 && w < 28 && ((w - 20) & 1) == 0) ? 8 : 4;
}
__host__ __device__ constexpr GeluSched gelu_sched(int R, int D) {
  GeluSched s{};
  for (int b = 4; b < 4096; b += 2) {
    int o = 0;
    for (int n = 0; n < 48; ++n) {
      s.start[n] = o;
      const int fixed = ((n & 1) == 0 ? R : 0) + ((n & 3) == 3 ? D : 0);
      int used = 0;
      while (o < GELU_SLOTS && used + gelu_slot_cost(o) <= b - fixed) used += gelu_slot_cost(o++);
    }
    s.start[48] = GELU_SLOTS;
    if (o == GELU_SLOTS) return s;
  }
  for (int n = 0; n <= 48; ++n) s.start[n] = n * GELU_SLOTS / 48;   // (not reached)
  return s;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

// A 16-byte global load the compiler does not count: it sits between LDS-DMA pieces (inline asm, invisible to hipcc),
// and hipcc's own vmcnt for a plain load -- "all but my N younger loads" -- would count the younger PIECES as
// loads and wait for the ring (stamps: the 24 residual MFMAs took 2 700 cycles waiting for pieces issued moments
// before).  The destination is valid only after a hand-placed s_waitcnt that covers it (the counted waits of the
// ring steps do: they retire everything older than the previous step's issues).
__device__ __forceinline__ void gload16(rf_u32x4& d, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}

// DBG: clock stamps per wave into a.dbg.  ABL (experiments build; results wrong): 1 = no LDS-DMA in the MLP
// steps, 2 = no GELU, 8 = no MFMAs in the MLP steps, 4 = QKV phase on accumulator-half (builtin) MFMAs, 16 = QKV stores into an L2-resident window.
template <int DBG, int ABL>
__global__ void __launch_bounds__(PB_WAVES * 64, 1) RF_NO_PACKED_FP32 k_post_block(const rf_post_args a) {
  constexpr int KS = HID / 16;   // 24
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  rf_u32x4* const slots = (rf_u32x4*)smem_raw;                           // [3][48 * 64]
  float* const b1_l = (float*)(slots + PB_SLOTS * PB_FRAGS * 64);        // parameter block: FFN1 bias [1536] ...
  const float* const pvf = b1_l + 4 * HID;                               // ... bo, g1, be1, b2, g2, be2 [6][384]
  const uint64_t ts_entry = DBG ? __builtin_amdgcn_s_memtime() : 0;
  const uint64_t tr_entry = DBG ? __builtin_amdgcn_s_memrealtime() : 0;
  uint64_t ts_a = 0, ts_ln1 = 0, ts_b = 0, ts_ln2 = 0, t_wait = 0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int t0 = blockIdx.x * PB_TOK;
  const size_t tb = (size_t)(t0 >> 5) + wave;   // this wave's token block

  // Ring step S (0..5: two out-projection feature blocks; 6 + i: step i of the MLP stream) = 48 contiguous
  // fragments of the pack; piece p of wave w = fragment 12 w + p.  The source is a wave-uniform pointer
  // (scalar registers) + the lane's 16 bytes.  Pieces past the last step are issued all the same (the counted
  // waits assume 12 per step): they re-read step 0 into a slot nobody reads any more.
  const int n_steps = a.qkv_out ? PB_STEPS + PB_STEPS_C : PB_STEPS;   // with or without the next layer's QKV projection
  const char* const ring_src = (const char*)a.pack + ((size_t)16 + (size_t)wave * PB_PW) * 1024;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  // (inline asm: hipcc's builtin took the source as a per-lane 64-bit address -- one v_lshl_add_u64 per piece in
  // a kernel bound by its VALU issue -- and every compiler-visible LDS read after it drew an s_waitcnt vmcnt(0).
  // All LDS-DMA of this kernel is asm, so the compiler never holds anything in M0 across these statements.)
  // One wave per SIMD issues one instruction per 4-cycle slot, scalar ones included.  The instruction's immediate
  // offset is added to BOTH addresses (global source and LDS destination), so a group of four pieces shares one
  // source base and one M0: s_mov m0, s_nop, load -- three instructions per piece where s_add, s_add / s_addc,
  // s_mov, s_nop, load were six.  (M0 is still written for every piece: nothing here relies on the compiler
  // leaving it alone between two asm statements.)
  auto issue_piece = [&](const char* src_step, uint32_t dst_step, auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    const char* const base = src_step + (size_t)(p >> 2) * 4096;
    const uint32_t dbase = dst_step + (uint32_t)(p >> 2) * 4096u;
    const uint32_t lo = lane_off;   // (a generic lambda does not capture a variable named only in an asm operand)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
                 :: "s"(dbase), "v"(lo), "s"(base), "n"((p & 3) * 1024) : "memory");
  };
  using PC0 = std::integral_constant<int, 0>;
  auto step_src = [&](int S) __attribute__((always_inline)) {
    return ring_src + (size_t)(S < n_steps ? S : 0) * (PB_FRAGS * 1024);
  };
  const uint32_t slots_s = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)slots;   // LDS byte address
  auto step_dst = [&](int S) __attribute__((always_inline)) {
    return slots_s + (uint32_t)(((S % PB_SLOTS) * PB_FRAGS + wave * PB_PW) * 1024);
  };

  // Prologue in dependency order: the wave's attention output (first MFMA), the parameter block and the first
  // two ring steps, then the token count (every read above is legal for any workgroup: buffers are padded).
  rf_u32x4 xf[KS];   // B-operand fragments: first ctx, then y
  const _Float16* const csrc = a.ctx + (tb * KS * 64 + lane) * 8;
  const _Float16* const rsrc = a.res + (tb * KS * 64 + lane) * 8;
  // only the k-steps of the first two ring steps here; the rest of the tile (and the residual) is fetched a few
  // loads at a time under the out-projection's MFMAs: 24 loads in a row stalled the issuing wave for ~4 000 cycles
  // (every CU asks for its 96 KB at the same moment: stamps)
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) xf[kk] = *(const rf_u32x4*)(csrc + (size_t)kk * 512);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int f = wave + 4 * q;   // parameter fragment (15 of them)
    if (f < PB_PARAM_FRAGS)
      issue_piece((const char*)a.pack + (size_t)f * 1024, slots_s + (uint32_t)((PB_SLOTS * PB_FRAGS + f) * 1024), PC0{});
  }
  static_for<0, PB_PW>([&](auto pc) __attribute__((always_inline)) { issue_piece(step_src(0), step_dst(0), pc); });
  static_for<0, PB_PW>([&](auto pc) __attribute__((always_inline)) { issue_piece(step_src(1), step_dst(1), pc); });
  const int M = *a.m_ptr;
  if (t0 >= M) {   // whole workgroup; its LDS-DMA pieces must land before the LDS is handed on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my parameter pieces (and the first ring pieces: needed at once anyway)
  __syncthreads();   // parameters in LDS

  // lane's features of a 32-feature block b: 32 b + 8 g + 4 h + j  (register 4 g + j)
  auto param4 = [&](int which, int b, int g) __attribute__((always_inline)) {
    return *(const f32x4v*)(pvf + which * HID + 32 * b + 8 * g + 4 * h);
  };
  auto load_vec = [&](int which, f32x16 (&dst)[12]) __attribute__((always_inline)) {
    static_for<0, 12>([&](auto Bc) __attribute__((always_inline)) {
      constexpr int b = decltype(Bc)::value;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4v v = param4(which, b, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[b][4 * g + j] = v[j];
      }
    });
  };
  const uint32_t slots_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)slots + lane_off;
  // extra = vector-memory operations the previous step issued besides its 12 ring pieces (plain loads)
  auto sync_step = [&](auto extra_c) __attribute__((always_inline)) {
    constexpr int EXTRA = decltype(extra_c)::value;
    uint64_t ts0 = 0;
    if (DBG) ts0 = __builtin_amdgcn_s_memtime();
    // my pieces of this step have landed (what the previous step issued may stay in flight) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB_PW + EXTRA) : "memory");
    // ... after the barrier everybody's have, and everybody has read the previous step (its slot is free)
    __builtin_amdgcn_s_barrier();
    if (DBG) t_wait += __builtin_amdgcn_s_memtime() - ts0;
  };
  // The residual of a LayerNorm is added on the matrix pipe: acc += I x^T with an identity fragment as the A
  // operand -- exact (1.0 x value, fp32 accumulate), and the residual arrives as the B-operand fragments it is
  // stored as: no half exchange between lanes, no conversions, no adds (24 MFMAs instead of ~430 VALU operations
  // in a kernel whose waves are bound by their VALU issue).  std: k = 16 s + 8 h + e (the tiled activations);
  // perm: k = 16 s + 8 (e >> 2) + 4 h + (e & 3) (y as the accumulator left it).
  half8 id_std[2], id_perm[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      id_std[s][e] = (r32 == 16 * s + 8 * h + e) ? (_Float16)1.f : (_Float16)0.f;
      id_perm[s][e] = (r32 == 16 * s + 8 * (e >> 2) + 4 * h + (e & 3)) ? (_Float16)1.f : (_Float16)0.f;
    }

  // ---- phase A: out-projection ------------------------------------------------------------------------
  f32x16 acc[12];   // [feature block]: acc_o, later acc2
  load_vec(0, acc);  // accumulator input = bias
  rf_u32x4 xr[KS];   // the layer input (residual)
  if (DBG) ts_a = __builtin_amdgcn_s_memtime();
  uint64_t ts_step[PB_STEPS_A + 1];
  // Step S = k-steps 4 S .. 4 S + 3 of all 12 feature blocks (48 MFMAs on 12 independent accumulation chains; fragment
  // m of the slot = k-step m / 12, block m % 12), so the tile's k-steps are needed four per step.  Eight loads (gload16)
  // ride in every step but the last, one after every sixth MFMA: steps 0, 1 the k-steps 8..23 of ctx, steps 2-4 the
  // residual.  (They are older than the ring pieces issued after them: the counted waits below let the previous
  // step's 12 pieces AND its 8 loads stay in flight.)
  static_for<0, PB_STEPS_A>([&](auto Sc) __attribute__((always_inline)) {
    constexpr int S = decltype(Sc)::value;
    // LPS loads per step, one after every (48 / LPS)-th MFMA, until the 40 are out.  (12 / 16 per step only move the time
    // into the first steps -- 3 240 3 820 4 044 2 740 1 760 1 696 and 4 084 4 568 3 344 1 748 1 696 1 696 cycles against
    // 2 032 2 672 2 488 2 576 2 632 1 828: the phase waits for HBM either way, and a step without loads is 1 696.)
    constexpr int LPS = 8;
    constexpr int prev_loads = S == 0 ? 0 : (40 - LPS * (S - 1) < 0 ? 0 : (40 - LPS * (S - 1) < LPS ? 40 - LPS * (S - 1) : LPS));
    sync_step(std::integral_constant<int, prev_loads>{});
    const uint32_t sa = slots_a + (uint32_t)((S % PB_SLOTS) * PB_FRAGS * 1024);
    const char* const nsrc = step_src(S + 2);
    const uint32_t ndst = step_dst(S + 2);
    run_step<PB_FRAGS>([](int m) constexpr { return m; }, sa, [&](auto Mc, const rf_u32x4& af) __attribute__((always_inline)) {
      constexpr int m = decltype(Mc)::value, fb = m % 12, kk = 4 * S + m / 12;
      acc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, af), __builtin_bit_cast(half8, xf[kk]), acc[fb], 0, 0, 0);
      if constexpr ((m & 3) == 3) issue_piece(nsrc, ndst, std::integral_constant<int, (m >> 2)>{});
      if constexpr (m % (48 / LPS) == 48 / LPS - 1 && LPS * S + m / (48 / LPS) < 40) {
        constexpr int q = LPS * S + m / (48 / LPS);   // 0..39
        if constexpr (q < 16) gload16(xf[8 + q], csrc + (size_t)(8 + q) * 512);
        else gload16(xr[q - 16], rsrc + (size_t)(q - 16) * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (DBG) ts_step[S] = __builtin_amdgcn_s_memtime();
  });
  // step 4's loads (the last third of the residual) may still be in flight: everything but step 5's 12 pieces has landed
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]) : "n"(PB_PW));
  asm volatile("" : "+v"(xr[8]), "+v"(xr[9]), "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]), "+v"(xr[15]));
  asm volatile("" : "+v"(xr[16]), "+v"(xr[17]), "+v"(xr[18]), "+v"(xr[19]), "+v"(xr[20]), "+v"(xr[21]), "+v"(xr[22]), "+v"(xr[23]));
  static_for<0, KS>([&](auto Kc) __attribute__((always_inline)) {   // + x
    constexpr int kk = decltype(Kc)::value;
    acc[kk >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(id_std[kk & 1], __builtin_bit_cast(half8, xr[kk]), acc[kk >> 1], 0, 0, 0);
  });
  if (DBG) {
    asm volatile("" : "+v"(acc[11]));
    ts_ln1 = __builtin_amdgcn_s_memtime();
  }

  // ---- LayerNorm 1 over v = acc; y -> fp16 -> the B operands of FFN1 ---------------------------------
  const float inv_h = 1.f / HID;
  auto ln_stats = [&](float& sc, float& sh) __attribute__((always_inline)) {
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};   // four chains each
    static_for<0, 12>([&](auto Fc) __attribute__((always_inline)) {
      constexpr int fb = decltype(Fc)::value;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s1[i & 3] += acc[fb][i];
        s2[i & 3] = fmaf(acc[fb][i], acc[fb][i], s2[i & 3]);
      }
    });
    float t1 = (s1[0] + s1[1]) + (s1[2] + s1[3]), t2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    t1 += __shfl_xor(t1, 32);
    t2 += __shfl_xor(t2, 32);
    const float mu = t1 * inv_h;
    // E[v^2] - mu^2 in fp32: residual-stream values of order 1 with |mu| << spread (as k_gemm_tile)
    sc = rsqrtf(fmaxf(t2 * inv_h - mu * mu, 0.f) + a.eps);
    sh = -mu * sc;
  };
  {
    float sc, sh;
    ln_stats(sc, sh);
    static_for<0, 12>([&](auto Fc) __attribute__((always_inline)) {
      constexpr int fb = decltype(Fc)::value;
      uint32_t w[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4v gv = param4(1, fb, g), be = param4(2, fb, g);
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = fmaf(fmaf(acc[fb][4 * g + j], sc, sh), gv[j], be[j]);
        const half2v lo = {(_Float16)y[0], (_Float16)y[1]}, hi = {(_Float16)y[2], (_Float16)y[3]};
        w[2 * g] = __builtin_bit_cast(uint32_t, lo);
        w[2 * g + 1] = __builtin_bit_cast(uint32_t, hi);
      }
      rf_u32x4 f0 = {w[0], w[1], w[2], w[3]}, f1 = {w[4], w[5], w[6], w[7]};
      xf[2 * fb] = f0;
      xf[2 * fb + 1] = f1;
    });
  }
  load_vec(3, acc);   // acc2: accumulator input = FFN2 bias
  if (DBG) {
    asm volatile("" : "+v"(acc[11]));
    ts_b = __builtin_amdgcn_s_memtime();
  }

  // ---- phase B: the MLP -------------------------------------------------------------------------------
  f32x16 acc1[2];
  uint32_t hw[2][8];   // h of a block as two B operands (k-steps): words 4 s .. 4 s + 3
#pragma unroll
  for (int i = 0; i < 8; ++i) hw[0][i] = hw[1][i] = 0u;
  GeluTmp gt;
  const uint32_t bias_a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)b1_l + (uint32_t)h * 16u;
  // step i: FFN2 of block i - 2 (F2), FFN1 of block i (F1), GELU of block i - 1 (GE); PAR = i & 1
  // extra_c: what the previous step issued besides its 12 pieces; qb_c: this step also brings the next layer's QKV bias
  // into the (now idle) FFN1-bias block -- every wave two pieces (`wave` and 4: the fifth is written four times over)
  auto mlp_step = [&](auto f2c, auto f1c, auto gec, auto parc, int i, auto extra_c, auto qb_c) __attribute__((always_inline)) {
    constexpr bool F2 = decltype(f2c)::value, F1 = decltype(f1c)::value, GE = decltype(gec)::value && !(ABL & 2);
    constexpr int PAR = decltype(parc)::value;
    constexpr int NM = (F2 ? 24 : 0) + (F1 ? 24 : 0);    // MFMAs (= gaps) of the step
    const int S = PB_STEPS_A + i;
    sync_step(extra_c);
    if constexpr (decltype(qb_c)::value) {
      const char* const qb = (const char*)a.pack + ((size_t)16 + PB_RING_FRAGS) * 1024;
      const uint32_t b1_s = slots_s + (uint32_t)(PB_SLOTS * PB_FRAGS * 1024);
      issue_piece(qb + (size_t)wave * 1024, b1_s + (uint32_t)wave * 1024u, PC0{});
      issue_piece(qb + 4 * 1024, b1_s + 4u * 1024u, PC0{});
    }
    const uint32_t sa = slots_a + (uint32_t)((S % PB_SLOTS) * PB_FRAGS * 1024);
    const char* const nsrc = step_src(S + 2);
    const uint32_t ndst = step_dst(S + 2);
    f32x4v bq[4];
    if constexpr (F1) {   // FFN1 bias of block i: the accumulator input; older than every fragment read of the step
      const uint32_t ba = bias_a + (uint32_t)i * 128u;
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(bq[0]) : "v"(ba));
      asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(bq[1]) : "v"(ba));
      asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(bq[2]) : "v"(ba));
      asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(bq[3]) : "v"(ba));
    }
    // MFMA gap n of the step -> its A fragment.  In a full step the two GEMMs ALTERNATE (n even: FFN2 number n / 2,
    // n odd: FFN1 k-step n / 2), so that a pair of LDS reads feeds one MFMA of each accumulation chain.
    auto frag_of = [](int n) constexpr { return (F2 && F1) ? ((n & 1) ? (n >> 1) : 24 + (n >> 1)) : (F2 ? 24 + n : n); };
    // (Probes of round 3, taken out again: pairs read four / five ahead instead of three 2 117 / 2 172 cycles per step against
    // 2 107; no wait for the reads at all 2 083; FFN1 on two accumulation chains 2 113 -- neither the LDS latency nor the
    // FFN1 chain is what the step waits for.)
    run_step<NM>(frag_of, sa, [&](auto Nc, const rf_u32x4& afr) __attribute__((always_inline)) {
      constexpr int n = decltype(Nc)::value;             // MFMA gap of the step
      constexpr int fr = frag_of(n);
      if constexpr (F1 && n == 0) asm volatile("" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));   // landed: older than the first pair
      if constexpr (ABL & 8) {
        if constexpr (F1 && n == NM - 1) asm volatile("" : "=v"(acc1[PAR]));
      } else if constexpr (fr >= 24) {
        constexpr int m = fr - 24, ob = m >> 1, s = m & 1;   // FFN2: output block ob, k-step s of block i - 2
        const rf_u32x4 hb = {hw[PAR][4 * s], hw[PAR][4 * s + 1], hw[PAR][4 * s + 2], hw[PAR][4 * s + 3]};
        acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, afr), __builtin_bit_cast(half8, hb), acc[ob], 0, 0, 0);
      } else {
        constexpr int kk = fr;                           // FFN1: k-step kk of block i
        // FFN1's accumulator lives in the VECTOR half of the register file (hipcc gives every MFMA of a kernel
        // that uses accumulator registers an accumulator-half destination, and the GELU then paid one
        // v_accvgpr_read per value): inline asm with vector-register C / D.  Nothing reads it before the next
        // step's GELU (a barrier and hundreds of instructions away: no XDL-write hazard to pad by hand).
        if constexpr (kk == 0) {
          f32x16 b0 = {bq[0][0], bq[0][1], bq[0][2], bq[0][3], bq[1][0], bq[1][1], bq[1][2], bq[1][3],
                       bq[2][0], bq[2][1], bq[2][2], bq[2][3], bq[3][0], bq[3][1], bq[3][2], bq[3][3]};
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(b0) : "v"(afr), "v"(xf[kk]));
          acc1[PAR] = b0;
        } else {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1[PAR]) : "v"(afr), "v"(xf[kk]));
        }
      }
      if constexpr (GE) {   // GELU of block i - 1 (accumulator and h of the other parity): this gap's slots
        // (R, D) = (4, 24) measured: 2 151 cycles per step against 2 207 for the uniform spread (ABL 32); (4, 12) 2 218,
        // (0, 8) 2 191, (8, 16) 2 224, (0, 24) 2 217, (8, 32) 2 221, (12, 36) 2 212 -- no slot in an LDS-DMA gap is what counts
        constexpr GeluSched sched = gelu_sched(4, 24);
        constexpr bool uniform = NM != 48 || (ABL & 32);
        constexpr int o0 = uniform ? n * GELU_SLOTS / NM : sched.start[n], o1 = uniform ? (n + 1) * GELU_SLOTS / NM : sched.start[n + 1];
        static_for<o0, o1>([&](auto Oc) __attribute__((always_inline)) {
          gelu_slot<decltype(Oc)::value>(acc1[PAR ^ 1], gt, hw[PAR ^ 1]);
        });
      }
      if constexpr (!(ABL & 1)) {
        // 12 pieces per step whatever the step holds: one per four MFMAs, or one per two where the step has 24
        // (the 12 pieces as three groups of four behind one M0 write -- 18 instructions fewer -- measured 2 187 cycles per step
        // against 2 107: four loads in one gap overflow it)
        if constexpr (NM == 48) {
          if constexpr ((n & 3) == 3) issue_piece(nsrc, ndst, std::integral_constant<int, (n >> 2)>{});
        } else {
          if constexpr ((n & 1) == 1) issue_piece(nsrc, ndst, std::integral_constant<int, (n >> 1)>{});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using E0 = std::integral_constant<int, 0>;
  using E2 = std::integral_constant<int, 2>;
  mlp_step(F_{}, T_{}, F_{}, P0{}, 0, E0{}, F_{});
  mlp_step(F_{}, T_{}, T_{}, P1{}, 1, E0{}, F_{});
#pragma unroll 1
  for (int i = 2; i < 48; i += 2) {
    mlp_step(T_{}, T_{}, T_{}, P0{}, i, E0{}, F_{});
    mlp_step(T_{}, T_{}, T_{}, P1{}, i + 1, E0{}, F_{});
  }
  mlp_step(T_{}, F_{}, T_{}, P0{}, 48, E0{}, T_{});
  mlp_step(T_{}, F_{}, F_{}, P1{}, 49, E2{}, F_{});
  static_for<0, KS>([&](auto Kc) __attribute__((always_inline)) {   // + y (the fp16 operands still in registers)
    constexpr int kk = decltype(Kc)::value;
    acc[kk >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(id_perm[kk & 1], __builtin_bit_cast(half8, xf[kk]), acc[kk >> 1], 0, 0, 0);
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pieces past the last step must land before the LDS is handed on
  if (DBG) {
    asm volatile("" : "+v"(acc[11]));
    ts_ln2 = __builtin_amdgcn_s_memtime();
  }

  // ---- LayerNorm 2 over v = acc; x' -> HBM ------------------------------------------------------------
  {
    float sc, sh;
    ln_stats(sc, sh);
    _Float16* const dst = a.out + (tb * KS * 64 + lane) * 8;
    const bool live = t0 + 32 * wave + r32 < M;   // the token whose 16-byte slots this lane stores
    static_for<0, 12>([&](auto Fc) __attribute__((always_inline)) {
      constexpr int fb = decltype(Fc)::value;
      uint2 pk[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4v gv = param4(4, fb, g), be = param4(5, fb, g);
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (_Float16)fmaf(fmaf(acc[fb][4 * g + j], sc, sh), gv[j], be[j]);
        pk[g] = __builtin_bit_cast(uint2, o);
      }
      // the 16-byte slot (token, 8 features) of the tiled layout is split between lanes c and c + 32: the packed
      // quads 2 m and 2 m + 1 go through v_permlane32_swap and the wave then holds fragment 2 fb + m lane-linear
      const auto sx0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
      const auto sy0 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
      const auto sx1 = __builtin_amdgcn_permlane32_swap(pk[2].x, pk[3].x, false, false);
      const auto sy1 = __builtin_amdgcn_permlane32_swap(pk[2].y, pk[3].y, false, false);
      if (live) {
        *(uint4*)(dst + (size_t)(2 * fb) * 512) = make_uint4(sx0[0], sy0[0], sx0[1], sy0[1]);
        *(uint4*)(dst + (size_t)(2 * fb + 1) * 512) = make_uint4(sx1[0], sy1[0], sx1[1], sy1[1]);
      }
      // ... and as the accumulator left them they are the B operands of the next layer's QKV projection
      xf[2 * fb] = rf_u32x4{pk[0].x, pk[0].y, pk[1].x, pk[1].y};
      xf[2 * fb + 1] = rf_u32x4{pk[2].x, pk[2].y, pk[3].x, pk[3].y};
    });
  }
  const uint64_t ts_c = DBG ? __builtin_amdgcn_s_memtime() : 0;

  // ---- phase C: Q | K | V of the NEXT layer = x' Wqkv^T + b, 36 blocks of 32 features, two per ring step --------
  // The same loop shape as FFN1 (vector-register accumulators through inline-asm MFMAs, the two blocks' chains
  // alternating, bias as the accumulator input); the fp16 conversion, half exchange and the two 16-byte stores of
  // the previous step's blocks ride in the MFMA gaps.  Stores are unconditional (the buffers are padded to whole
  // tiles, nobody reads rows past the token count): exactly four per step, which the counted waits rely on.
  if (a.qkv_out) {
    f32x16 qa[2][2];   // [step parity][block of the step]
    // (ABL 16: every workgroup stores into the first tile's rows -- an L2-resident window: what the stores cost without HBM)
    _Float16* const qdst = a.qkv_out + (((ABL & 16) ? (size_t)wave : tb) * (3 * KS) * 64 + lane) * 8;
    uint32_t pkw[8];
    uint32_t sw[4];
    auto epi_slot = [&](auto Oc, const f32x16 (&src)[2], int blk0) __attribute__((always_inline)) {
      constexpr int O = decltype(Oc)::value, b = O / 14, w = O % 14;   // 28 slots: 2 blocks x (8 conversions, 4 exchanges, 2 stores)
      if constexpr (w < 8) {
        const half2v o = {(_Float16)src[b][2 * w], (_Float16)src[b][2 * w + 1]};
        pkw[w] = __builtin_bit_cast(uint32_t, o);
      } else if constexpr (w < 12) {
        constexpr int m = (w - 8) >> 1, c = (w - 8) & 1;
        const auto sx = __builtin_amdgcn_permlane32_swap(pkw[4 * m + c], pkw[4 * m + 2 + c], false, false);
        sw[2 * c] = sx[0];
        sw[2 * c + 1] = sx[1];
        if constexpr (c == 1) {   // both words of the pair exchanged: fragment 2 (blk0 + b) + m, lane-linear
          // (non-temporal stores measured no better: 65 k against 53-62 k cycles for the phase)
          *(uint4*)(qdst + (size_t)(2 * (blk0 + b) + m) * 512) = make_uint4(sw[0], sw[2], sw[1], sw[3]);
        }
      }
    };
    auto qkv_step = [&](auto epic, auto parc, int t) __attribute__((always_inline)) {
      constexpr bool EPI = decltype(epic)::value;     // the previous step's blocks are finished under this step's MFMAs
      constexpr int PAR = decltype(parc)::value;
      const int S = PB_STEPS + t;
      if (t >= 2 && !(ABL & 2)) sync_step(std::integral_constant<int, 4>{});
      else sync_step(E0{});
      const uint32_t sa = slots_a + (uint32_t)((S % PB_SLOTS) * PB_FRAGS * 1024);
      const char* const nsrc = step_src(S + 2);
      const uint32_t ndst = step_dst(S + 2);
      f32x4v bq[2][4];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const uint32_t ba = bias_a + (uint32_t)(2 * t + b) * 128u;
        asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(bq[b][0]) : "v"(ba));
        asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(bq[b][1]) : "v"(ba));
        asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(bq[b][2]) : "v"(ba));
        asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(bq[b][3]) : "v"(ba));
      }
      run_step<PB_FRAGS>([](int m) constexpr { return m; }, sa, [&](auto Mc, const rf_u32x4& afr) __attribute__((always_inline)) {
        constexpr int m = decltype(Mc)::value, b = m & 1, kk = m >> 1;
        if constexpr (m == 0)
          asm volatile("" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[1][2]), "+v"(bq[1][3]));
        if constexpr (kk == 0) {
          f32x16 b0 = {bq[b][0][0], bq[b][0][1], bq[b][0][2], bq[b][0][3], bq[b][1][0], bq[b][1][1], bq[b][1][2], bq[b][1][3],
                       bq[b][2][0], bq[b][2][1], bq[b][2][2], bq[b][2][3], bq[b][3][0], bq[b][3][1], bq[b][3][2], bq[b][3][3]};
          if constexpr (ABL & 4) {
            qa[PAR][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, afr), __builtin_bit_cast(half8, xf[kk]), b0, 0, 0, 0);
          } else {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(b0) : "v"(afr), "v"(xf[kk]));
            qa[PAR][b] = b0;
          }
        } else {
          if constexpr (ABL & 4)
            qa[PAR][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, afr), __builtin_bit_cast(half8, xf[kk]), qa[PAR][b], 0, 0, 0);
          else
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(qa[PAR][b]) : "v"(afr), "v"(xf[kk]));
        }
        if constexpr (EPI && !(ABL & 2)) {
          constexpr int o0 = m * 28 / PB_FRAGS, o1 = (m + 1) * 28 / PB_FRAGS;
          static_for<o0, o1>([&](auto Oc) __attribute__((always_inline)) { epi_slot(Oc, qa[PAR ^ 1], 2 * t - 2); });
        }
        if constexpr ((m & 3) == 3) issue_piece(nsrc, ndst, std::integral_constant<int, (m >> 2)>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    qkv_step(F_{}, P0{}, 0);
    qkv_step(T_{}, P1{}, 1);
#pragma unroll 1
    for (int t = 2; t < PB_STEPS_C; t += 2) {
      qkv_step(T_{}, P0{}, t);
      qkv_step(T_{}, P1{}, t + 1);
    }
    // the last step's blocks: the MFMA results are read by the vector ALU right away -- the XDL write hazard is padded by hand
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(qa[1][0]), "+v"(qa[1][1]));
    static_for<0, 28>([&](auto Oc) __attribute__((always_inline)) { epi_slot(Oc, qa[1], 2 * PB_STEPS_C - 2); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (DBG && a.dbg && lane == 0 && blockIdx.x < 512) {   // the buffer holds 4096 waves x 8 floats
    float* d = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    const uint64_t te = __builtin_amdgcn_s_memtime();
    d[0] = (float)(te - ts_entry);       // cycles, whole wave
    d[1] = (float)(ts_a - ts_entry);     // prologue
    d[2] = (float)(ts_ln1 - ts_a);       // out-projection (6 steps + residual)
    d[3] = (float)(ts_b - ts_ln1);       // LayerNorm 1
    d[4] = (float)(ts_ln2 - ts_b);       // MLP (50 steps + residual)
    d[5] = (float)(ts_c - ts_ln2);       // LayerNorm 2 + stores
    d[6] = (float)t_wait;                // vmcnt wait + barrier, all 56 steps
    d[7] = (float)(__builtin_amdgcn_s_memrealtime() - tr_entry);   // 100-MHz ticks, whole wave
    float* d2 = d + 4 * 8;               // the rows of waves 4-7 (the workgroup has four): out-projection step by step
    d2[0] = (float)(ts_step[0] - ts_a);
    for (int q = 1; q < PB_STEPS_A; ++q) d2[q] = (float)(ts_step[q] - ts_step[q - 1]);
    d2[6] = (float)(ts_ln1 - ts_step[PB_STEPS_A - 1]);   // residual MFMAs
    d2[7] = (float)(te - ts_c);                          // the next layer's QKV projection (18 steps)
  }
}

int rf_launch_post_block(const rf_post_args& a_in, int token_slots, hipStream_t st) {
  rf_post_args a = a_in;
#ifdef RF_EXPERIMENTS
  if (a.abl & 256) {   // stamps of the launches WITH the QKV phase only (the last layer's launch has none and would overwrite them)
    if (!a.qkv_out) a.dbg = nullptr;
    a.abl &= ~256;
  }
#endif
  const dim3 grid((token_slots + PB_TOK - 1) / PB_TOK), block(PB_WAVES * 64);
  const size_t lds = PB_LDS_BYTES;
#define RF_PB_LAUNCH(D, A)                                                  \
  do {                                                                      \
    static rf_lds_attr attr_;   /* per instantiation, per device */         \
    RF_HIP(rf_ensure_lds(attr_, (const void*)k_post_block<D, A>, lds));      \
    hipLaunchKernelGGL((k_post_block<D, A>), grid, block, lds, st, a);       \
    return RF_OK;                                                           \
  } while (0)
#ifdef RF_EXPERIMENTS
  if (a.dbg || a.abl) {
    switch (a.abl) {
      case 1: RF_PB_LAUNCH(1, 1);
      case 2: RF_PB_LAUNCH(1, 2);
      case 3: RF_PB_LAUNCH(1, 3);
      case 8: RF_PB_LAUNCH(1, 8);
      case 4: RF_PB_LAUNCH(1, 4);
      case 16: RF_PB_LAUNCH(1, 16);
      case 32: RF_PB_LAUNCH(1, 32);
      default: RF_PB_LAUNCH(1, 0);
    }
  }
#endif
  RF_PB_LAUNCH(0, 0);
#undef RF_PB_LAUNCH
}
