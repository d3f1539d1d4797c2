// Declarations shared by the encoder's translation units (encoder.hip, encoder_post.hip).
#pragma once
#include "rf_internal.h"
#include "lds_ring.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define HID 384
#define HEAD_DIM 32

// Activations live in the same fragment tiling as the corpus and the weights:
// [token block of 32][k-step = feature/16][lane = 32*((feature/8)&1) + token%32][8 halfs].
// A 32-token x 16-feature fragment is 1 KiB contiguous, so the GEMMs read their B
// operands with one coalesced wave load, and an epilogue's 4-consecutive-feature
// stores of a wave fill 512 contiguous bytes.  (Row-major activations made every
// B-fragment load touch 32 different cache lines: the GEMMs were TA-bound at ~17 %
// of the matrix peak.)  toff() = offset in halfs of (token t, feature f); KSf = width/16.
__device__ __forceinline__ size_t toff(int t, int f, int KSf) {
  return (((size_t)(t >> 5) * KSf + (f >> 4)) * 64 + (size_t)(((f >> 3) & 1) * 32 + (t & 31))) * 8 + (f & 7);
}

// (the attribute means something in the device pass only; the host pass of the same source would warn)
#ifdef __HIP_DEVICE_COMPILE__
#define RF_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define RF_NO_PACKED_FP32
#endif

// ---- encoder_post.hip: the layer's post-attention half in one launch ---------------------
// out-projection + residual + LayerNorm, FFN1 + GELU, FFN2 + residual + LayerNorm for 128 tokens per
// workgroup; the activations between the three GEMMs never leave the registers.
#define PB_STEPS_A 6    // ring steps of the out-projection: 12 feature blocks, two per step
#define PB_STEPS_B 50   // ring steps of the MLP part: 48 intermediate blocks + 2 of pipeline drain
#define PB_FRAGS 48     // 1-KiB fragments per ring step
#define PB_PARAM_FRAGS 15   // fp32 parameter block: b1 [1536], bo, g1, be1, b2, g2, be2 [384 each] = 15 KiB
#define PB_STEPS_C 18   // ring steps of the NEXT layer's QKV projection: 36 feature blocks, two per step
#define PB_QB_FRAGS 8   // the next layer's QKV bias as fp32 [1152] (4.5 KiB), padded
// per-layer pack the kernel reads: [parameters, padded to 16 fragments][out-projection 288][MLP stream 2400]
// [QKV of the next layer 864][its bias 8]
#define PB_RING_FRAGS ((PB_STEPS_A + PB_STEPS_B + PB_STEPS_C) * PB_FRAGS)
#define PB_PACK_FRAGS (16 + PB_RING_FRAGS + PB_QB_FRAGS)
static inline size_t rf_post_pack_elems(void) { return (size_t)PB_PACK_FRAGS * 512; }   // halfs per layer
// row-major weights / biases of all L layers -> pack [L][PB_PACK_FRAGS][64 lanes][16 B]
void rf_launch_post_pack_build(const rf_encoder_weights* w, void* pack, int L, hipStream_t st);
struct rf_post_args {
  const _Float16* ctx;      // [Mpad, 384] tiled: attention output
  const _Float16* res;      // [Mpad, 384] tiled: the layer's input (residual of the first LayerNorm)
  _Float16* out;            // [Mpad, 384] tiled: the layer's output
  _Float16* qkv_out;        // [Mpad, 1152] tiled: Q | K | V of the NEXT layer (its weights are in this layer's pack), or nullptr
  const uint4* pack;        // this layer's pack (rf_launch_post_pack_build)
  float eps;
  const int32_t* m_ptr;     // packed token count
  float* dbg;               // clock stamps (experiments build), or nullptr
  int abl;                  // ablation bits (experiments build; results wrong): see k_post_block
};
int rf_launch_post_block(const rf_post_args& a, int token_slots, hipStream_t st);
