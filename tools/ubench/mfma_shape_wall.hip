// Micro-benchmark (diagnostic, not product): the k_post_block inner loop in its two possible MFMA shapes, by WALL
// time on random data.  One wave per SIMD (256 threads per CU, 512-register budget), the A operand (weights) re-read
// from LDS with one ds_read_b128 per 1-KiB fragment, the B operand (32 tokens of activations) resident in
// registers, fp32 accumulators:
//   shape 0: v_mfma_f32_32x32x16_f16 -- fragment = 32 features x 16 k, ONE MFMA on the wave's 32 tokens
//   shape 1: v_mfma_f32_16x16x32_f16 -- fragment = 16 features x 32 k, TWO MFMAs (token tiles of 16)
// Same FLOPs and the same LDS bytes per fragment (32 768 FLOP per KiB).  MI355X_MICROARCH.md, "DVFS give-back" item 7:
// where the chip holds its clock down under MFMA load the clock it holds depends on the shape.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_wall tools/ubench/mfma_shape_wall.hip && ./mfma_shape_wall
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define FRAGS 48   // fragments per step (48 KiB of LDS)
#define KS 24      // resident B fragments per wave (32 tokens x 384 features)

template <int SHAPE, int VALU>
__global__ void __launch_bounds__(256, 1) k(const u32x4* __restrict__ w, const u32x4* __restrict__ x, float* out, float* stamps, int steps) {
  __shared__ u32x4 lds[FRAGS * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < FRAGS * 64; i += 256) lds[i] = w[i];
  u32x4 b[KS];
#pragma unroll
  for (int i = 0; i < KS; ++i) b[i] = x[(blockIdx.x * 4 + (tid >> 6)) * KS * 64 + i * 64 + lane];
  __syncthreads();
  f32x16 acc32[4];
  f32x4 acc16[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[4] = {1.f + lane, 2.f, 3.f, 4.f};
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int f = 0; f < FRAGS; ++f) {
      const u32x4 a = lds[f * 64 + lane];
      if (SHAPE == 0) {
        acc32[f & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b[f % KS]), acc32[f & 3], 0, 0, 0);
      } else if (SHAPE == 2) {   // 32x32x16, the fragment used for TWO token blocks (64 tokens per wave): half the LDS bytes per FLOP
        acc32[f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b[f % KS]), acc32[f & 1], 0, 0, 0);
        acc32[2 + (f & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b[(f + 5) % KS]), acc32[2 + (f & 1)], 0, 0, 0);
      } else if (SHAPE == 3) {   // 16x16x32, the fragment used for FOUR token tiles (64 tokens per wave)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc16[(4 * f + t) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b[(f + 5 * t) % KS]), acc16[(4 * f + t) & 15], 0, 0, 0);
      } else {
        acc16[(2 * f) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b[f % KS]), acc16[(2 * f) & 15], 0, 0, 0);
        acc16[(2 * f + 1) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b[(f + 7) % KS]), acc16[(2 * f + 1) & 15], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < VALU; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 3]) : "v"(0.999f), "v"(0.001f));
    }
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = v[0] + v[1] + v[2] + v[3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc32[i][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
  out[blockIdx.x * 256 + tid] = sum;
  if (lane == 0) {
    float* o = stamps + (blockIdx.x * 4 + (tid >> 6)) * 2;
    o[0] = (float)(c1 - c0);
    o[1] = (float)(r1 - r0);
  }
}

template <int SHAPE, int VALU>
static void run(const char* name, const u32x4* w, const u32x4* x, float* out, float* stamps, int cus, int steps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  // >= 1 s of back-to-back launches first (the clock settles), then the timed ones
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k<SHAPE, VALU>), dim3(cus), dim3(256), 0, 0, w, x, out, stamps, steps);
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<SHAPE, VALU>), dim3(cus), dim3(256), 0, 0, w, x, out, stamps, steps);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h((size_t)cus * 4 * 2);
  (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int i = 0; i < cus * 4; ++i) {
    clk.push_back(h[2 * i] / h[2 * i + 1] * 0.1);
    cyc.push_back(h[2 * i] / ((double)steps * FRAGS));
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  const double flop = (double)reps * cus * 4 * (double)steps * FRAGS * 32768.0 * (SHAPE >= 2 ? 2.0 : 1.0);
  printf("%-44s %7.1f TFLOP/s  %6.2f ms/launch  clock %.2f GHz  %.1f cycles per fragment\n", name, flop / (ms * 1e-3) / 1e12,
         ms / reps, clk[clk.size() / 2], cyc[cyc.size() / 2]);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const size_t nw = (size_t)FRAGS * 64, nx = (size_t)cus * 4 * KS * 64;
  std::vector<uint16_t> hw(nw * 8), hx(nx * 8);
  srand(7);
  auto rnd_half = []() {   // random fp16 in (-1, 1): sign, exponent 10..14, random mantissa
    return (uint16_t)(((rand() & 1) << 15) | ((10 + rand() % 5) << 10) | (rand() & 1023));
  };
  for (auto& e : hw) e = rnd_half();
  for (auto& e : hx) e = rnd_half();
  u32x4 *w, *x;
  float *out, *stamps;
  (void)hipMalloc(&w, nw * 16);
  (void)hipMalloc(&x, nx * 16);
  (void)hipMalloc(&out, (size_t)cus * 256 * 4);
  (void)hipMalloc(&stamps, (size_t)cus * 4 * 2 * 4);
  (void)hipMemcpy(w, hw.data(), nw * 16, hipMemcpyHostToDevice);
  (void)hipMemcpy(x, hx.data(), nx * 16, hipMemcpyHostToDevice);
  const int steps = 12000;   // ~25 ms per launch
  printf("%s, %d CUs, one wave per SIMD, random fp16 operands; fragment = 1 KiB of A from LDS = 32 768 FLOP\n", p.gcnArchName, cus);
  run<0, 0>("32x32x16, no VALU", w, x, out, stamps, cus, steps);
  run<1, 0>("16x16x32, no VALU", w, x, out, stamps, cus, steps);
  run<0, 3>("32x32x16 + 3 v_fma_f32 per fragment", w, x, out, stamps, cus, steps);
  run<1, 3>("16x16x32 + 3 v_fma_f32 per fragment", w, x, out, stamps, cus, steps);
  run<0, 0>("32x32x16, no VALU (again)", w, x, out, stamps, cus, steps);
  run<1, 0>("16x16x32, no VALU (again)", w, x, out, stamps, cus, steps);
  run<2, 0>("32x32x16, fragment used twice (64 tokens)", w, x, out, stamps, cus, steps / 2);
  run<3, 0>("16x16x32, fragment used four times (64 t.)", w, x, out, stamps, cus, steps / 2);
  return 0;
}
