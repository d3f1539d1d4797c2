// Microbenchmark: cycles per v_mfma_f32_32x32x16_f16 with the B operand in arch VGPRs vs in
// the accumulator half (AGPRs), one wave per SIMD, two accumulation chains.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_operand_bench mfma_operand_bench.hip && ./mfma_operand_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int BKIND, int NACC>   // BKIND 0: B in VGPR, 1: B in AGPR, 2: A and B in AGPR
__global__ void __launch_bounds__(256, 1) k(const u32x4* in, float* out, uint64_t* cyc, int iters) {
  const int lane = threadIdx.x;
  u32x4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = in[lane + 256 * i];
    b[i] = in[lane + 256 * (8 + i)];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (BKIND >= 1) asm volatile("" : "+a"(b[i]));
    else asm volatile("" : "+v"(b[i]));
    if (BKIND == 2) asm volatile("" : "+a"(a[i]));
    else asm volatile("" : "+v"(a[i]));
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int n = 0; n < NACC; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[i]),
                                                        __builtin_bit_cast(half8, b[(i + n) & 7]), acc[n], 0, 0, 0);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int BKIND, int NACC>
static void run(const char* name, const u32x4* in, float* out, uint64_t* cyc, int iters) {
  hipLaunchKernelGGL((k<BKIND, NACC>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<BKIND, NACC>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += (double)h[i];
  m /= 256;
  const double n_mfma = (double)iters * 8 * NACC;
  printf("%-28s %6.2f cycles/MFMA  %7.1f us  (%.2f GHz effective)\n", name, m / n_mfma, ms * 1e3,
         m / (ms * 1e-3) / 1e9);
}


typedef float f32x4 __attribute__((ext_vector_type(4)));
// same flops per iteration with v_mfma_f32_16x16x32_f16: 2 x the instructions, 4-register tiles
template <int NACC>
__global__ void __launch_bounds__(256, 1) k16(const u32x4* in, float* out, uint64_t* cyc, int iters) {
  const int lane = threadIdx.x;
  u32x4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = in[lane + 256 * i];
    b[i] = in[lane + 256 * (8 + i)];
    asm volatile("" : "+v"(a[i]));
    asm volatile("" : "+v"(b[i]));
  }
  f32x4 acc[NACC];
#pragma unroll
  for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int n = 0; n < NACC; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a[i & 7]),
                                                        __builtin_bit_cast(half8, b[(i + n) & 7]), acc[n], 0, 0, 0);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  out[blockIdx.x * 256 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
static void run16(const char* name, const u32x4* in, float* out, uint64_t* cyc, int iters) {
  hipLaunchKernelGGL((k16<NACC>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k16<NACC>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[256];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += (double)h[i];
  m /= 256;
  const double n_mfma = (double)iters * 16 * NACC;
  printf("%-28s %6.2f cycles/MFMA  %7.1f us  (%.2f GHz effective)  [same flops as the 32x32x16 rows]\n", name,
         m / n_mfma, ms * 1e3, m / (ms * 1e-3) / 1e9);
}

int main() {
  u32x4* in;
  float* out;
  uint64_t* cyc;
  hipMalloc(&in, 256 * 16 * sizeof(u32x4));
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 256 * 8);
  // random fp16 in [-1, 1)
  uint16_t* h = (uint16_t*)malloc(256 * 16 * 16);
  uint32_t x = 12345;
  for (int i = 0; i < 256 * 16 * 8; ++i) {
    x = x * 1664525u + 1013904223u;
    const float f = ((x >> 8) & 0xffff) / 32768.0f - 1.0f;
    _Float16 hf = (_Float16)f;
    h[i] = *(uint16_t*)&hf;
  }
  hipMemcpy(in, h, 256 * 16 * 16, hipMemcpyHostToDevice);
  const int iters = 2000;
  run<0, 2>("B VGPR, 2 chains", in, out, cyc, iters);
  run<1, 2>("B AGPR, 2 chains", in, out, cyc, iters);
  run<2, 2>("A+B AGPR, 2 chains", in, out, cyc, iters);
  run<0, 1>("B VGPR, 1 chain", in, out, cyc, iters);
  run<1, 1>("B AGPR, 1 chain", in, out, cyc, iters);
  run<0, 4>("B VGPR, 4 chains", in, out, cyc, iters);
  run<1, 4>("B AGPR, 4 chains", in, out, cyc, iters);
  run16<2>("16x16x32, 2 chains", in, out, cyc, iters);
  run16<4>("16x16x32, 4 chains", in, out, cyc, iters);
  run16<8>("16x16x32, 8 chains", in, out, cyc, iters);
  run<0, 2>("B VGPR, 2 chains (again)", in, out, cyc, iters);
  run16<4>("16x16x32, 4 chains (again)", in, out, cyc, iters);
  return 0;
}
