// Micro-benchmark 3 (diagnostic): issue cost of v_exp_f32 (transcendental) on gfx950 -- alone, with two
// waves per SIMD, alternating with plain v_fma_f32, and under MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one iteration = 8 x { MF MFMAs, NE v_exp_f32 (8 chains), NF v_fma_f32 (8 chains) }
template <int MF, int NE, int NF>
__device__ __forceinline__ float body(int iters, float seed) {
  f32x16 acc = {};
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)seed; b[i] = (_Float16)(seed + i); }
  float e[8], v[8];
  for (int c = 0; c < 8; ++c) { e[c] = seed * 0.01f + c; v[c] = seed + c; }
  const float m = seed, k = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
      for (int c = 0; c < (NE > NF ? NE : NF); ++c) {
        if (c < NE) asm volatile("v_exp_f32 %0, %0" : "+v"(e[c % 8]));
        if (c < NF) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c % 8]) : "v"(m), "v"(k));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int c = 0; c < 8; ++c) s += v[c] + e[c];
  return s;
}

template <int MF, int NE, int NF, int BOTH>
__global__ void __launch_bounds__(512, 1) k(int iters, float seed, float* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  if (BOTH || wave < 4) s = body<MF, NE, NF>(iters, seed);
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (float)(t1 - t0) / (float)iters;
}

template <int MF, int NE, int NF, int BOTH>
static void run(int cus) {
  float *out, *sink;
  (void)hipMalloc(&out, cus * 8 * sizeof(float));
  (void)hipMalloc(&sink, 512 * sizeof(float));
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MF, NE, NF, BOTH>), dim3(cus), dim3(512), 0, 0, iters, 1.0f, out, sink);
  (void)hipDeviceSynchronize();
  std::vector<float> h(cus * 8);
  (void)hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  double a = 0;
  for (int i = 0; i < cus; ++i)
    for (int w = 0; w < 4; ++w) a += h[i * 8 + w];
  a /= cus * 4;
  printf("per slot: mfma %d  v_exp_f32 %2d  v_fma_f32 %2d  waves/SIMD %d : %6.1f cycles per slot\n", MF, NE, NF, BOTH + 1, a / 8);
  (void)hipFree(out);
  (void)hipFree(sink);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.gcnArchName, cus);
  run<0, 8, 0, 0>(cus); run<0, 8, 0, 1>(cus);
  run<0, 0, 8, 0>(cus); run<0, 0, 8, 1>(cus);
  run<0, 8, 8, 0>(cus); run<0, 8, 8, 1>(cus);
  run<0, 4, 12, 0>(cus); run<0, 4, 12, 1>(cus);
  run<1, 4, 0, 0>(cus); run<1, 4, 0, 1>(cus); run<1, 8, 0, 1>(cus);
  run<1, 4, 8, 1>(cus);
  return 0;
}
