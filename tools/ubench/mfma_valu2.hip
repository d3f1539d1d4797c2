// Micro-benchmark 2 (diagnostic): how many plain fp32 VALU operations fit under one MFMA 32x32x16
// when BOTH waves of a SIMD run the same mixed stream, as a function of (a) VALU operations per
// MFMA and (b) the number of independent dependency chains they form; and the issue rate of a
// dependent v_fma_f32 chain.   hipcc --offload-arch=gfx950 -O3 -o mfma_valu2 mfma_valu2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one iteration = 8 x { 1 MFMA (if MF), VPM VALU operations spread over NCH chains }; LIT: v_fmaak_f32 (literal)
template <int MF, int VPM, int NCH, int LIT>
__device__ __forceinline__ float body(int iters, float seed) {
  f32x16 acc = {};
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)seed; b[i] = (_Float16)(seed + i); }
  float v[NCH];
  for (int c = 0; c < NCH; ++c) v[c] = seed + c;
  const float m = seed, k = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
      for (int c = 0; c < VPM; ++c) {
        if (LIT) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3dcccccd" : "+v"(v[c % NCH]) : "v"(m));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c % NCH]) : "v"(m), "v"(k));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  for (int c = 0; c < NCH; ++c) s += v[c];
  return s;
}

template <int MF, int VPM, int NCH, int LIT, int BOTH>
__global__ void __launch_bounds__(512, 1) k(int iters, float seed, float* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  if (BOTH || wave < 4) s = body<MF, VPM, NCH, LIT>(iters, seed);
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (float)(t1 - t0) / (float)iters;
}

template <int MF, int VPM, int NCH, int LIT, int BOTH>
static void run(int cus) {
  float *out, *sink;
  (void)hipMalloc(&out, cus * 8 * sizeof(float));
  (void)hipMalloc(&sink, 512 * sizeof(float));
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MF, VPM, NCH, LIT, BOTH>), dim3(cus), dim3(512), 0, 0, iters, 1.0f, out, sink);
  (void)hipDeviceSynchronize();
  std::vector<float> h(cus * 8);
  (void)hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  double a = 0;
  for (int i = 0; i < cus; ++i)
    for (int w = 0; w < 4; ++w) a += h[i * 8 + w];
  a /= cus * 4;
  printf("mfma %d  valu/mfma %2d  chains %2d  literal %d  waves/SIMD %d : %7.1f cycles per 8 slots = %5.1f per slot", MF, VPM, NCH, LIT,
         BOTH + 1, a, a / 8);
  if (VPM) printf(" = %4.2f per VALU op", a / 8 / VPM);
  printf("\n");
  (void)hipFree(out);
  (void)hipFree(sink);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.gcnArchName, cus);
  // dependent-chain issue rate of v_fma_f32, one wave and two waves per SIMD
  run<0, 8, 1, 0, 0>(cus); run<0, 8, 2, 0, 0>(cus); run<0, 8, 4, 0, 0>(cus); run<0, 8, 8, 0, 0>(cus);
  run<0, 8, 1, 0, 1>(cus); run<0, 8, 2, 0, 1>(cus); run<0, 8, 4, 0, 1>(cus); run<0, 8, 8, 0, 1>(cus);
  run<0, 8, 2, 1, 0>(cus); run<0, 8, 2, 1, 1>(cus); run<0, 8, 8, 1, 1>(cus);
  // mixed streams
  run<1, 0, 1, 0, 0>(cus); run<1, 0, 1, 0, 1>(cus);
  run<1, 4, 2, 0, 1>(cus); run<1, 8, 2, 0, 1>(cus); run<1, 12, 2, 0, 1>(cus); run<1, 16, 2, 0, 1>(cus);
  run<1, 8, 4, 0, 1>(cus); run<1, 12, 4, 0, 1>(cus); run<1, 16, 4, 0, 1>(cus);
  run<1, 12, 12, 0, 1>(cus); run<1, 12, 2, 1, 1>(cus); run<1, 12, 4, 1, 1>(cus);
  run<1, 12, 2, 0, 0>(cus); run<1, 12, 4, 0, 0>(cus);
  return 0;
}
