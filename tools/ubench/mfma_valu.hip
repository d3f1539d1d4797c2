// Micro-benchmark (diagnostic, not product): do VALU instructions of one wave issue under the
// MFMAs of the OTHER wave of the same SIMD on gfx950, and does it matter whether they are plain
// fp32 (v_fma_f32), packed fp32 (v_pk_fma_f32) or packed fp16 (v_pk_fma_f16)?
// One workgroup of 512 threads per CU: waves 0-3 take role A, waves 4-7 role B (wave w and w+4
// share a SIMD).  Prints cycles per loop iteration for each role, alone and together.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu tools/ubench/mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

enum { R_IDLE = 0, R_MFMA32 = 1, R_MFMA16 = 2, R_FMA = 3, R_PKFMA = 4, R_PKF16 = 5, R_MIX32_FMA = 6, R_MIX32_PK = 7 };

#define CHAINS 8
template <int ROLE>
__device__ __forceinline__ void body(int iters, float seed, float* sink) {
  if (ROLE == R_MFMA32 || ROLE == R_MIX32_FMA || ROLE == R_MIX32_PK) {
    f32x16 acc = {};
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)seed; b[i] = (_Float16)(seed + i); }
    f32x2 v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = f32x2{seed + c, seed - c};
    const f32x2 m = {seed, seed}, k = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (ROLE == R_MIX32_FMA) {
#pragma unroll
          for (int c = 0; c < 4; ++c) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c][0]) : "v"(m[0]), "v"(k[0])); }
        }
        if (ROLE == R_MIX32_PK) {
#pragma unroll
          for (int c = 0; c < 4; ++c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(m), "v"(k));
        }
      }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int c = 0; c < CHAINS; ++c) s += v[c][0] + v[c][1];
    *sink = s;
  } else if (ROLE == R_MFMA16) {
    f32x4 acc0 = {}, acc1 = {};
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)seed; b[i] = (_Float16)(seed + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc1, 0, 0, 0);
      }
    }
    *sink = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  } else if (ROLE == R_FMA) {
    float v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = seed + c;
    const float m = seed, k = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(m), "v"(k));
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    *sink = s;
  } else if (ROLE == R_PKFMA) {
    f32x2 v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = f32x2{seed + c, seed - c};
    const f32x2 m = {seed, seed}, k = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(m), "v"(k));
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c][0] + v[c][1];
    *sink = s;
  } else if (ROLE == R_PKF16) {
    half2v v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = half2v{(_Float16)(seed + c), (_Float16)(seed - c)};
    const half2v m = {(_Float16)seed, (_Float16)seed}, k = {(_Float16)0.5f, (_Float16)0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v[c]) : "v"(m), "v"(k));
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += (float)v[c][0] + (float)v[c][1];
    *sink = s;
  }
}

template <int RA, int RB>
__global__ void __launch_bounds__(512, 1) k_pair(int iters, float seed, float* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  if (wave < 4) body<RA>(iters, seed, &s);
  else body<RB>(iters, seed, &s);
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (float)(t1 - t0) / (float)iters;
}

template <int RA, int RB>
static void run(const char* name, int cus) {
  float *out, *sink;
  (void)hipMalloc(&out, cus * 8 * sizeof(float));
  (void)hipMalloc(&sink, 512 * sizeof(float));
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_pair<RA, RB>), dim3(cus), dim3(512), 0, 0, iters, 1.0f, out, sink);
  (void)hipDeviceSynchronize();
  std::vector<float> h(cus * 8);
  (void)hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < cus; ++i)
    for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[i * 8 + w];
  printf("%-34s waves0-3 %8.1f cycles/iter   waves4-7 %8.1f cycles/iter\n", name, a / (cus * 4), b / (cus * 4));
  (void)hipFree(out);
  (void)hipFree(sink);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs; one iteration = 8 MFMA 32x32x16 (or 16 MFMA 16x16x32) | 64 VALU ops (8 chains x 8); mixed = 8 MFMA + 32 VALU\n",
         p.gcnArchName, cus);
  run<R_MFMA32, R_IDLE>("mfma32 | idle", cus);
  run<R_MFMA32, R_MFMA32>("mfma32 | mfma32", cus);
  run<R_MFMA16, R_IDLE>("mfma16 | idle", cus);
  run<R_MFMA16, R_MFMA16>("mfma16 | mfma16", cus);
  run<R_FMA, R_IDLE>("v_fma_f32 | idle", cus);
  run<R_FMA, R_FMA>("v_fma_f32 | v_fma_f32", cus);
  run<R_PKFMA, R_IDLE>("v_pk_fma_f32 | idle", cus);
  run<R_PKFMA, R_PKFMA>("v_pk_fma_f32 | v_pk_fma_f32", cus);
  run<R_PKF16, R_IDLE>("v_pk_fma_f16 | idle", cus);
  run<R_MFMA32, R_FMA>("mfma32 | v_fma_f32", cus);
  run<R_MFMA32, R_PKFMA>("mfma32 | v_pk_fma_f32", cus);
  run<R_MFMA32, R_PKF16>("mfma32 | v_pk_fma_f16", cus);
  run<R_MFMA16, R_FMA>("mfma16 | v_fma_f32", cus);
  run<R_MFMA16, R_PKFMA>("mfma16 | v_pk_fma_f32", cus);
  run<R_MIX32_FMA, R_IDLE>("mfma32+4 v_fma_f32 each | idle", cus);
  run<R_MIX32_PK, R_IDLE>("mfma32+4 v_pk_fma_f32 each | idle", cus);
  run<R_MIX32_FMA, R_MIX32_FMA>("mixed fma | mixed fma", cus);
  run<R_MIX32_PK, R_MIX32_PK>("mixed pk | mixed pk", cus);
  return 0;
}
