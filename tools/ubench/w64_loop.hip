// Micro-benchmark (diagnostic, not product): the inner loop of k_scan_w64 (csrc/scan_wide.hip) taken apart.  One wave per
// SIMD, 192 resident query registers in the accumulator half, the corpus slot in LDS read through run_step's pair ring
// (csrc/lds_ring.h), per fragment two inline-asm MFMAs on four rotating vector-register accumulators.  Variants add
// the pieces of the real loop one at a time: V0 MFMAs + reads only | V1 + a barrier per phase | V2 + the filter's 64
// v_max3 per phase | V3 + 12 LDS-DMA pieces per phase (re-reading one cached KiB).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rag_fin_amd/csrc -o w64_loop tools/ubench/w64_loop.hip && ./w64_loop
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "lds_ring.h"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define FRAGS 48
#define KS 24

template <int V>
__global__ void __launch_bounds__(256, 1) k(const u32x4* __restrict__ w, const u32x4* __restrict__ x, float* out, float* stamps, int phases) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* lds = (u32x4*)smem;   // 3 slots x 48 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 3 * FRAGS * 64; i += 256) lds[i] = w[i % (FRAGS * 64)];
  u32x4 qf[2][KS];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      u32x4 v = x[((blockIdx.x * 4 + wave) * 2 * KS + jb * KS + kk) * 64 + lane];
      asm volatile("" : "+a"(v));
      qf[jb][kk] = v;
    }
  __syncthreads();
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float fm = 0.f;
  const uint32_t slots_s = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int ph = 0; ph < phases; ++ph) {
    if (V >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if (V >= 1) __builtin_amdgcn_s_barrier();
    const uint32_t sa = slots_s + (uint32_t)(ph % 3) * (FRAGS * 1024) + lane_off;
    const uint32_t nd = slots_s + (uint32_t)((ph + 2) % 3) * (FRAGS * 1024) + (uint32_t)wave * 12u * 1024u;
    run_step<FRAGS>([](int n) constexpr { return (n & 1) * KS + (n >> 1); }, sa, [&](auto Nc, const rf_u32x4& af) __attribute__((always_inline)) {
      constexpr int n = decltype(Nc)::value, blk = n & 1, kk = n >> 1;
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[blk][0]) : "v"(af), "a"(qf[0][kk]));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[blk][1]) : "v"(af), "a"(qf[1][kk]));
      if constexpr (V >= 2 && n >= 2 && n < 34) {
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(fm) : "v"(acc[blk ^ 1][0][(n >> 1) & 15]), "v"(acc[blk ^ 1][1][(n >> 1) & 15]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(fm) : "v"(acc[blk ^ 1][0][(n >> 2) & 15]), "v"(acc[blk ^ 1][1][(n >> 2) & 15]));
      }
      if constexpr (V >= 3 && (n & 3) == 3)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt"
                     :: "s"(nd + (uint32_t)(n >> 2) * 1024u), "v"(lane_off), "s"((const char*)w + (size_t)(n >> 2) * 1024) : "memory");
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  if (V >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
  float sum = fm;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = sum;
  if (lane == 0) {
    float* o = stamps + (blockIdx.x * 4 + wave) * 2;
    o[0] = (float)(c1 - c0);
    o[1] = (float)(r1 - r0);
  }
}

template <int V>
static void run(const char* name, const u32x4* w, const u32x4* x, float* out, float* stamps, int cus, int phases) {
  const size_t ldsb = (size_t)3 * FRAGS * 1024;
  (void)hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((k<V>), dim3(cus), dim3(256), ldsb, 0, w, x, out, stamps, phases);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<V>), dim3(cus), dim3(256), ldsb, 0, w, x, out, stamps, phases);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<float> h((size_t)cus * 4 * 2);
  (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int i = 0; i < cus * 4; ++i) {
    clk.push_back(h[2 * i] / h[2 * i + 1] * 0.1);
    cyc.push_back(h[2 * i] / ((double)phases * FRAGS * 2));
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  const double flop = (double)reps * cus * 4 * (double)phases * FRAGS * 2 * 32768.0;
  printf("%-52s %7.1f TFLOP/s  clock %.2f GHz  %.1f cycles per MFMA\n", name, flop / (ms * 1e-3) / 1e12, clk[clk.size() / 2], cyc[cyc.size() / 2]);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const size_t nw = (size_t)FRAGS * 64, nx = (size_t)cus * 4 * 2 * KS * 64;
  std::vector<uint16_t> hw(nw * 8), hx(nx * 8);
  srand(7);
  auto rnd_half = []() { return (uint16_t)(((rand() & 1) << 15) | ((10 + rand() % 5) << 10) | (rand() & 1023)); };
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
  const int phases = 4000;
  printf("%s, %d CUs: the k_scan_w64 loop piece by piece (96 MFMAs per phase and wave)\n", p.gcnArchName, cus);
  run<0>("V0 MFMAs + pair-ring LDS reads", w, x, out, stamps, cus, phases);
  run<1>("V1 + barrier per phase", w, x, out, stamps, cus, phases);
  run<2>("V2 + 64 v_max3 per phase", w, x, out, stamps, cus, phases);
  run<3>("V3 + 12 LDS-DMA pieces per phase and wave (cached KiB)", w, x, out, stamps, cus, phases);
  return 0;
}
