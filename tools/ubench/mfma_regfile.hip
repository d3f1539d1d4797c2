// Micro-benchmark (diagnostic, not product): cycles per v_mfma_f32_32x32x16_f16 by the register file each operand
// sits in -- accumulator (C / D) in the vector half or the accumulator half, B in the vector half or the accumulator
// half; A always fresh from LDS (ds_read_b128), four rotating accumulation chains, one wave per SIMD.
// (k_scan_w64 keeps its B operands -- the queries -- in the accumulator half and its accumulators in the vector half.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_regfile tools/ubench/mfma_regfile.hip && ./mfma_regfile
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define FRAGS 48
#define NB 8

template <int ACC_A, int B_A>   // 1 = accumulator half
__global__ void __launch_bounds__(256, 1) k(const u32x4* __restrict__ w, const u32x4* __restrict__ x, float* out, float* stamps, int steps) {
  __shared__ u32x4 lds[FRAGS * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < FRAGS * 64; i += 256) lds[i] = w[i];
  u32x4 b[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    b[i] = x[(blockIdx.x * 4 + (tid >> 6)) * NB * 64 + i * 64 + lane];
    if (B_A) asm volatile("" : "+a"(b[i]));
    else asm volatile("" : "+v"(b[i]));
  }
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (ACC_A) asm volatile("" : "+a"(acc[i]));
    else asm volatile("" : "+v"(acc[i]));
  }
  const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // fragments are read four ahead in program order (the volatile asm statements are scheduling barriers: a read placed
  // next to its MFMA would expose the whole LDS latency, 64 cycles per MFMA)
  u32x4 ring[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) ring[f] = lds[f * 64 + lane];
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int f = 0; f < FRAGS; ++f) {
      const u32x4 a = ring[f & 3];
      asm volatile("" :: "v"(a));
      ring[f & 3] = lds[((f + 4) % FRAGS) * 64 + lane];
      if (ACC_A && B_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[f & 3]) : "v"(a), "a"(b[f % NB]));
      else if (ACC_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[f & 3]) : "v"(a), "v"(b[f % NB]));
      else if (B_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[f & 3]) : "v"(a), "a"(b[f % NB]));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[f & 3]) : "v"(a), "v"(b[f % NB]));
    }
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_nop 7\n\ts_nop 7");
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x16 t = acc[i];
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += t[r];
  }
  out[blockIdx.x * 256 + tid] = sum;
  if (lane == 0) {
    float* o = stamps + (blockIdx.x * 4 + (tid >> 6)) * 2;
    o[0] = (float)(c1 - c0);
    o[1] = (float)(r1 - r0);
  }
}

template <int ACC_A, int B_A>
static void run(const char* name, const u32x4* w, const u32x4* x, float* out, float* stamps, int cus, int steps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((k<ACC_A, B_A>), dim3(cus), dim3(256), 0, 0, w, x, out, stamps, steps);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<ACC_A, B_A>), dim3(cus), dim3(256), 0, 0, w, x, out, stamps, steps);
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
  const double flop = (double)reps * cus * 4 * (double)steps * FRAGS * 32768.0;
  printf("%-46s %7.1f TFLOP/s  clock %.2f GHz  %.1f cycles per MFMA\n", name, flop / (ms * 1e-3) / 1e12, clk[clk.size() / 2], cyc[cyc.size() / 2]);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const size_t nw = (size_t)FRAGS * 64, nx = (size_t)cus * 4 * NB * 64;
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
  const int steps = 8000;
  printf("%s, %d CUs, one wave per SIMD, A from LDS, random fp16 operands, inline-asm v_mfma_f32_32x32x16_f16\n", p.gcnArchName, cus);
  run<0, 0>("C/D vector half, B vector half", w, x, out, stamps, cus, steps);
  run<0, 1>("C/D vector half, B accumulator half (w64)", w, x, out, stamps, cus, steps);
  run<1, 0>("C/D accumulator half, B vector half", w, x, out, stamps, cus, steps);
  run<1, 1>("C/D accumulator half, B accumulator half", w, x, out, stamps, cus, steps);
  return 0;
}
