// Microbenchmark: HBM streaming-read ceiling on gfx950, register loads vs LDS-DMA.
//   hipcc --offload-arch=gfx950 -O3 -o stream_bench stream_bench.hip && ./stream_bench
// Every variant reads the same 768 MB buffer once per launch (the corpus of the headline
// search config), in 1-KiB wave pieces, with a fixed number of pieces in flight per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A: global_load_dwordx4 (nt) into a register ring of DEPTH pieces; wave w of the grid takes
// chunks of CH pieces: chunk index w, w + nwaves, ...
template <int DEPTH, int CH, bool NT>
__global__ void __launch_bounds__(256, 2) k_reg(const uint4* src, size_t n_pieces, uint32_t* sink) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t nchunks = n_pieces / CH;
  u32x4 ring[DEPTH];
  u32x4 acc = {0, 0, 0, 0};
  // flatten this wave's pieces: piece t -> chunk (wave + (t / CH) * nwaves), offset t % CH
  const size_t my_chunks = nchunks > wave ? (nchunks - wave + nwaves - 1) / nwaves : 0;
  const size_t my = my_chunks * CH;
  auto addr = [&](size_t t) {
    const size_t tc = t < my ? t : my - 1;
    return src + ((wave + (tc / CH) * nwaves) * CH + tc % CH) * 64 + lane;
  };
  if (my == 0) return;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    ring[d] = NT ? __builtin_nontemporal_load((const u32x4*)addr(d)) : *(const u32x4*)addr(d);
  for (size_t t = 0; t < my; t += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc ^= ring[d];
      ring[d] = NT ? __builtin_nontemporal_load((const u32x4*)addr(t + d + DEPTH)) : *(const u32x4*)addr(t + d + DEPTH);
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

// B: LDS-DMA into a private per-wave LDS ring of DEPTH pieces (never read back)
template <int DEPTH, int CH, int AUX, int WAVES, int WGS>
__global__ void __launch_bounds__(WAVES * 64, WGS) k_dma(const uint4* src, size_t n_pieces) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* ring = (u32x4*)smem + (size_t)(threadIdx.x >> 6) * DEPTH * 64;
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * WAVES;
  const size_t nchunks = n_pieces / CH;
  const size_t my_chunks = nchunks > wave ? (nchunks - wave + nwaves - 1) / nwaves : 0;
  const size_t my = my_chunks * CH;
  if (my == 0) return;
  auto addr = [&](size_t t) { return src + ((wave + (t / CH) * nwaves) * CH + t % CH) * 64 + lane; };
  size_t t = 0;
  for (; t + DEPTH <= my; t += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      // keep DEPTH pieces in flight: wait until at most DEPTH - 1 are outstanding, then reuse slot d
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr(t + d),
                                       (__attribute__((address_space(3))) void*)(ring + d * 64), 16, 0, AUX);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static float time_it(void (*launch)(), int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  launch();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

static uint4* g_src;
static uint32_t* g_sink;
static size_t g_pieces;
#define REG(DEPTH, CH, NT, GRID)                                                                       \
  {                                                                                                    \
    float ms = time_it([] { hipLaunchKernelGGL((k_reg<DEPTH, CH, NT>), dim3(GRID), dim3(256), 0, 0, g_src, g_pieces, g_sink); }, 20); \
    printf("reg  depth %2d chunk %2d %s grid %4d: %7.1f us  %6.0f GB/s\n", DEPTH, CH, NT ? "nt " : "def", GRID, ms * 1e3, g_pieces * 1024.0 / ms / 1e6); \
  }
#define DMA(DEPTH, CH, AUX, WAVES, WGS)                                                                \
  {                                                                                                    \
    auto kern = k_dma<DEPTH, CH, AUX, WAVES, WGS>;                                                     \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES * DEPTH * 1024); \
    float ms = time_it([] { hipLaunchKernelGGL((k_dma<DEPTH, CH, AUX, WAVES, WGS>), dim3(256 * WGS), dim3(WAVES * 64), WAVES * DEPTH * 1024, 0, g_src, g_pieces); }, 20); \
    printf("dma  depth %2d chunk %2d %s waves/CU %2d: %7.1f us  %6.0f GB/s\n", DEPTH, CH, AUX ? "nt " : "def", WAVES * WGS, ms * 1e3, g_pieces * 1024.0 / ms / 1e6); \
  }

int main() {
  g_pieces = 750000;   // 768 MB in 1-KiB pieces
  (void)hipMalloc(&g_src, g_pieces * 1024 + (1 << 20));
  (void)hipMalloc(&g_sink, 64);
  (void)hipMemset(g_src, 1, g_pieces * 1024 + (1 << 20));
  REG(8, 24, true, 512) REG(8, 24, false, 512) REG(16, 24, true, 512) REG(24, 24, true, 512)
  REG(8, 24, true, 768) REG(12, 24, true, 768) REG(8, 12, true, 512) REG(8, 48, true, 512)
  DMA(8, 24, 2, 8, 1) DMA(8, 24, 0, 8, 1) DMA(16, 24, 2, 8, 1) DMA(12, 24, 2, 8, 1)
  DMA(16, 24, 2, 4, 1) DMA(24, 24, 2, 4, 1) DMA(32, 24, 2, 4, 1) DMA(8, 24, 2, 4, 2) DMA(16, 24, 2, 4, 2)
  DMA(16, 48, 2, 4, 1) DMA(8, 12, 2, 8, 1) DMA(4, 24, 2, 8, 2)
  return 0;
}
