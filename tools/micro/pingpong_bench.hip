// Microbenchmark: does sweeping the 768 MB corpus alternately forwards and backwards let the
// 256 MiB Infinity Cache serve the part of the buffer the previous sweep read last?
//   hipcc --offload-arch=gfx950 -O3 -o pingpong_bench pingpong_bench.hip && ./pingpong_bench
// Same access pattern as stream_bench's register variant (1-KiB wave pieces, chunks of 24 pieces
// strided over the waves, 8 pieces in flight per wave); `reverse` mirrors the chunk order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int CH, bool NT>
__global__ void __launch_bounds__(256, 2) k_sweep(const uint4* src, size_t n_pieces, int reverse, uint32_t* sink) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t nchunks = n_pieces / CH;
  u32x4 ring[DEPTH];
  u32x4 acc = {0, 0, 0, 0};
  const size_t my_chunks = nchunks > wave ? (nchunks - wave + nwaves - 1) / nwaves : 0;
  const size_t my = my_chunks * CH;
  auto addr = [&](size_t t) {
    const size_t tc = t < my ? t : my - 1;
    size_t chunk = wave + (tc / CH) * nwaves;
    if (reverse) chunk = nchunks - 1 - chunk;
    return src + (chunk * CH + tc % CH) * 64 + lane;
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

template <bool NT>
static void run(const uint4* src, size_t pieces, uint32_t* sink, bool pingpong, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int reps = 40;
  for (int i = 0; i < 4; ++i)
    hipLaunchKernelGGL((k_sweep<8, 24, NT>), dim3(512), dim3(256), 0, 0, src, pieces, pingpong ? (i & 1) : 0, sink);
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((k_sweep<8, 24, NT>), dim3(512), dim3(256), 0, 0, src, pieces, pingpong ? (i & 1) : 0, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%-28s %5.0f MB: %7.1f us  %6.0f GB/s\n", name, pieces * 1024.0 / 1e6, ms * 1e3, pieces * 1024.0 / ms / 1e6);
}

int main() {
  uint4* src;
  uint32_t* sink;
  const size_t max_pieces = 1500000;
  (void)hipMalloc(&src, max_pieces * 1024 + (1 << 20));
  (void)hipMalloc(&sink, 64);
  (void)hipMemset(src, 1, max_pieces * 1024 + (1 << 20));
  const size_t sizes[] = {75000, 187500, 375000, 750000, 1500000};   // 77, 192, 384, 768, 1536 MB
  for (size_t p : sizes) {
    run<true>(src, p, sink, false, "forward only, nt");
    run<true>(src, p, sink, true, "forward / backward, nt");
    run<false>(src, p, sink, false, "forward only, default");
    run<false>(src, p, sink, true, "forward / backward, default");
  }
  return 0;
}
