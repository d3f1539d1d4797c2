// Inline-asm LDS fragment reads with hand-counted lgkmcnt waits, shared by the kernels that
// stream one MFMA operand through an LDS-DMA ring (scan_wide.hip: corpus blocks; encoder.hip:
// weight blocks).  A ring slot holds 1-KiB fragments (64 lanes x 16 B, lane-linear: what one
// global_load_lds_dwordx4 wave-instruction writes and one ds_read_b128 reads back).
//
// Why asm: as compiler-visible LDS loads each read gets an s_waitcnt vmcnt(0) from hipcc (it
// cannot tell the read from the LDS-DMA writes still in flight), which drains the prefetch ring
// at every phase.  The reads are issued in groups of WL_GRP; lds_wait_group<N> waits until at
// most N reads are outstanding and ties the group's registers to the wait, so the MFMAs that
// consume them cannot be scheduled above it.
#pragma once
#include <stdint.h>

#ifndef RF_U32X4_DEFINED
#define RF_U32X4_DEFINED
typedef uint32_t rf_u32x4 __attribute__((ext_vector_type(4)));
#endif

#define WL_GRP 4   // fragments per LDS read group

// `addr` is the lane's byte address of fragment 0 of the slot; fragment f sits 1 KiB * f
// further on (immediate offset, < 64 KiB)
template <int F>
__device__ __forceinline__ void lds_read_frag(rf_u32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(F * 1024));
}
template <int F0>
__device__ __forceinline__ void lds_read_group(rf_u32x4 (&d)[WL_GRP], uint32_t addr) {
  lds_read_frag<F0 + 0>(d[0], addr);
  lds_read_frag<F0 + 1>(d[1], addr);
  lds_read_frag<F0 + 2>(d[2], addr);
  lds_read_frag<F0 + 3>(d[3], addr);
}
__device__ __forceinline__ void lds_read_group_dyn(rf_u32x4 (&d)[WL_GRP], uint32_t addr, int f0) {
  // f0 is a compile-time constant after unrolling; dispatch to the immediate-offset forms
  switch (f0) {
#define WL_CASE(x) case x: lds_read_group<x>(d, addr); break;
    WL_CASE(0) WL_CASE(4) WL_CASE(8) WL_CASE(12) WL_CASE(16) WL_CASE(20) WL_CASE(24) WL_CASE(28)
    WL_CASE(32) WL_CASE(36) WL_CASE(40) WL_CASE(44)
#undef WL_CASE
    default: break;
  }
}
template <int N>
__device__ __forceinline__ void lds_wait_group(rf_u32x4 (&d)[WL_GRP]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "n"(N));
}
