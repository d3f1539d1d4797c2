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
#include <type_traits>

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

// ---- compile-time loops ---------------------------------------------------------------------------
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}


// ---- LDS fragment reads of a ring step: pairs of fragments, four register pairs, three pairs ahead ----
// One ds_read_b128 per MFMA, all four waves of the CU: the LDS runs at half its rate and a read issued four
// MFMAs (128 cycles) ahead is late (stamps: 60 cycles per MFMA in the out-projection steps with groups of four
// read one group ahead).  Pair g (MFMAs 2 g, 2 g + 1) is read while pair g - 3 computes: six MFMAs of lead,
// the same 32 registers.  run_step<NM>(frag_of, sa, body): frag_of(n) = fragment of MFMA n (a constexpr
// callable), body(n_c, fragment registers) issues MFMA n and whatever rides in its gap.
template <int N>
__device__ __forceinline__ void lds_wait_pair(rf_u32x4 (&d)[2]) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(d[0]), "+v"(d[1]) : "n"(N));
}
template <int NM, int AHEAD = 3, bool NOWAIT = false, class FragOf, class Body>
__device__ __forceinline__ void run_step(FragOf frag_of, uint32_t sa, Body&& body) {
  constexpr int NP = NM / 2;   // pairs
  constexpr int NB = AHEAD + 1;
  rf_u32x4 fa[NB][2];
  auto read_pair = [&](auto Gc) __attribute__((always_inline)) {
    constexpr int G = decltype(Gc)::value;
    lds_read_frag<frag_of(2 * G)>(fa[G % NB][0], sa);
    lds_read_frag<frag_of(2 * G + 1)>(fa[G % NB][1], sa);
  };
  static_for<0, (NP < AHEAD ? NP : AHEAD)>([&](auto Gc) __attribute__((always_inline)) { read_pair(Gc); });
  static_for<0, NP>([&](auto Gc) __attribute__((always_inline)) {
    constexpr int G = decltype(Gc)::value;
    // pair G + AHEAD is read BEHIND the pair's first MFMA (its two issue slots sit in that MFMA's shadow, not in front
    // of it together with the previous gap's vector work and LDS-DMA piece): pairs read after pair G at its wait = AHEAD - 1
    constexpr int newer = (NP - 1 - G) < (AHEAD - 1) ? (NP - 1 - G) : (AHEAD - 1);
    if constexpr (NOWAIT) asm volatile("" : "+v"(fa[G % NB][0]), "+v"(fa[G % NB][1]));
    else lds_wait_pair<2 * newer>(fa[G % NB]);
    body(std::integral_constant<int, 2 * G>{}, fa[G % NB][0]);
    if constexpr (G + AHEAD < NP) read_pair(std::integral_constant<int, G + AHEAD>{});
    body(std::integral_constant<int, 2 * G + 1>{}, fa[G % NB][1]);
  });
}
