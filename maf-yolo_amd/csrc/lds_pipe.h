// LDS fragment reads as inline assembly with hand-counted waits, shared by the kernels whose MFMA operands come out of LDS.
//
// Written as plain loads, hipcc (ROCm 7.2) reads one or two fragments into the same register quads right in front of the MFMAs that need them and
// waits with lgkmcnt(1) / lgkmcnt(0): a full LDS round trip (~130 cycles) per one or two 17-cycle MFMAs (measured: 9.4k of the 9.6k cycles of a
// conv3s2_wreg tile, 4.7k instead of 1.2k for the 72 MFMAs of a 288 -> 128 conv1x1_stream_lds tile).  The pattern used instead: the read of
// step t + RD is issued before the MFMAs of step t; LDS operations return in order, so when step t is consumed at most min(RD, steps left)
// newer reads may still be in flight: `s_waitcnt lgkmcnt(that)`.  (Scalar loads share the counter; one in flight only makes a wait longer.)
#pragma once
#include "maf_common.h"
#include <type_traits>

template <int OFF> __device__ __forceinline__ void lp_ds_read_b128(u32x4_t& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
// "+v": whatever reads `a` cannot move above the wait
template <int N> __device__ __forceinline__ void lp_wait_lgkm(u32x4_t& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }

template <int N, int I = 0, typename F>
__device__ __forceinline__ void lp_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lp_static_for<N, I + 1>(f);
    }
}

__device__ __forceinline__ uint32_t lp_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
