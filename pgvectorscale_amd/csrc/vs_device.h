// vs_device.h — device helpers shared by the gfx950 kernels (wave64).
#pragma once
#include "vs_internal.h"

#define WAVE 64

// ---------------------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// add the value held by the lane with index (lane ^ 1) / (lane ^ 2) via DPP quad_perm (no LDS traffic)
__device__ __forceinline__ uint32_t quad_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true);
    return v;
}

// 16 bytes of a code row read once per scan: a non-temporal load (global_load_dwordx4 ... nt), so the stream of gathered
// rows does not push the scans' own working sets (dedup tables, heap spill) out of L2
// (the address space is spelled out: a pointer that went through in_vgpr() — or any other opaque step — is a generic one to the compiler,
// and its load a FLAT load, which also counts against the LDS counter and waits for it: round 6 found every stream load of
// k_search_fast compiled that way)
typedef __attribute__((address_space(1))) __uint128_t vs_glb_u128;
typedef __attribute__((address_space(1))) uint64_t vs_glb_u64;
typedef __attribute__((address_space(1))) uint32_t vs_glb_u32;
__device__ __forceinline__ ulonglong2 load_stream16(const uint64_t* p) {
    const __uint128_t v = __builtin_nontemporal_load((const vs_glb_u128*)p);
    return make_ulonglong2((unsigned long long)v, (unsigned long long)(v >> 64));
}

// the same for data a scan reads once and never again: its visit's neighbor row (and the neighbors' label masks next to it), a heap tid
__device__ __forceinline__ uint32_t load_stream32(const uint32_t* p) { return __builtin_nontemporal_load((const vs_glb_u32*)p); }
__device__ __forceinline__ uint64_t load_stream64(const uint64_t* p) { return __builtin_nontemporal_load((const vs_glb_u64*)p); }

// per 16-bit half: min(a, b) (v_pk_min_u16).  Written with the GCC vector extension so that the host build of the test interpreter
// compiles it too.
typedef unsigned short vs_u16x2 __attribute__((vector_size(4)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    vs_u16x2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const vs_u16x2 m = x < y ? x : y;
    uint32_t r;
    __builtin_memcpy(&r, &m, 4);
    return r;
}
// A wave-uniform value the compiler is told nothing about, held in a VECTOR register: scalar registers are what k_search_fast runs
// out of (106 with 226 spilled in round 5), vector registers it has to spare, and a vector instruction takes its operand from
// either file at the same cost — so the loop-invariant constants that only ever feed vector instructions (array base pointers, hash
// masks) are parked there by hand instead of being re-read from the kernel arguments inside the loop (s_load + s_waitcnt).
template <class T>
__device__ __forceinline__ T in_vgpr(T x) {
    asm volatile("" : "+v"(x));
    return x;
}
// a wave-uniform 64-bit lane mask as a per-lane condition (no vector instruction: the mask goes straight into exec)
__device__ __forceinline__ bool lane_of(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// Hamming distance of one code row against the query code held in LDS, computed by a group of 4 lanes
// (lane l4 covers words 2*l4 + 8t, 2*l4+1 + 8t: 16 B per lane per step, 64 B contiguous per group per step).
// Rows are code_stride (even) words, zero padded, so the padded tail contributes popcount(0^0)=0.
__device__ __forceinline__ uint32_t ham_row4(const uint64_t* __restrict__ row, const uint64_t* qc, int l4,
                                             uint32_t code_stride, bool active) {
    uint32_t acc = 0;
    if (active) {
        for (uint32_t w = 2u * (uint32_t)l4; w < code_stride; w += 8) {
            const ulonglong2 r = *reinterpret_cast<const ulonglong2*>(row + w);
            const ulonglong2 qq = *reinterpret_cast<const ulonglong2*>(qc + w);
            acc += (uint32_t)__popcll(r.x ^ qq.x) + (uint32_t)__popcll(r.y ^ qq.y);
        }
    }
    return quad_sum(acc);
}

