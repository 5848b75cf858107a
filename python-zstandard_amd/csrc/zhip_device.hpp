// zhip_device.hpp -- the few wave-level primitives the kernels are written against.
//
// Product build (hipcc, gfx950): thin inline wrappers over HIP/AMDGCN builtins. 64-lane wavefronts,
// one wavefront per workgroup (blockDim.x == 64), so __syncthreads() is a wave-local LDS fence.
//
// ZHIP_EMU build (g++, tests/emu only): the same kernel source runs on the host with each lane as a ucontext fiber and every
// collective implemented as a rendezvous, so kernel LOGIC can be debugged in a container without a GPU. That implementation of this
// interface lives in tests/emu/zhip_device_emu.hpp; it is compiled only by tests/emu/build.sh, is never part of libzstd_hip.so and is
// not reachable from the python package.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifndef ZHIP_EMU
// =====================================================================================  device
#include <hip/hip_runtime.h>
#define ZH_DEV __device__ __forceinline__
#ifdef ZD_NOINLINE
#define ZH_DEVFN __device__ __attribute__((noinline))
#else
#define ZH_DEVFN __device__
#endif
#define ZH_COLD __device__ __attribute__((noinline))     // once-per-frame routines: their own register budget, kept out of the callers' hot loops
#define ZH_GLOBAL extern "C" __global__
#define ZH_SHARED __shared__
#define ZH_CONST __device__ const
// a pointer the caller KNOWS points into LDS, typed so: reads through it are ds_read, not flat loads (a noinline routine sees its LDS arguments as generic pointers, and where the
// compiler cannot infer the address space back it emits flat_load, which waits on both counters)
#define ZH_LDS_CPTR(type, p) ((const __attribute__((address_space(3))) type*)(p))

ZH_DEV uint32_t zh_lane() { return threadIdx.x; }
ZH_DEV uint32_t zh_block() { return blockIdx.x; }
ZH_DEV uint32_t zh_nblocks() { return gridDim.x; }
ZH_DEV void zh_sync() { __syncthreads(); }
// LDS hand-off between lanes of ONE wave: the wave's LDS instructions execute in issue order, so all this has to do is keep the compiler
// from moving LDS accesses across it (no s_barrier, no vmcnt wait -- __syncthreads() also waits for the wave's global stores)
ZH_DEV void zh_wave_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
ZH_DEV uint64_t zh_ballot(bool p) { return __ballot(p); }
ZH_DEV uint32_t zh_shfl(uint32_t v, uint32_t srcLane) { return (uint32_t)__shfl((int)v, (int)srcLane, 64); }
ZH_DEV uint32_t zh_shfl_up(uint32_t v, uint32_t d) { return (uint32_t)__shfl_up((int)v, d, 64); }
ZH_DEV uint32_t zh_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// lane `l`'s value to every lane, l wave-uniform: v_readlane_b32 (a scalar result -- no LDS crossbar trip as zh_shfl's ds_bpermute)
ZH_DEV uint32_t zh_bcast(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// lane K of the caller's quad (lanes 4q .. 4q+3), to all four: a DPP quad_perm operand, no LDS crossbar trip
template <int K> ZH_DEV uint32_t zh_quad(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, K * 0x55, 0xf, 0xf, false); }
// v + (v of quad lane CTRL[2r+1:2r] for lane r of the quad): one v_add_u32_dpp
// (r03f: written as update_dpp(0, v, CTRL, 0xf, 0xf, bound_ctrl) LLVM's DPP combiner folds 7 of the 32 quad moves of K2's unrolled body into their
// consumers -- 320 -> 313 instructions -- but the K2 built that way reports corrupt streams on the MI355X where the emulator and this form
// agree with libzstd: not adopted)
template <int CTRL> ZH_DEV uint32_t zh_quad_add(uint32_t acc, uint32_t v) { return acc + (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, false); }
ZH_DEV uint32_t zh_atomic_inc(uint32_t* p) { return atomicAdd(p, 1u); }
ZH_DEV uint32_t zh_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
ZH_DEV void zh_atomic_add64(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
ZH_DEV void zh_atomic_max(uint32_t* p, uint32_t v) { atomicMax(p, v); }
ZH_DEV void zh_atomic_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
ZH_DEV void zh_lds_atomic_or(uint32_t* p, uint32_t v) { atomicOr(p, v); }
ZH_DEV void zh_lds_atomic_inc(uint32_t* p) { atomicAdd(p, 1u); }
ZH_DEV uint32_t zh_lds_atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
ZH_DEV uint32_t zh_wave_max(uint32_t v) { for (int d = 32; d; d >>= 1) { uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64); v = o > v ? o : v; } return v; }
ZH_DEV void ze_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
// hides a value's provenance from the optimizer (used so `lane == 0` is not provably loop-invariant)
ZH_DEV uint32_t zh_opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
ZH_DEV uint64_t zh_opaque64(uint64_t v) { asm volatile("" : "+v"(v)); return v; }   // also pins a load: it cannot sink below this point
ZH_DEV int zh_popc64(uint64_t v) { return __popcll(v); }
ZH_DEV int zh_ctz64(uint64_t v) { return __ffsll((unsigned long long)v) - 1; }   // v != 0
ZH_DEV int zh_clz64(uint64_t v) { return __clzll((long long)v); }                  // v != 0
ZH_DEV int zh_highbit32(uint32_t v) { return 31 - __clz((int)v); }               // v != 0
// the machine scheduler moves nothing across this point (hand-placed software pipelining stays where it was put)
#define ZH_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// the four registers are considered read here: their values (and so their registers) live on to this point
#define ZH_KEEP4(a, b, c, d) asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d))
// v_bfe_u32: (v >> (off & 31)) & ((1 << (width & 31)) - 1); width 0 gives 0 whatever off is
ZH_DEV uint32_t zh_bfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }
// v_alignbit_b32: low 32 bits of ((hi:lo) >> (sh & 31))
ZH_DEV uint32_t zh_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

#else
#include "zhip_device_emu.hpp"     // tests/emu/ (found through its -I): the host wave emulator's implementation of the interface above
#endif

// ------------------------------------------------------------------------------------- common helpers
typedef uint64_t __attribute__((aligned(1))) zh_u64u;
typedef uint32_t __attribute__((aligned(1))) zh_u32u;
typedef uint16_t __attribute__((aligned(1))) zh_u16u;
struct __attribute__((packed, aligned(1))) zh_v16 { uint64_t lo, hi; };      // 16 bytes at any address: one global_load_dwordx4
ZH_DEV zh_v16 zh_ld128(const uint8_t* p) { return *(const zh_v16*)p; }
ZH_DEV uint64_t zh_ld64(const uint8_t* p) { return *(const zh_u64u*)p; }
ZH_DEV uint32_t zh_ld32(const uint8_t* p) { return *(const zh_u32u*)p; }
ZH_DEV uint32_t zh_ld16(const uint8_t* p) { return *(const zh_u16u*)p; }
ZH_DEV uint32_t zh_ld24(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
ZH_DEV void zh_st64(uint8_t* p, uint64_t v) { *(zh_u64u*)p = v; }
ZH_DEV void zh_st32(uint8_t* p, uint32_t v) { *(zh_u32u*)p = v; }
ZH_DEV void zh_st16(uint8_t* p, uint16_t v) { *(zh_u16u*)p = v; }
ZH_DEV uint64_t zh_lt_mask() { return (1ull << zh_lane()) - 1; }
// streaming ("nt") loads: data read once -- sequences, literals -- should not push a frame's recent OUTPUT out of its XCD's L2, which is
// what far matches want to find there (r03a: random 16-byte gathers run at 230 G/s from L2 and at 40-50 G/s from anywhere behind it)
#ifndef ZHIP_EMU
typedef uint32_t __attribute__((ext_vector_type(4))) zh_v4raw;
typedef zh_v4raw __attribute__((aligned(1))) zh_v4rawu;
ZH_DEV uint64_t zh_ld64_nt(const uint8_t* p) { return __builtin_nontemporal_load((const zh_u64u*)p); }
ZH_DEV zh_v16 zh_ld128_nt(const uint8_t* p) { const zh_v4raw v = __builtin_nontemporal_load((const zh_v4rawu*)p); zh_v16 r; r.lo = (uint64_t)v.x | ((uint64_t)v.y << 32); r.hi = (uint64_t)v.z | ((uint64_t)v.w << 32); return r; }
ZH_DEV uint64_t zh_ldq_nt(const uint64_t* p) { return __builtin_nontemporal_load(p); }
#else
ZH_DEV uint64_t zh_ld64_nt(const uint8_t* p) { return zh_ld64(p); }
ZH_DEV zh_v16 zh_ld128_nt(const uint8_t* p) { return zh_ld128(p); }
ZH_DEV uint64_t zh_ldq_nt(const uint64_t* p) { return *p; }
#endif

// inclusive wave prefix sum (all 64 lanes must call)
#ifndef ZHIP_EMU
// six DPP adds: Hillis-Steele inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the row totals carried across rows with the
// gfx9 row broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3). Lanes without a source add 0.
ZH_DEV uint32_t zh_scan_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
#endif
