// zhip_device_emu.hpp -- the host wave emulator's implementation of the wave-level primitives of python-zstandard_amd/csrc/zhip_device.hpp
// (included from there under -DZHIP_EMU; test infrastructure, never compiled into the product library).
#pragma once
// =====================================================================================  emulation
#include <string.h>
#define ZH_DEV static inline
#define ZH_DEVFN
#define ZH_COLD
#define ZH_GLOBAL extern "C"
#define ZH_SHARED static
#define ZH_CONST static const
#define ZH_LDS_CPTR(type, p) ((const type*)(p))

namespace zhemu {
void collective_wait();                 // rendezvous of all live lanes of the current wave
extern thread_local uint32_t lane, block, nblocks;
extern thread_local uint64_t slot[64];  // per-lane exchange slots
extern thread_local uint64_t result;
typedef void (*lane_fn)(void*);
void run_grid(uint32_t nBlocks, lane_fn fn, void* arg);   // runs fn(arg) on 64 fibers per block
}
ZH_DEV uint32_t zh_lane() { return zhemu::lane; }
ZH_DEV uint32_t zh_block() { return zhemu::block; }
ZH_DEV uint32_t zh_nblocks() { return zhemu::nblocks; }
ZH_DEV void zh_sync() { zhemu::collective_wait(); }
ZH_DEV void zh_wave_fence() { zhemu::collective_wait(); }
ZH_DEV uint64_t zh_ballot(bool p)
{
    zhemu::slot[zhemu::lane] = p ? 1 : 0;
    zhemu::collective_wait();
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) m |= (uint64_t)(zhemu::slot[i] & 1) << i;
    zhemu::collective_wait();
    return m;
}
ZH_DEV uint32_t zh_shfl(uint32_t v, uint32_t srcLane)
{
    zhemu::slot[zhemu::lane] = v;
    zhemu::collective_wait();
    uint32_t r = (uint32_t)zhemu::slot[srcLane & 63];
    zhemu::collective_wait();
    return r;
}
ZH_DEV uint32_t zh_shfl_up(uint32_t v, uint32_t d)
{
    zhemu::slot[zhemu::lane] = v;
    zhemu::collective_wait();
    uint32_t r = zhemu::lane >= d ? (uint32_t)zhemu::slot[zhemu::lane - d] : v;
    zhemu::collective_wait();
    return r;
}
ZH_DEV uint32_t zh_first(uint32_t v) { return zh_shfl(v, 0); }
ZH_DEV uint32_t zh_bcast(uint32_t v, uint32_t l) { return zh_shfl(v, l); }
template <int K> ZH_DEV uint32_t zh_quad(uint32_t v) { return zh_shfl(v, (zhemu::lane & ~3u) | (uint32_t)K); }
template <int CTRL> ZH_DEV uint32_t zh_quad_add(uint32_t acc, uint32_t v) { return acc + zh_shfl(v, (zhemu::lane & ~3u) | ((CTRL >> (2 * (zhemu::lane & 3))) & 3)); }
ZH_DEV uint32_t zh_atomic_inc(uint32_t* p) { return __sync_fetch_and_add(p, 1u); }
ZH_DEV uint32_t zh_atomic_add(uint32_t* p, uint32_t v) { return __sync_fetch_and_add(p, v); }
ZH_DEV void zh_atomic_add64(unsigned long long* p, unsigned long long v) { __sync_fetch_and_add(p, v); }
ZH_DEV void zh_atomic_max(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
ZH_DEV void zh_atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
ZH_DEV void zh_lds_atomic_or(uint32_t* p, uint32_t v) { *p |= v; }
ZH_DEV uint32_t zh_opaque(uint32_t v) { return v; }
ZH_DEV uint64_t zh_opaque64(uint64_t v) { return v; }
ZH_DEV void zh_lds_atomic_inc(uint32_t* p) { (*p)++; }
ZH_DEV uint32_t zh_lds_atomic_add(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
ZH_DEV uint32_t zh_wave_max(uint32_t v)
{
    zhemu::slot[zhemu::lane] = v;
    zhemu::collective_wait();
    uint32_t m = 0;
    for (int i = 0; i < 64; i++) if ((uint32_t)zhemu::slot[i] > m) m = (uint32_t)zhemu::slot[i];
    zhemu::collective_wait();
    return m;
}
ZH_DEV void ze_fence() { zhemu::collective_wait(); }
#define ZH_SCHED_FENCE() do { } while (0)
#define ZH_KEEP4(a, b, c, d) do { } while (0)
ZH_DEV int zh_popc64(uint64_t v) { return __builtin_popcountll(v); }
ZH_DEV int zh_ctz64(uint64_t v) { return __builtin_ctzll(v); }
ZH_DEV int zh_clz64(uint64_t v) { return __builtin_clzll(v); }
ZH_DEV int zh_highbit32(uint32_t v) { return 31 - __builtin_clz(v); }
ZH_DEV uint32_t zh_bfe(uint32_t v, uint32_t off, uint32_t width) { width &= 31; return (v >> (off & 31)) & ((1u << width) - 1); }
ZH_DEV uint32_t zh_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31)); }
// inclusive wave prefix sum (the device does it in six DPP adds)
ZH_DEV uint32_t zh_scan_add(uint32_t v)
{
    for (uint32_t d = 1; d < 64; d <<= 1) {
        uint32_t t = zh_shfl_up(v, d);
        if (zh_lane() >= d) v += t;
    }
    return v;
}
