// tablebench.hip -- the flat match kernel's table traffic in isolation (test infrastructure; result under profiles/): every lane owns a table of
// TBYTES bytes in one big buffer (like a frame's two hash tables) and runs a dependent chain of trips; a trip touches four pseudo-random
// 4-byte cells of the lane's table --   ldst : load each cell, then store a new value to it (what the search does: 4 loads + 4 stores)
//                                        xchg : one atomic exchange per cell (the same reads and writes as 4 requests instead of 8)
//                                        ld   : the loads alone                          st : the stores alone
//                                        + nontemporal forms of the stores / loads, and 2-byte cells (same cell count, half the bytes)
//                                        + round 6: the stores as whole 64-byte lines / 32-byte sectors, and the load-line / change-one-cell / store-line form
// Usage: tablebench [lanes = 65536] [table KiB = 384] [trips = 2000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(64) void tb(uint32_t* tab, uint32_t cells, uint32_t lanes, uint32_t trips, uint32_t* sink)
{
    const uint32_t l = blockIdx.x * 64 + threadIdx.x;
    if (l >= lanes) return;
    uint32_t* t = tab + (size_t)l * cells;
    uint32_t s = mix(l + 1), acc = 0;
    for (uint32_t i = 0; i < trips; i++) {
        uint32_t ix[4], v[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) { s = s * 1664525u + 1013904223u; ix[k] = mix(s) % cells; }
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = t[ix[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) t[ix[k]] = i + k;
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = atomicExch(t + ix[k], i + k);
        } else if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = t[ix[k]];
        } else if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 4; k++) t[ix[k]] = i + k;
        } else if (MODE == 4) {                         // nontemporal stores
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_nontemporal_store(i + k, t + ix[k]);
        } else if (MODE == 5) {                         // loads + nontemporal stores
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = t[ix[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_nontemporal_store(i + k, t + ix[k]);
        } else if (MODE == 6) {                         // nontemporal loads + nontemporal stores
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = __builtin_nontemporal_load(t + ix[k]);
#pragma unroll
            for (int k = 0; k < 4; k++) __builtin_nontemporal_store(i + k, t + ix[k]);
        } else if (MODE == 8) {                         // round 6: the stores as WHOLE 64-byte lines (4 x 16 bytes to the cell's line): no partial write-back
#pragma unroll
            for (int k = 0; k < 4; k++) { uint4* q = (uint4*)(t + (ix[k] & ~15u)); const uint4 w = make_uint4(i, k, i, k); q[0] = w; q[1] = w; q[2] = w; q[3] = w; }
        } else if (MODE == 9) {                         // whole 32-byte sectors
#pragma unroll
            for (int k = 0; k < 4; k++) { uint4* q = (uint4*)(t + (ix[k] & ~7u)); const uint4 w = make_uint4(i, k, i, k); q[0] = w; q[1] = w; }
        } else if (MODE == 10) {                        // what a search with line-wide cells would do: load the cell's 64-byte line, change one cell, store the line
            uint4 ln[4][4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint4* q = (const uint4*)(t + (ix[k] & ~15u)); ln[k][0] = q[0]; ln[k][1] = q[1]; ln[k][2] = q[2]; ln[k][3] = q[3]; }
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = ln[k][0].x ^ ln[k][1].y ^ ln[k][2].z ^ ln[k][3].w; uint4* q = (uint4*)(t + (ix[k] & ~15u)); ln[k][ix[k] & 3].x = i + k; q[0] = ln[k][0]; q[1] = ln[k][1]; q[2] = ln[k][2]; q[3] = ln[k][3]; }
        } else if (MODE == 11) {                        // the 64-byte loads alone
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint4* q = (const uint4*)(t + (ix[k] & ~15u)); const uint4 a = q[0], b = q[1], c = q[2], d = q[3]; v[k] = a.x ^ b.y ^ c.z ^ d.w; }
        } else if (MODE == 12) {                        // 4-byte loads + whole-line stores (isolates the write side of mode 10)
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = t[ix[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) { uint4* q = (uint4*)(t + (ix[k] & ~15u)); const uint4 w = make_uint4(i, k, i, k); q[0] = w; q[1] = w; q[2] = w; q[3] = w; }
        } else if (MODE >= 13 && MODE <= 16) {          // load the line / change one cell / store the line, with the cache policy varied: 13 loads nt, 14 stores nt, 15 both nt, 16 loads nt + DEPENDENT only through one dword
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            u4 ln[4][4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const u4* q = (const u4*)(t + (ix[k] & ~15u));
#pragma unroll
                for (int j = 0; j < 4; j++) ln[k][j] = (MODE == 14) ? q[j] : __builtin_nontemporal_load(q + j); }
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = ln[k][0].x ^ ln[k][1].y ^ ln[k][2].z ^ ln[k][3].w; u4* q = (u4*)(t + (ix[k] & ~15u)); ln[k][ix[k] & 3].x = i + k;
#pragma unroll
                for (int j = 0; j < 4; j++) { if (MODE == 13 || MODE == 16) q[j] = ln[k][j]; else __builtin_nontemporal_store(ln[k][j], q + j); } }
        } else {                                        // 2-byte cells: loads + stores of half the width
            uint16_t* t2 = (uint16_t*)t;
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = t2[ix[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) t2[ix[k]] = (uint16_t)(i + k);
        }
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        s += acc & 1;                                   // the next trip's cells depend on what this one read (as the search's do)
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <class F> static float timed(F f)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    f(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const uint32_t lanes = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536, kib = argc > 2 ? (uint32_t)atoi(argv[2]) : 384, trips = argc > 3 ? (uint32_t)atoi(argv[3]) : 2000;
    const uint32_t cells = kib * 256;
    uint32_t* tab; uint32_t* sink; CHECK(hipMalloc(&tab, (size_t)lanes * cells * 4)); CHECK(hipMemset(tab, 0, (size_t)lanes * cells * 4)); CHECK(hipMalloc(&sink, 64));
    const dim3 g((lanes + 63) / 64), b(64);
    const char* names[17] = {"ldst (4 loads + 4 stores)", "xchg (4 atomic exchanges)", "ld   (4 loads)", "st   (4 stores)", "st nontemporal", "ld + st nontemporal", "ld nt + st nt", "ldst, 2-byte cells (half the table)",
                             "st   whole 64-byte lines", "st   whole 32-byte sectors", "ldst whole 64-byte lines", "ld   whole 64-byte lines", "ld 4 bytes + st whole lines",
                             "ldst whole lines, loads nt", "ldst whole lines, stores nt", "ldst whole lines, both nt", "(as 13)"};
    for (int m = 0; m < 16; m++) {
        float ms = 0;
        if (m == 0) ms = timed([&] { hipLaunchKernelGGL(tb<0>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 1) ms = timed([&] { hipLaunchKernelGGL(tb<1>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 2) ms = timed([&] { hipLaunchKernelGGL(tb<2>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 3) ms = timed([&] { hipLaunchKernelGGL(tb<3>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 4) ms = timed([&] { hipLaunchKernelGGL(tb<4>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 5) ms = timed([&] { hipLaunchKernelGGL(tb<5>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 6) ms = timed([&] { hipLaunchKernelGGL(tb<6>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 7) ms = timed([&] { hipLaunchKernelGGL(tb<7>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 8) ms = timed([&] { hipLaunchKernelGGL(tb<8>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 9) ms = timed([&] { hipLaunchKernelGGL(tb<9>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 10) ms = timed([&] { hipLaunchKernelGGL(tb<10>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 11) ms = timed([&] { hipLaunchKernelGGL(tb<11>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 12) ms = timed([&] { hipLaunchKernelGGL(tb<12>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 13) ms = timed([&] { hipLaunchKernelGGL(tb<13>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 14) ms = timed([&] { hipLaunchKernelGGL(tb<14>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        if (m == 15) ms = timed([&] { hipLaunchKernelGGL(tb<15>, g, b, 0, 0, tab, cells, lanes, trips, sink); });
        const double cellsTouched = (double)lanes * trips * 4;
        printf("%-28s %u lanes x %u KiB tables, %u trips: %8.2f ms   %6.1f G cells/s\n", names[m], lanes, kib, trips, ms, cellsTouched / ms / 1e6);
    }
    return 0;
}
