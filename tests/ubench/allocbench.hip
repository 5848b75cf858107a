// allocbench.hip -- does the way the flat match kernel's 25 GiB of tables were ALLOCATED decide their random-access rate? (test infrastructure;
// result under profiles/). Background: zhip_encode_match_flat_kernel runs in two regimes, ~420 and ~480 ms per 65 536 frames, within one box and
// one build; r04v / r04w tied the slow one to tables that landed in VRAM released earlier in the same process. This
// program runs tablebench's ldst kernel (a dependent chain of trips, four random 4-byte cells loaded and stored per trip, 65 536 lanes x 384 KiB)
// over a buffer obtained in one of several ways, one way per process:
//     fresh          hipMalloc, first allocation of the process
//     prefrag        hipMalloc after allocating, touching and releasing 9 + 24 + 3 + 1 GiB (the decode direction's arenas)
//     contig         hipExtMallocWithFlags(hipDeviceMallocContiguous)
//     vmm <MiB>      a reserved address range backed by separately created physical chunks of <MiB> (hipMemCreate / hipMemMap)
//     prefrag+vmm <MiB>, prefrag+contig
// Usage: allocbench <mode> [chunk MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__global__ __launch_bounds__(64) void tb(uint32_t* tab, uint32_t cells, uint32_t lanes, uint32_t trips, uint32_t* sink)
{
    const uint32_t l = blockIdx.x * 64 + threadIdx.x;
    if (l >= lanes) return;
    uint32_t* t = tab + (size_t)l * cells;
    uint32_t s = mix(l + 1), acc = 0;
    for (uint32_t i = 0; i < trips; i++) {
        uint32_t ix[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { s = s * 1664525u + 1013904223u; ix[k] = mix(s) % cells; }
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = t[ix[k]];
#pragma unroll
        for (int k = 0; k < 4; k++) t[ix[k]] = i + k;
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        s += acc & 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char* mode = argc > 1 ? argv[1] : "fresh";
    const size_t chunkMiB = argc > 2 ? (size_t)atol(argv[2]) : 2;
    const uint32_t lanes = 65536, cells = 384 * 256, trips = 2000;
    const size_t total = (size_t)lanes * cells * 4;
    int dev = 0; CHECK(hipGetDevice(&dev));
    if (strstr(mode, "prefrag")) {
        const size_t gib[4] = {9, 24, 3, 1}; void* p[4];
        for (int i = 0; i < 4; i++) { CHECK(hipMalloc(&p[i], gib[i] << 30)); CHECK(hipMemset(p[i], 1, gib[i] << 30)); }
        CHECK(hipDeviceSynchronize());
        for (int i = 0; i < 4; i++) CHECK(hipFree(p[i]));
    }
    uint32_t* tab = nullptr; const char* how = "hipMalloc";
    if (strstr(mode, "contig")) { CHECK(hipExtMallocWithFlags((void**)&tab, total, hipDeviceMallocContiguous)); how = "hipExtMallocWithFlags(contiguous)"; }
    else if (strstr(mode, "vmm")) {
        hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        size_t gran = 0; CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
        size_t chunk = chunkMiB << 20; if (chunk < gran) chunk = gran;
        const size_t n = (total + chunk - 1) / chunk;
        void* va = nullptr; CHECK(hipMemAddressReserve(&va, n * chunk, chunk < ((size_t)1 << 30) ? chunk : ((size_t)1 << 30), nullptr, 0));
        for (size_t i = 0; i < n; i++) {
            hipMemGenericAllocationHandle_t h; CHECK(hipMemCreate(&h, chunk, &prop, 0));
            CHECK(hipMemMap((char*)va + i * chunk, chunk, 0, h, 0));
            CHECK(hipMemRelease(h));
        }
        hipMemAccessDesc ad; memset(&ad, 0, sizeof ad); ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        CHECK(hipMemSetAccess(va, n * chunk, &ad, 1));
        tab = (uint32_t*)va; how = "hipMemCreate chunks";
        printf("   (granularity %zu KiB, %zu chunks of %zu MiB)\n", gran >> 10, n, chunk >> 20);
    }
    else CHECK(hipMalloc((void**)&tab, total));
    uint32_t* sink; CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemset(tab, 0, total)); CHECK(hipDeviceSynchronize());
    const dim3 g((lanes + 63) / 64), b(64);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(tb, g, b, 0, 0, tab, cells, lanes, trips, sink); CHECK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(tb, g, b, 0, 0, tab, cells, lanes, trips, sink); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; if (ms < best) best = ms;
    }
    printf("%-16s %-36s base %p : ldst %7.2f ms (best of 3, mean %.2f)  %5.1f G cells/s\n", mode, how, (void*)tab, best, sum / 3, (double)lanes * trips * 4 / best / 1e6);
    return 0;
}
