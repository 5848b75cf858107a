// ubench.hip -- issue-rate / latency microbenchmarks for gfx950 that the decode kernels' shapes are chosen from (test
// infrastructure; built by tests/ubench/build.sh, run on the GPU box by the tests/run_r02*.sh scripts, results under profiles/).
//   valu_dep   : one dependent chain of integer VALU ops per wave          -> cycles per instruction, W waves per SIMD
//   valu_ind   : four independent chains per wave
//   lds_chase  : dependent ds_read_b32 chain (conflict-free / random banks) -> cycles per hop
//   lds_u16    : the K2 access: ds_read_u16 at random cells of a per-lane table, three in flight, then a dependent ALU op
//   dpp_chain  : dependent v_add_u32 with quad_perm DPP
//   gather     : 64 lanes x distinct 128-byte lines, L2-resident, four loads in flight
// Every kernel is launched with one workgroup of 256 * W threads per CU (W waves per SIMD), timed per wave with s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP 64
#define ITER 64

__device__ __forceinline__ uint64_t now() { return __builtin_amdgcn_s_memtime(); }

extern "C" __global__ void valu_dep(uint64_t* out, uint32_t seed)
{
    uint32_t x = threadIdx.x + seed, y = seed | 1;
    const uint64_t t0 = now();
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < REP / 4; k++) {
            asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(y));
            asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(x));
            asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(x) : "v"(y));
            asm volatile("v_and_or_b32 %0, %0, %1, 1" : "+v"(x) : "v"(y));
        }
    }
    const uint64_t t1 = now();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (x == 0x12345) out[0] = x;
}

extern "C" __global__ void valu_ind(uint64_t* out, uint32_t seed)
{
    uint32_t a = threadIdx.x + seed, b = a * 3, c = a * 5, d = a * 7, y = seed | 1;
    const uint64_t t0 = now();
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < REP / 4; k++) {
            asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(y));
            asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(b));
            asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(c) : "v"(y));
            asm volatile("v_and_or_b32 %0, %0, %1, 1" : "+v"(d) : "v"(y));
        }
    }
    const uint64_t t1 = now();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if ((a ^ b ^ c ^ d) == 0x12345) out[0] = a;
}

// mixed VALU + SALU stream the way K3 looks: two VALU, one SALU
extern "C" __global__ void mix_vs(uint64_t* out, uint32_t seed)
{
    uint32_t a = threadIdx.x + seed, b = a * 3, y = seed | 1; uint32_t s = seed;
    const uint64_t t0 = now();
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < REP / 4; k++) {
            asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(y));
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
            asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(b));
            asm volatile("s_lshl_b32 %0, %0, 1" : "+s"(s));
        }
    }
    const uint64_t t1 = now();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if ((a ^ b ^ s) == 0x12345) out[0] = a;
}

extern "C" __global__ void dpp_chain(uint64_t* out, uint32_t seed)
{
    uint32_t x = threadIdx.x + seed;
    const uint64_t t0 = now();
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < REP; k++) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
    }
    const uint64_t t1 = now();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (x == 0x12345) out[0] = x;
}

// LDS pointer chase. mode 0: lane l walks its own column (word index = 64 * k + l: conflict-free); mode 1: random words
extern "C" __global__ void lds_chase(uint64_t* out, uint32_t mode)
{
    extern __shared__ uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t W = 2048;                                  // words per wave
    uint32_t* base = lds + wave * W;
    for (uint32_t j = lane; j < W; j += 64) {
        uint32_t nxt;
        if (mode == 0) nxt = ((j + 64) % W);
        else nxt = (j * 1103515245u + 12345u + lane * 7u) % W;
        base[j] = (uint32_t)((wave * W + nxt) * 4);
    }
    __syncthreads();
    uint32_t a = (wave * W + lane) * 4;
    const uint64_t t0 = now();
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < REP; k++) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) :: "memory");
    }
    const uint64_t t1 = now();
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
    if (a == 0x12345) out[0] = a;
}

// K2's access: three ds_read_u16 at random cells of a per-lane table region, waited for together, then dependent ALU feeding
// the next addresses (chain). lanes = active lanes per wave (others masked off by the caller through `lanes`).
extern "C" __global__ void lds_u16(uint64_t* out, uint32_t lanes)
{
    extern __shared__ uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const uint32_t total = nw * lanes;                       // frames in the workgroup
    const uint32_t stride = 2564;                             // bytes per frame
    for (uint32_t j = threadIdx.x; j < total * stride / 4; j += blockDim.x) lds[j] = j * 2654435761u;
    __syncthreads();
    uint64_t dt = 0;
    if (lane < lanes) {
        const uint32_t fb = (wave * lanes + lane) * stride;
        uint32_t s0 = lane * 2, s1 = lane * 6, s2 = lane * 10;
        const uint64_t t0 = now();
        for (int i = 0; i < ITER * 8; i++) {
            uint32_t c0, c1, c2;
            uint32_t a0 = fb + ((s0 & 511) << 1), a1 = fb + 1024 + ((s1 & 511) << 1), a2 = fb + 2048 + ((s2 & 255) << 1);
            asm volatile("ds_read_u16 %0, %3\n\tds_read_u16 %1, %4\n\tds_read_u16 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(c0), "=&v"(c1), "=&v"(c2) : "v"(a0), "v"(a1), "v"(a2) : "memory");
            s0 = c0 + s2; s1 = c1 + s0; s2 = c2 + s1;
        }
        const uint64_t t1 = now();
        dt = t1 - t0;
        if ((s0 ^ s1 ^ s2) == 0x12345) out[0] = s0;
    }
    if (lane == 0) out[blockIdx.x * nw + wave] = dt;
}

extern "C" __global__ void gather(uint64_t* out, const uint32_t* buf, uint32_t mask)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t idx = (threadIdx.x * 977u + blockIdx.x * 131071u) & mask;
    uint32_t acc = 0;
    const uint64_t t0 = now();
    for (int i = 0; i < ITER * 4; i++) {
        const uint32_t i0 = idx, i1 = (idx + 0x1357u * 32) & mask, i2 = (idx + 0x2468u * 32) & mask, i3 = (idx + 0x3579u * 32) & mask;
        const uint32_t v0 = buf[i0 * 32], v1 = buf[i1 * 32], v2 = buf[i2 * 32], v3 = buf[i3 * 32];   // 128-byte lines
        acc += v0 + v1 + v2 + v3;
        idx = (idx * 5u + 1u + (acc & 1)) & mask;
    }
    const uint64_t t1 = now();
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x12345) out[0] = acc;
}

static double run(const char* name, const void* fn, int wavesPerSimd, int nblocks, size_t ldsBytes, uint64_t* d_out, void** args, double opsPerWave)
{
    const int threads = 256 * wavesPerSimd;
    const int nw = nblocks * threads / 64;
    CHECK(hipMemset(d_out, 0, nw * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipLaunchKernel(fn, dim3(nblocks), dim3(threads), args, ldsBytes, 0));   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    CHECK(hipLaunchKernel(fn, dim3(nblocks), dim3(threads), args, ldsBytes, 0));
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h(nw);
    CHECK(hipMemcpy(h.data(), d_out, nw * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[nw / 2], mx = (double)h[nw - 1];
    printf("%-10s waves/SIMD %d  cycles/op per wave: median %.2f max %.2f | per SIMD (median / waves): %.2f | kernel %.3f ms\n",
           name, wavesPerSimd, med / opsPerWave, mx / opsPerWave, med / opsPerWave / wavesPerSimd, ms);
    return med / opsPerWave;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int nb = p.multiProcessorCount;
    uint64_t* d_out; CHECK(hipMalloc(&d_out, 1 << 20));
    uint32_t seed = 12345;
    void* a1[] = {&d_out, &seed};
    const int ws[] = {1, 2, 3, 4};      // 256 * W threads per workgroup: 1024 is the limit
    for (int w : ws) run("valu_dep", (const void*)valu_dep, w, nb, 0, d_out, a1, (double)REP * ITER);
    for (int w : ws) run("valu_ind", (const void*)valu_ind, w, nb, 0, d_out, a1, (double)REP * ITER);
    run("dpp_chain", (const void*)dpp_chain, 1, nb, 0, d_out, a1, (double)REP * ITER);
    CHECK(hipFuncSetAttribute((const void*)lds_chase, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void*)lds_u16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (uint32_t mode = 0; mode < 2; mode++) {
        void* a2[] = {&d_out, &mode};
        for (int w : {1, 2, 4}) { printf("mode %u ", mode); run("lds_chase", (const void*)lds_chase, w, nb, (size_t)w * 4 * 2048 * 4, d_out, a2, (double)REP * ITER); }
    }
    // K2 shapes: (waves per CU, lanes per wave)
    const int shapes[][2] = {{1, 60}, {4, 15}, {8, 7}};
    for (auto& s : shapes) {
        uint32_t lanes = (uint32_t)s[1]; void* a3[] = {&d_out, &lanes};
        const int threads = 64 * s[0];
        const int nw = nb * s[0];
        CHECK(hipMemset(d_out, 0, nw * 8));
        CHECK(hipLaunchKernel((const void*)lds_u16, dim3(nb), dim3(threads), a3, 160 * 1024 - 1024, 0));
        CHECK(hipDeviceSynchronize());
        std::vector<uint64_t> h(nw);
        CHECK(hipMemcpy(h.data(), d_out, nw * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("lds_u16    %d waves x %d lanes: cycles per step (3 reads + 3 dependent adds) median %.1f max %.1f\n", s[0], s[1],
               (double)h[nw / 2] / (ITER * 8), (double)h[nw - 1] / (ITER * 8));
    }
    {
        uint32_t* buf; const uint32_t lines = 1u << 15;                         // 4 MiB: L2-resident per XCD
        CHECK(hipMalloc(&buf, (size_t)lines * 128)); CHECK(hipMemset(buf, 1, (size_t)lines * 128));
        uint32_t mask = lines - 1; void* a4[] = {&d_out, &buf, &mask};
        for (int w : {1, 2, 4}) run("gather", (const void*)gather, w, nb, 0, d_out, a4, (double)ITER * 4 * 4);
        const uint32_t big = 1u << 23; uint32_t* buf2;                          // 1 GiB: HBM
        CHECK(hipMalloc(&buf2, (size_t)big * 128)); CHECK(hipMemset(buf2, 1, (size_t)big * 128));
        uint32_t mask2 = big - 1; void* a5[] = {&d_out, &buf2, &mask2};
        for (int w : {1, 2, 4}) { printf("HBM "); run("gather", (const void*)gather, w, nb, 0, d_out, a5, (double)ITER * 4 * 4); }
    }
    return 0;
}
