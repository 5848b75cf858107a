// membench.hip -- memory-system microbenchmarks behind the round-3 decode design decisions (test infrastructure; results under profiles/).
//   gather16   : every lane loads 16 bytes at a pseudo-random (unaligned) address of a region of R bytes, four loads in flight;
//                R from L2-sized to HBM-sized -> random-access rate of L2 / Infinity Cache / HBM (what K3's far-match gathers can reach)
//   lzmock     : the memory behaviour of a LANE-PER-FRAME sequence executor (64 frames per wave, every frame's own streams):
//                per sequence an 8-byte sequence load, a 16-byte literal load, a 16-byte load from the frame's own earlier output
//                (offset distribution of the bench corpus), 16-byte unaligned "wildcopy" stores at the output cursor.
//                No table logic, no correctness: an upper bound for such a design before it is written.
//   rawcheck   : same-wave store -> load through global memory without a fence (other lane's bytes, unaligned): must always see the store
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct alignas(16) V16 { uint32_t a, b, c, d; };
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__device__ __forceinline__ V16 ld16(const uint8_t* p) { V16 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st16(uint8_t* p, const V16& v) { __builtin_memcpy(p, &v, 16); }

extern "C" __global__ void gather16(const uint8_t* buf, uint64_t mask, uint32_t iters, uint32_t* sink)
{
    uint32_t s = mix(blockIdx.x * 1024u + threadIdx.x + 1u), acc = 0;
    for (uint32_t i = 0; i < iters; i++) {
        uint64_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { s = s * 1664525u + 1013904223u; const uint32_t hi = mix(s); o[k] = ((((uint64_t)hi << 32) | s) >> 7) & mask; }
        V16 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = ld16(buf + o[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) acc += v[k].a ^ v[k].d;
        s += acc & 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// ---- lane-per-frame LZ executor mock. seq word: ll[0:8) | ml[8:16) | off[16:34)
extern "C" __global__ void lzmock_fill(uint64_t* seq, uint32_t frames, uint32_t nseq)
{
    const uint64_t total = (uint64_t)frames * nseq;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t h = mix((uint32_t)i * 2654435761u + 17u), h2 = mix(h + 0x9E3779B9u);
        const uint32_t ll = (h % 100) < 67 ? 0u : 1u + ((h >> 8) % 15);                 // 67 % no literals, else 1..15 (mean 8: 2.7 per sequence)
        uint32_t ml = 5u + ((h >> 16) % 13);                                             // 5..17 (mean 11)
        if ((h2 & 63) == 0) ml += (h2 >> 8) % 48;                                        // a few long ones
        const uint32_t lg = 4 + (h2 >> 16) % 13;                                         // offsets log-uniform 16 .. 128 Ki
        const uint32_t off = (1u << lg) + ((h2 >> 4) & ((1u << lg) - 1));
        seq[i] = (uint64_t)ll | ((uint64_t)ml << 8) | ((uint64_t)off << 16);
    }
}
// direct form: every copy is global -> global, stores at the cursor (16 bytes, unaligned, later stores overwrite the slop)
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void lzmock(const uint64_t* seq, const uint8_t* lit, uint8_t* dst, uint32_t frames, uint32_t nseq, uint32_t* sink)
{
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= frames) return;
    const uint64_t* sq = seq + (uint64_t)f * nseq;
    const uint8_t* lp = lit + (uint64_t)f * 32768;
    uint8_t* out = dst + (uint64_t)f * 131072;
    uint32_t op = 0, l = 0;
    uint64_t q = sq[0];
    for (uint32_t n = 0; n < nseq; n++) {
        const uint64_t qn = sq[n + 1 < nseq ? n + 1 : n];
        const uint32_t ll = (uint32_t)q & 255, ml = (uint32_t)(q >> 8) & 255; uint32_t off = (uint32_t)(q >> 16);
        if (op + ll + ml + 64 > 131072) break;
        const V16 lv = ld16(lp + (l & 32767 - 31));
        const uint32_t mpos = op + ll;
        if (off > mpos) off = mpos ? mpos : 1;
        const uint8_t* ms = mpos ? out + mpos - off : lp;
        const V16 m0 = ld16(ms);
        if (ll) st16(out + op, lv);
        st16(out + mpos, m0);
        if (ml > 16) { const V16 m1 = ld16(ms + 16); st16(out + mpos + 16, m1); if (ml > 32) for (uint32_t j = 32; j < ml; j += 16) st16(out + mpos + j, ld16(ms + j)); }
        op = mpos + ml; l += ll; q = qn;
    }
    if (op == 0xFFFFFFFFu) sink[0] = op;
}
// ring form: the lane's output goes through a private LDS ring (RING bytes, [16-byte word][lane] layout), flushed to global memory in whole
// 64-byte pieces by the lane itself; matches closer than the ring's safe reach are LDS -> LDS, the others global -> LDS
template <int WAVES, uint32_t RING>
__global__ __launch_bounds__(64, WAVES) void lzmock_ring(const uint64_t* seq, const uint8_t* lit, uint8_t* dst, uint32_t frames, uint32_t nseq, uint32_t* sink)
{
    __shared__ uint8_t ring[64 * (RING + 32)];                 // lane-major: a lane's ring is contiguous (+ 32 bytes of slop mirrored by hand)
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= frames) return;
    uint8_t* R = ring + threadIdx.x * (RING + 32);
    const uint64_t* sq = seq + (uint64_t)f * nseq;
    const uint8_t* lp = lit + (uint64_t)f * 32768;
    uint8_t* out = dst + (uint64_t)f * 131072;
    uint32_t op = 0, l = 0, flushed = 0;
    uint64_t q = sq[0];
    for (uint32_t n = 0; n < nseq; n++) {
        const uint64_t qn = sq[n + 1 < nseq ? n + 1 : n];
        const uint32_t ll = (uint32_t)q & 255, ml = (uint32_t)(q >> 8) & 255; uint32_t off = (uint32_t)(q >> 16);
        if (op + ll + ml + 64 > 131072) break;
        const V16 lv = ld16(lp + (l & 32767 - 31));
        const uint32_t mpos = op + ll;
        if (off > mpos) off = mpos ? mpos : 1;
        const bool nearM = off + 48 <= RING - 64 && mpos >= off;                        // the source is still in the ring
        const uint8_t* ms = mpos ? out + mpos - off : lp;
        V16 m0 = ld16(nearM ? lp : ms), m1 = m0;
        if (ll) st16(R + (op & (RING - 1)), lv);
        if (nearM) m0 = ld16(R + ((mpos - off) & (RING - 1)));
        st16(R + (mpos & (RING - 1)), m0);
        if (ml > 16) { m1 = ld16(nearM ? R + ((mpos - off + 16) & (RING - 1)) : ms + 16); st16(R + ((mpos + 16) & (RING - 1)), m1);
                       if (ml > 32) for (uint32_t j = 32; j < ml; j += 16) st16(R + ((mpos + j) & (RING - 1)), ld16(nearM ? R + ((mpos - off + j) & (RING - 1)) : ms + j)); }
        op = mpos + ml; l += ll; q = qn;
        while (flushed + 64 <= op) {                                                     // whole 64-byte pieces leave for global memory
            const uint8_t* s = R + (flushed & (RING - 1));
            const V16 a = ld16(s), b = ld16(s + 16), c = ld16(s + 32), d = ld16(s + 48);
            st16(out + flushed, a); st16(out + flushed + 16, b); st16(out + flushed + 32, c); st16(out + flushed + 48, d);
            flushed += 64;
        }
    }
    if (op == 0xFFFFFFFFu) sink[0] = op;
}

extern "C" __global__ void rawcheck(uint8_t* buf, uint32_t rounds, uint32_t* bad)
{
    const uint32_t lane = threadIdx.x;
    uint8_t* base = buf + (size_t)blockIdx.x * 4096;
    uint32_t errs = 0;
    for (uint32_t r = 1; r <= rounds; r++) {
        V16 v; v.a = r * 64 + lane; v.b = ~v.a; v.c = v.a * 3; v.d = v.a * 5;
        const uint32_t shift = (r * 7) & 1023;
        st16(base + shift + 16 * lane, v);
        // read what lane + 1 wrote, 5 bytes early (straddles two lanes' stores)
        const uint32_t o = (lane + 1) & 63;
        uint8_t got[16]; __builtin_memcpy(got, base + shift + 16 * o - (o ? 5 : 0), 16);
        V16 w; w.a = r * 64 + o; w.b = ~w.a; w.c = w.a * 3; w.d = w.a * 5;
        uint8_t want[16]; __builtin_memcpy(want, &w, 16);
        const uint32_t skip = o ? 5 : 0;
        for (uint32_t k = skip; k < 16; k++) if (got[k] != want[k - skip]) errs++;
    }
    if (errs) atomicAdd(bad, errs);
}

template <class F> static float timed(F launch)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0)); launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs\n", p.name, cus);
    uint32_t* sink; CHECK(hipMalloc(&sink, 64)); CHECK(hipMemset(sink, 0, 64));
    const bool skipGather = argc > 2;
    if (!skipGather) {   // ---- gather16
        const size_t big = (size_t)16 << 30;
        uint8_t* buf; CHECK(hipMalloc(&buf, big + 4096)); CHECK(hipMemset(buf, 1, big + 4096));
        const size_t regions[] = {(size_t)2 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)192 << 20, (size_t)512 << 20, (size_t)2 << 30, (size_t)16 << 30};
        for (size_t R : regions) for (int w : {2, 4, 8}) {
            const uint64_t mask = R - 1; const uint32_t iters = 256;
            const int blocks = cus * w * 4;
            const float ms = timed([&] { hipLaunchKernelGGL(gather16, dim3(blocks), dim3(64), 0, 0, buf, mask, iters, sink); });
            const double loads = (double)blocks * 64 * iters * 4;
            printf("gather16 region %6zu MiB waves/SIMD %d: %.3f ms  %.1f G loads/s  (%.2f TB/s at 64 B per load)\n", R >> 20, w, ms, loads / ms / 1e6, loads * 64 / ms / 1e9);
        }
        CHECK(hipFree(buf));
    }
    if (!skipGather) {   // ---- rawcheck
        uint8_t* buf; CHECK(hipMalloc(&buf, (size_t)cus * 8 * 4096 + 4096)); uint32_t* bad; CHECK(hipMalloc(&bad, 4)); CHECK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(rawcheck, dim3(cus * 8), dim3(64), 0, 0, buf, 20000u, bad); CHECK(hipDeviceSynchronize());
        uint32_t h = 0; CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("rawcheck: %u wrong bytes in %u x 20000 store->load rounds without a fence\n", h, cus * 8);
        CHECK(hipFree(buf)); CHECK(hipFree(bad));
    }
    {   // ---- lzmock
        const uint32_t frames = argc > 1 ? (uint32_t)atoi(argv[1]) : 65536, nseq = 9300;
        uint64_t* seq; uint8_t* lit; uint8_t* dst;
        CHECK(hipMalloc(&seq, (size_t)frames * nseq * 8)); CHECK(hipMalloc(&lit, (size_t)frames * 32768 + 64)); CHECK(hipMalloc(&dst, (size_t)frames * 131072 + 4096));
        CHECK(hipMemset(lit, 7, (size_t)frames * 32768 + 64)); CHECK(hipMemset(dst, 0, (size_t)frames * 131072 + 4096));
        hipLaunchKernelGGL(lzmock_fill, dim3(cus * 16), dim3(256), 0, 0, seq, frames, nseq); CHECK(hipDeviceSynchronize());
        const double bytes = (double)frames * 131072;
#define RUN(name, K) do { const float ms = timed([&] { hipLaunchKernelGGL(K, dim3((frames + 63) / 64), dim3(64), 0, 0, seq, lit, dst, frames, nseq, sink); }); \
            printf("%-28s %u frames x %u sequences: %.3f ms  (%.0f GB/s of output)\n", name, frames, nseq, ms, bytes / ms / 1e6); } while (0)
        RUN("lzmock direct  w1", (lzmock<1>)); RUN("lzmock direct  w2", (lzmock<2>)); RUN("lzmock direct  w4", (lzmock<4>)); RUN("lzmock direct  w8", (lzmock<8>));
        RUN("lzmock ring256 w2", (lzmock_ring<2, 256>)); RUN("lzmock ring512 w1", (lzmock_ring<1, 512>)); RUN("lzmock ring512 w2", (lzmock_ring<2, 512>));
        RUN("lzmock ring1024 w1", (lzmock_ring<1, 1024>));
    }
    return 0;
}
