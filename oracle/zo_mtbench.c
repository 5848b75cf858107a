/* oracle/zo_mtbench.c -- native multi-threaded CPU baseline driver (TEST / BENCH INFRASTRUCTURE ONLY, never shipped).
 *
 * bench.py's cpu_baseline leg times the reference libzstd on the host cores. Driving it from Python threads measures the
 * interpreter lock as much as libzstd, so this file does what the reference's batch workers do, natively: a static contiguous
 * partition of the frames over pthreads (c-ext/compressor.c:1127-1216, c-ext/decompressor.c:1237-1320), one context per thread,
 * ZSTD_CCtx_setPledgedSrcSize + ZSTD_compressStream2(ZSTD_e_end) per item (compressor.c:1035-1043) or ZSTD_decompressStream per
 * frame (decompressor.c:1150). The libzstd to time is passed by path (oracle/_ref/libzstd_ref.so, i.e. the reference's own
 * zstd.c) and bound with dlopen, so nothing of the reference is compiled into this file.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { void* p; size_t size; size_t pos; } zbuf;
typedef struct {
    void* (*createCCtx)(void); size_t (*freeCCtx)(void*);
    size_t (*setParam)(void*, int, int); size_t (*setPledged)(void*, unsigned long long);
    size_t (*compressStream2)(void*, zbuf*, zbuf*, int); size_t (*compressBound)(size_t);
    void* (*createDCtx)(void); size_t (*freeDCtx)(void*);
    size_t (*decompressStream)(void*, zbuf*, zbuf*); unsigned (*isError)(size_t);
    /* shared dictionaries, as the reference's workers use them: one ZSTD_CDict / ZSTD_DDict referenced by every worker context
     * (c-ext/compressor.c:1155-1163 ZSTD_CCtx_refCDict, c-ext/decompressor.c:1262-1270 ZSTD_DCtx_refDDict) */
    void* (*createCDict)(const void*, size_t, int); size_t (*freeCDict)(void*); size_t (*refCDict)(void*, const void*);
    void* (*createDDict)(const void*, size_t); size_t (*freeDDict)(void*); size_t (*refDDict)(void*, const void*);
} zapi;

typedef struct {
    const zapi* z; int decompress, level; const uint8_t* src; const uint64_t* offs; uint32_t lo, hi; size_t maxOut;
    pthread_barrier_t* start; int failed;
    const void* cdict; const void* ddict;
    struct timespec t0, t1;          /* this worker's first and last instant of work */
} job;

static void* worker(void* arg)
{
    job* j = (job*)arg;
    const zapi* z = j->z;
    uint8_t* out = (uint8_t*)malloc(j->maxOut ? j->maxOut : 1);
    void* ctx = j->decompress ? z->createDCtx() : z->createCCtx();
    if (!j->decompress) { z->setParam(ctx, 100, j->level); z->setParam(ctx, 200, 1); z->setParam(ctx, 201, 0); z->setParam(ctx, 202, 1); }
    if (j->cdict && !j->decompress) z->refCDict(ctx, j->cdict);
    if (j->ddict && j->decompress) z->refDDict(ctx, j->ddict);
    pthread_barrier_wait(j->start);
    clock_gettime(CLOCK_MONOTONIC, &j->t0);
    for (uint32_t i = j->lo; i < j->hi; i++) {
        zbuf o = { out, j->maxOut, 0 };
        zbuf in = { (void*)(j->src + j->offs[i]), (size_t)(j->offs[i + 1] - j->offs[i]), 0 };
        size_t r;
        if (j->decompress) r = z->decompressStream(ctx, &o, &in);
        else { z->setPledged(ctx, in.size); r = z->compressStream2(ctx, &o, &in, 2); }
        if (r != 0) j->failed = 1;
    }
    clock_gettime(CLOCK_MONOTONIC, &j->t1);
    if (j->decompress) z->freeDCtx(ctx); else z->freeCCtx(ctx);
    free(out);
    return 0;
}

/* Times `passes` passes over frames [0, n) (offs has n + 1 entries into src) with `threads` threads; returns the best pass in
 * seconds, or a negative value on failure. maxOut = capacity of each thread's output buffer (uncompressed frame size for
 * decompress, ZSTD_compressBound for compress when 0 is passed). */
double zo_mt_bench_dict(const char* libpath, int decompress, const uint8_t* src, const uint64_t* offs, uint32_t n, size_t maxOut,
                        int level, int threads, int passes, const void* dict, size_t dictSize, double* times);

double zo_mt_bench(const char* libpath, int decompress, const uint8_t* src, const uint64_t* offs, uint32_t n, size_t maxOut,
                   int level, int threads, int passes)
{
    return zo_mt_bench_dict(libpath, decompress, src, offs, n, maxOut, level, threads, passes, 0, 0, 0);
}

/* the same with an optional shared dictionary; times[passes] (if not NULL) receives every pass's duration so that the caller can
 * quote a median next to the best */
double zo_mt_bench_dict(const char* libpath, int decompress, const uint8_t* src, const uint64_t* offs, uint32_t n, size_t maxOut,
                        int level, int threads, int passes, const void* dict, size_t dictSize, double* times)
{
    void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1.0;
    zapi z;
    *(void**)&z.createCCtx = dlsym(h, "ZSTD_createCCtx"); *(void**)&z.freeCCtx = dlsym(h, "ZSTD_freeCCtx");
    *(void**)&z.setParam = dlsym(h, "ZSTD_CCtx_setParameter"); *(void**)&z.setPledged = dlsym(h, "ZSTD_CCtx_setPledgedSrcSize");
    *(void**)&z.compressStream2 = dlsym(h, "ZSTD_compressStream2"); *(void**)&z.compressBound = dlsym(h, "ZSTD_compressBound");
    *(void**)&z.createDCtx = dlsym(h, "ZSTD_createDCtx"); *(void**)&z.freeDCtx = dlsym(h, "ZSTD_freeDCtx");
    *(void**)&z.decompressStream = dlsym(h, "ZSTD_decompressStream"); *(void**)&z.isError = dlsym(h, "ZSTD_isError");
    *(void**)&z.createCDict = dlsym(h, "ZSTD_createCDict"); *(void**)&z.freeCDict = dlsym(h, "ZSTD_freeCDict"); *(void**)&z.refCDict = dlsym(h, "ZSTD_CCtx_refCDict");
    *(void**)&z.createDDict = dlsym(h, "ZSTD_createDDict"); *(void**)&z.freeDDict = dlsym(h, "ZSTD_freeDDict"); *(void**)&z.refDDict = dlsym(h, "ZSTD_DCtx_refDDict");
    if (!z.createCCtx || !z.compressStream2 || !z.createDCtx || !z.decompressStream || !z.compressBound) return -2.0;
    void* cdict = 0; void* ddict = 0;
    if (dict && dictSize) {
        if (!z.createCDict || !z.refCDict || !z.createDDict || !z.refDDict) return -2.0;
        if (decompress) ddict = z.createDDict(dict, dictSize); else cdict = z.createCDict(dict, dictSize, level);
        if (!cdict && !ddict) return -5.0;
    }
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > n) threads = (int)n;
    if (!decompress && maxOut == 0) {
        size_t big = 0;
        for (uint32_t i = 0; i < n; i++) { size_t s = (size_t)(offs[i + 1] - offs[i]); if (s > big) big = s; }
        maxOut = z.compressBound(big);
    }
    double best = -3.0;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    job* jobs = (job*)malloc(sizeof(job) * (size_t)threads);
    for (int p = 0; p < passes; p++) {
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, 0, (unsigned)threads + 1);
        const uint32_t per = (n + (uint32_t)threads - 1) / (uint32_t)threads;
        for (int t = 0; t < threads; t++) {
            job* j = &jobs[t];
            j->z = &z; j->decompress = decompress; j->level = level; j->src = src; j->offs = offs; j->maxOut = maxOut; j->start = &bar; j->failed = 0; j->cdict = cdict; j->ddict = ddict;
            j->lo = (uint32_t)t * per; j->hi = j->lo + per > n ? n : j->lo + per; if (j->lo > n) j->lo = j->hi = n;
            pthread_create(&th[t], 0, worker, j);
        }
        pthread_barrier_wait(&bar);                      /* contexts exist, buffers allocated: the work starts now */
        int failed = 0;
        for (int t = 0; t < threads; t++) { pthread_join(th[t], 0); failed |= jobs[t].failed; }
        pthread_barrier_destroy(&bar);
        if (failed) { best = -4.0; break; }
        /* the pass lasts from the first worker's first instant of work to the last worker's last one */
        double first = 0, last = 0;
        for (int t = 0; t < threads; t++) {
            const double s0 = (double)jobs[t].t0.tv_sec + 1e-9 * (double)jobs[t].t0.tv_nsec, s1 = (double)jobs[t].t1.tv_sec + 1e-9 * (double)jobs[t].t1.tv_nsec;
            if (t == 0 || s0 < first) first = s0;
            if (t == 0 || s1 > last) last = s1;
        }
        const double dt = last - first;
        if (times) times[p] = dt;
        if (best < 0 || dt < best) best = dt;
    }
    free(th); free(jobs);
    if (cdict) z.freeCDict(cdict);
    if (ddict) z.freeDDict(ddict);
    dlclose(h);
    return best;
}

/* ---- input preparation for bench.py: every item [offs[i], offs[i+1]) of src compressed by `threads` native threads into slot i of
 * out (slotCap bytes each), sizes to outSizes. Same per-item calls as the timed worker above. Returns 0, or a negative value. */
typedef struct { const zapi* z; int level; const uint8_t* src; const uint64_t* offs; uint32_t lo, hi; uint8_t* out; size_t slotCap; uint64_t* outSizes;
                 const void* cdict; int failed; } prepjob;
static void* prep_worker(void* arg)
{
    prepjob* j = (prepjob*)arg;
    const zapi* z = j->z;
    void* ctx = z->createCCtx();
    z->setParam(ctx, 100, j->level); z->setParam(ctx, 200, 1); z->setParam(ctx, 201, 0); z->setParam(ctx, 202, 1);
    if (j->cdict) z->refCDict(ctx, j->cdict);
    for (uint32_t i = j->lo; i < j->hi; i++) {
        zbuf o = { j->out + (size_t)i * j->slotCap, j->slotCap, 0 };
        zbuf in = { (void*)(j->src + j->offs[i]), (size_t)(j->offs[i + 1] - j->offs[i]), 0 };
        z->setPledged(ctx, in.size);
        if (z->compressStream2(ctx, &o, &in, 2) != 0) j->failed = 1;
        j->outSizes[i] = o.pos;
    }
    z->freeCCtx(ctx);
    return 0;
}
int zo_mt_compress_all(const char* libpath, const uint8_t* src, const uint64_t* offs, uint32_t n, int level, int threads,
                       const void* dict, size_t dictSize, uint8_t* out, size_t slotCap, uint64_t* outSizes)
{
    void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    zapi z; memset(&z, 0, sizeof z);
    *(void**)&z.createCCtx = dlsym(h, "ZSTD_createCCtx"); *(void**)&z.freeCCtx = dlsym(h, "ZSTD_freeCCtx");
    *(void**)&z.setParam = dlsym(h, "ZSTD_CCtx_setParameter"); *(void**)&z.setPledged = dlsym(h, "ZSTD_CCtx_setPledgedSrcSize");
    *(void**)&z.compressStream2 = dlsym(h, "ZSTD_compressStream2");
    *(void**)&z.createCDict = dlsym(h, "ZSTD_createCDict"); *(void**)&z.freeCDict = dlsym(h, "ZSTD_freeCDict"); *(void**)&z.refCDict = dlsym(h, "ZSTD_CCtx_refCDict");
    if (!z.createCCtx || !z.compressStream2 || !z.setParam || !z.setPledged) return -2;
    void* cdict = 0;
    if (dict && dictSize) { if (!z.createCDict || !z.refCDict) return -2; cdict = z.createCDict(dict, dictSize, level); if (!cdict) return -5; }
    if (threads < 1) threads = 1;
    if ((uint32_t)threads > n) threads = (int)(n ? n : 1);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    prepjob* jobs = (prepjob*)malloc(sizeof(prepjob) * (size_t)threads);
    const uint32_t per = (n + (uint32_t)threads - 1) / (uint32_t)threads;
    for (int t = 0; t < threads; t++) {
        prepjob* j = &jobs[t];
        j->z = &z; j->level = level; j->src = src; j->offs = offs; j->out = out; j->slotCap = slotCap; j->outSizes = outSizes; j->cdict = cdict; j->failed = 0;
        j->lo = (uint32_t)t * per; j->hi = j->lo + per > n ? n : j->lo + per; if (j->lo > n) j->lo = j->hi = n;
        pthread_create(&th[t], 0, prep_worker, j);
    }
    int failed = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], 0); failed |= jobs[t].failed; }
    free(th); free(jobs);
    if (cdict) z.freeCDict(cdict);
    dlclose(h);
    return failed ? -4 : 0;
}
