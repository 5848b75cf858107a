#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path on MI355X.

Workload (BASELINE.json configs[1]): multi_decompress_to_buffer over 65 536 pre-compressed level-3 frames of
128 KiB "Silesia-like" slices per GPU, inputs and outputs resident in HBM. A step = one pass over all frames.
value = uncompressed GB/s (1e9) over all ranks; roofline = (compressed + uncompressed bytes) / kernel time vs HBM peak.
cpu_baseline = the reference libzstd 1.5.7 (oracle/_ref) decoding a bounded sample of the same frames on host cores.

  python bench.py [--gpus N --steps K --warmup W --frames F]
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FRAME = 131072
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _ref_lib():
    """libzstd 1.5.7 as the input generator and CPU baseline: the reference build, else the copy inside the image."""
    from tests import reflib
    if reflib.have_ref():
        return reflib.RefZstd(), "reference"
    import glob
    cands = glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libzstd-*.so.1.5.7")
    if cands:
        reflib.REF_SO = cands[0]
        return reflib.RefZstd(), "reference"
    raise RuntimeError("no libzstd 1.5.7 available to prepare the bench input")


def _threads():
    return max(1, min(os.cpu_count() or 1, 64))


def compress_on_host(ref, raw_np, nthreads):
    """level-3 frames of every row of raw_np [F, FRAME] with the reference library, contiguous partition by bytes."""
    F = raw_np.shape[0]
    bound = ref.lib.ZSTD_compressBound(FRAME)
    outs = [None] * F
    base = raw_np.ctypes.data

    def work(lo, hi):
        buf = C.create_string_buffer(bound)
        for i in range(lo, hi):
            n = ref.compress_into(C.addressof(buf), bound, base + i * FRAME, FRAME)
            outs[i] = buf.raw[:n]

    step = (F + nthreads - 1) // nthreads
    ts = [threading.Thread(target=work, args=(lo, min(F, lo + step))) for lo in range(0, F, step)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    return outs


def _mtbench():
    """oracle/libzo_mtbench.so: native pthread driver for the CPU baseline (Python threads would mostly measure the interpreter lock)"""
    path = os.path.join(ROOT, "oracle", "libzo_mtbench.so")
    if not os.path.exists(path):                                    # normally built by __graft_entry__.build(); gcc is in the image
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libzo_mtbench.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(path)
    lib.zo_mt_bench.restype = C.c_double
    lib.zo_mt_bench.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_int, C.c_int, C.c_int]
    return lib


def cpu_decompress_baseline(ref, frames, nthreads, passes=12):
    """reference ZSTD_decompressStream on a bounded sample: native threads, one DCtx each, static contiguous partition
    (decompress_worker's loop, c-ext/decompressor.c:1237-1320). Returns (GB/s of uncompressed bytes, passes)."""
    from tests import reflib
    F = len(frames)
    blob = np.frombuffer(b"".join(frames), dtype=np.uint8)
    offs = np.zeros(F + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(f) for f in frames])
    best = _mtbench().zo_mt_bench(reflib.REF_SO.encode(), 1, blob.ctypes.data, offs.ctypes.data, F, FRAME, 3, nthreads, passes)
    assert best > 0, "native CPU baseline failed (%r)" % best
    return F * FRAME / best / 1e9, passes


def cpu_compress_baseline(ref, raw_np, nthreads, passes=6):
    """reference ZSTD_compressStream2(e_end) at level 3 on a bounded sample: native threads, one CCtx each (compress_worker's loop,
    c-ext/compressor.c:1127-1216). Returns (GB/s of uncompressed bytes, passes)."""
    from tests import reflib
    F = raw_np.shape[0]
    raw_np = np.ascontiguousarray(raw_np)
    offs = (np.arange(F + 1, dtype=np.uint64) * np.uint64(FRAME))
    best = _mtbench().zo_mt_bench(reflib.REF_SO.encode(), 0, raw_np.ctypes.data, offs.ctypes.data, F, 0, 3, nthreads, passes)
    assert best > 0, "native CPU baseline failed (%r)" % best
    return F * FRAME / best / 1e9, passes


def measure_compress(ctx, raw, frames, rank, world, dev, F, steps, warmup):
    """multi_compress_to_buffer direction on the first F rows of raw: every frame must be bit-identical to libzstd's.
    Returns (elapsed seconds over `steps` passes (max over ranks), compressed total, per-kernel times)."""
    bound = FRAME + (FRAME >> 8)
    bound = (bound + 15) & ~15
    src_segs = torch.zeros((F, 2), dtype=torch.int64, device=dev)
    src_segs[:, 0] = torch.arange(F, device=dev, dtype=torch.int64) * FRAME
    src_segs[:, 1] = FRAME
    dst_segs = torch.zeros((F, 2), dtype=torch.int64, device=dev)
    dst_segs[:, 0] = torch.arange(F, device=dev, dtype=torch.int64) * bound
    dst_segs[:, 1] = bound
    dst = torch.zeros(F * bound, dtype=torch.uint8, device=dev)
    out_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
    status = torch.zeros(F, dtype=torch.int32, device=dev)
    src = raw[:F].reshape(-1)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
    barrier()
    for k in (1, 5, 6, 8):
        ctx.kernel_time(k)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status)
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = {k: ctx.kernel_time(k) for k in (1, 5, 6, 8)}
    assert int(status.abs().max().item()) == 0, "a frame failed to compress"
    sizes = out_sizes.cpu().numpy()
    out = dst.view(F, bound).cpu().numpy()
    for i in range(F):                                              # bit-exactness gate over every frame
        assert out[i, : sizes[i]].tobytes() == frames[i], "frame %d differs from libzstd 1.5.7" % i
    ctotal = int(sizes.sum())
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    del dst
    return elapsed, ctotal, ktimes


def bench_compress(args, ctx, raw, frames, ref, ref_kind, nthreads, rank, world, dev, F):
    """--direction compress: the whole line is about multi_compress_to_buffer on the same inputs."""
    elapsed, ctotal, ktimes = measure_compress(ctx, raw, frames, rank, world, dev, F, args.steps, args.warmup)
    kdom = max(ktimes, key=lambda k: ktimes[k][0] * ktimes[k][1])
    kernel_ms, launches = ktimes[kdom]
    value = world * F * FRAME * args.steps / elapsed / 1e9
    line = {
        "metric": "GB/s uncompressed throughput, batch compress of 128 KiB inputs at level 3 (bit-exact vs libzstd 1.5.7)",
        "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "multi_compress_to_buffer (device-resident): %d x 128 KiB Silesia-like inputs per GPU, level 3" % F,
                   "frames_per_gpu": F, "frame_bytes": FRAME, "level": 3, "compression_ratio": round(F * FRAME / ctotal, 3),
                   "parallelism": "frames sharded by rank, no data-path collective"},
    }
    if rank == 0:
        launches_per_step = max(1, int(launches) // max(1, args.steps))
        algo_bytes = (F * FRAME + ctotal) // launches_per_step
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        line["kernels"] = {ctx.kernel_name(k): {"avg_ms": round(v[0], 4), "launches": int(v[1])} for k, v in ktimes.items() if v[1]}
        traffic = None
        try:                                                         # PMC passes are separate runs (profiles/README.md); scaled per launch
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            per_frame = tj["bytes_per_frame"].get(ctx.kernel_name(kdom))
            if per_frame:
                traffic = int(per_frame * F / launches_per_step)
        except (OSError, ValueError, KeyError):
            pass
        line["roofline"] = {"bound": "hbm", "kernel": ctx.kernel_name(kdom), "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                            "kernel_ms": round(kernel_ms, 3), "launches": int(launches), "algorithmic_bytes_per_launch": int(algo_bytes)}
        if world == 1 and not args.no_cpu_baseline:
            sample = min(F, 4096)
            v, reps = cpu_compress_baseline(ref, raw[:sample].cpu().numpy(), nthreads)
            line["cpu_baseline"] = {"value": round(v, 3), "unit": "GB/s", "cores": nthreads, "kind": ref_kind,
                                    "sample": "libzstd 1.5.7 ZSTD_compressStream2(e_end) level 3 over the first %d inputs of the same "
                                              "workload, %d native threads (host has %d cores), best of %d passes"
                                              % (sample, nthreads, os.cpu_count() or 0, reps)}
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=65536, help="frames per GPU (BASELINE config: 65536)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compress-frames", type=int, default=65536,
                    help="after the timed decompress steps, also time multi_compress_to_buffer on this many of the same inputs (0 = skip)")
    ap.add_argument("--direction", choices=["decompress", "compress"], default="decompress",
                    help="decompress is the BASELINE.json headline; compress times multi_compress_to_buffer on the same inputs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import zstandard_amd as zstd
    from zstandard_amd.device import DeviceBatchContext
    from tests.corpus import Corpus

    F = args.frames
    t0 = time.time()
    corpus = Corpus(device=dev)
    raw = corpus.frames(rank * F, F, chunk=256)                      # [F, FRAME] uint8 in HBM (this rank's shard)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    ref, ref_kind = _ref_lib()
    nthreads = _threads()
    t0 = time.time()
    raw_np = raw.cpu().numpy()
    frames = compress_on_host(ref, raw_np, nthreads)
    t_comp = time.time() - t0
    csizes = np.array([len(f) for f in frames], dtype=np.int64)
    ctotal = int(csizes.sum())
    src_segs_np = np.zeros((F, 2), dtype=np.int64)
    src_segs_np[:, 1] = csizes
    src_segs_np[1:, 0] = np.cumsum(csizes)[:-1]
    dst_segs_np = np.zeros((F, 2), dtype=np.int64)
    dst_segs_np[:, 0] = np.arange(F, dtype=np.int64) * FRAME
    dst_segs_np[:, 1] = FRAME
    src = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).to(dev)
    src_segs = torch.from_numpy(src_segs_np).to(dev)
    dst_segs = torch.from_numpy(dst_segs_np).to(dev)
    dst = torch.zeros(F * FRAME, dtype=torch.uint8, device=dev)
    out_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
    status = torch.zeros(F, dtype=torch.int32, device=dev)

    ctx = DeviceBatchContext()
    if args.direction == "compress":
        return bench_compress(args, ctx, raw, frames, ref, ref_kind, nthreads, rank, world, dev, F)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status)
    barrier()
    for k in (0, 2, 7, 3, 4):
        ctx.kernel_time(k)                                           # reset the per-kernel timers
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status)
    barrier()
    elapsed = time.perf_counter() - t0
    # dominant decode kernel = the one with the largest total time over the timed steps (HIP events on the launch stream)
    ktimes = {k: ctx.kernel_time(k) for k in (0, 2, 7, 3, 4)}
    kdom = max(ktimes, key=lambda k: ktimes[k][0] * ktimes[k][1])
    kernel_ms, launches = ktimes[kdom]

    # correctness gate at full size: every frame decoded, every byte equals the original input
    assert int(status.abs().max().item()) == 0, "a frame failed to decode"
    assert bool((out_sizes == FRAME).all().item())
    assert torch.equal(dst.view(F, FRAME), raw), "round-trip mismatch"

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the only cross-rank exchange the path needs: the segment table of the sharded result (payload stays per GPU)
        gathered = [torch.zeros_like(out_sizes) for _ in range(world)]
        dist.all_gather(gathered, out_sizes)

    total_unc = world * F * FRAME
    value = total_unc * args.steps / elapsed / 1e9
    line = {
        "metric": "GB/s uncompressed throughput, batch decompress of 128 KiB level-3 frames",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "multi_decompress_to_buffer (device-resident): %d x 128 KiB Silesia-like frames per GPU, "
                               "level 3, frames compressed by libzstd 1.5.7" % F,
                   "frames_per_gpu": F, "frame_bytes": FRAME, "level": 3,
                   "compression_ratio": round(F * FRAME / ctotal, 3), "parallelism": "frames sharded by rank, no data-path collective"},
    }
    if rank == 0:
        # algorithmic bytes of one launch: compressed bytes read + uncompressed bytes written for the frames that launch covers
        launches_per_step = max(1, int(launches) // max(1, args.steps))
        algo_bytes = (F * FRAME + ctotal) // launches_per_step
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        line["kernels"] = {ctx.kernel_name(k): {"avg_ms": round(v[0], 4), "launches": int(v[1])} for k, v in ktimes.items() if v[1]}
        traffic = None
        try:                                                         # PMC passes are separate runs (profiles/README.md); scaled per launch
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            per_frame = tj["bytes_per_frame"].get(ctx.kernel_name(kdom))
            if per_frame:
                traffic = int(per_frame * F / launches_per_step)
        except (OSError, ValueError, KeyError):
            pass
        line["roofline"] = {"bound": "hbm", "kernel": ctx.kernel_name(kdom), "achieved": round(achieved, 2),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                            "traffic": traffic, "kernel_ms": round(kernel_ms, 4), "launches": int(launches),
                            "algorithmic_bytes_per_launch": int(algo_bytes)}
        if world == 1 and not args.no_cpu_baseline:
            sample = min(F, 4096)
            v, reps = cpu_decompress_baseline(ref, frames[:sample], nthreads)
            line["cpu_baseline"] = {"value": round(v, 3), "unit": "GB/s", "cores": nthreads, "kind": ref_kind,
                                    "sample": "libzstd 1.5.7 ZSTD_decompressStream over the first %d frames of the same "
                                              "workload, %d native threads (host has %d cores), best of %d passes"
                                              % (sample, nthreads, os.cpu_count() or 0, reps)}
        line["setup_s"] = {"generate": round(t_gen, 1), "host_compress": round(t_comp, 1)}
    if args.compress_frames > 0:
        # the other half of BASELINE.json's metric, on a bounded slice of the same inputs (the timed decompress region above is over)
        Fc = min(F, args.compress_frames)
        # the compressor is its own object in the reference API: release the decode direction's working set (its context's arenas,
        # the frames and their output) and start from a fresh context, as a ZstdCompressor next to a ZstdDecompressor would
        del dst, src, src_segs, dst_segs, out_sizes, status
        ctx.close()
        torch.cuda.empty_cache()
        ctx = DeviceBatchContext()
        c_elapsed, c_total, c_k = measure_compress(ctx, raw, frames, rank, world, dev, Fc, 3, 2)
        if rank == 0:
            line["compress"] = {"value": round(world * Fc * FRAME * 3 / c_elapsed / 1e9, 3), "unit": "GB/s", "frames_per_gpu": Fc, "steps": 3,
                                "ms_per_step": round(c_elapsed / 3 * 1e3, 3), "bit_exact_vs_libzstd": True,
                                "kernels": {ctx.kernel_name(k): {"avg_ms": round(v[0], 4), "launches": int(v[1])} for k, v in c_k.items() if v[1]}}
            if world == 1 and not args.no_cpu_baseline:
                sample = min(Fc, 4096)
                v, reps = cpu_compress_baseline(ref, raw[:sample].cpu().numpy(), nthreads)
                line["compress"]["cpu_baseline"] = {"value": round(v, 3), "unit": "GB/s", "cores": nthreads, "kind": ref_kind,
                                                    "sample": "libzstd 1.5.7 level 3, first %d inputs, %d threads, best of %d" % (sample, nthreads, reps)}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
