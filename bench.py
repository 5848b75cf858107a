#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path on MI355X.

Workloads (BASELINE.json configs):
  --config decompress (default; configs[1]): multi_decompress_to_buffer over 65 536 pre-compressed level-3 frames of 128 KiB
      "Silesia-like" slices per GPU, inputs and outputs resident in HBM. A step = one pass over all frames. value = uncompressed
      GB/s (1e9) over all ranks of the DECOMPRESS direction; the other half of BASELINE.json's metric string (compress, configs[2])
      is timed after the timed region on the same inputs and reported in the line's "compress" object.
  --config compress (configs[2]): the same inputs through multi_compress_to_buffer as its own line, every frame compared with libzstd's.
  --config dict (configs[3]): 262 144 x 4 KiB JSON-like documents with a shared trained dictionary (tests/golden/dict_json4k.bin,
      made by train_dictionary's own call in tests/golden/make_dict_json4k.py): value = compress GB/s, every frame compared with
      libzstd's; the line's "decompress" object is the opposite direction on those frames.
  --config roundtrip (configs[4]): 1 048 576 / 8 = 131 072 x 128 KiB mixed-entropy buffers per GPU generated in HBM, compressed and
      decompressed on the GPU (a step = both directions), the round trip compared in HBM, a sample of the frames compared with libzstd's;
      with N > 1 ranks the compressed payload is then all-gathered over RCCL (sharded.allgatherv_payload), timed on its own.
The 128 KiB corpus is the "silesia" class mix of tests/corpus.py (level-3 ratio ~3.1); --mix default is round 1's mix (2.55).
roofline = (compressed + uncompressed bytes) / time vs the HBM peak, for the dominant kernel (HIP events on its launch stream) and
end to end. cpu_baseline = the reference libzstd 1.5.7 (oracle/_ref) on host threads over a bounded sample of the same workload.

  python bench.py [--gpus N --steps K --warmup W --frames F --config ...]
  N > 1: either `python bench.py --gpus N ...` (it starts its own N ranks through torch.distributed.run on 127.0.0.1) or the driver's
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`;
  --dry-launch stops after the process group is formed and the partition agreed (tests/test_distributed_gloo.py runs it at N = 2 on CPU).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FRAME = 131072
DOC = 4096
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _ref_lib():
    """libzstd 1.5.7 as the input generator and CPU baseline: the reference build (oracle/_ref), else the copy inside the image
    (tests/reflib.have_ref picks; never the restatement)."""
    from tests import reflib
    if reflib.have_ref():
        return reflib.RefZstd(), "reference"
    raise RuntimeError("no libzstd 1.5.7 available to prepare the bench input")


def _mtbench():
    """oracle/libzo_mtbench.so: native pthread driver for input preparation and the CPU baseline (Python threads would mostly
    measure the interpreter lock)"""
    path = os.path.join(ROOT, "oracle", "libzo_mtbench.so")
    if not os.path.exists(path):                                    # normally built by __graft_entry__.build(); gcc is in the image
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libzo_mtbench.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(path)
    lib.zo_mt_bench_dict.restype = C.c_double
    lib.zo_mt_bench_dict.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                     C.c_char_p, C.c_size_t, C.c_void_p]
    lib.zo_mt_compress_all.restype = C.c_int
    lib.zo_mt_compress_all.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                       C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


HOST_THREADS = min(os.cpu_count() or 1, 64)      # input preparation on the host; main() divides the host's cores among the ranks of a node


def compress_on_host(raw_np, item, dict_data=None, level=3):
    """level-3 frames of every row of raw_np [F, item] with the reference library (native threads, contiguous partition).
    Returns (list of frames, int64 sizes)."""
    from tests import reflib
    F = raw_np.shape[0]
    raw_np = np.ascontiguousarray(raw_np)
    offs = np.arange(F + 1, dtype=np.uint64) * np.uint64(item)
    slot = item + (item >> 7) + 512
    out = np.empty((F, slot), dtype=np.uint8)
    sizes = np.zeros(F, dtype=np.uint64)
    rc = _mtbench().zo_mt_compress_all(reflib.REF_SO.encode(), raw_np.ctypes.data, offs.ctypes.data, F, level, HOST_THREADS,
                                       dict_data, len(dict_data) if dict_data else 0, out.ctypes.data, slot, sizes.ctypes.data)
    assert rc == 0, "reference compression of the bench input failed (%d)" % rc
    sizes = sizes.astype(np.int64)
    return [out[i, : sizes[i]].tobytes() for i in range(F)], sizes


def cpu_baseline(decompress, blob, offs, n, max_out, unc_bytes, dict_data=None, passes=5):
    """the reference libzstd on host threads over items [0, n) of blob (offs: n + 1 offsets): native threads, one context each, static
    contiguous partition (compress_worker / decompress_worker, c-ext/compressor.c:1127-1216, c-ext/decompressor.c:1237-1320), at 64
    threads and at every host core; per thread count the MEDIAN and best of `passes` passes. Returns the cpu_baseline object."""
    from tests import reflib
    lib = _mtbench()
    ncpu = os.cpu_count() or 1
    counts = sorted({min(64, ncpu), ncpu})
    by = {}
    for t in counts:
        times = (C.c_double * passes)()
        best = lib.zo_mt_bench_dict(reflib.REF_SO.encode(), 1 if decompress else 0, blob.ctypes.data, offs.ctypes.data, n, max_out, 3, t, passes,
                                    dict_data, len(dict_data) if dict_data else 0, times)
        assert best > 0, "native CPU baseline failed (%r)" % best
        ts = sorted(times)
        by[t] = {"median": round(unc_bytes / ts[len(ts) // 2] / 1e9, 3), "best": round(unc_bytes / ts[0] / 1e9, 3)}
    top = max(by, key=lambda t: by[t]["median"])
    # the per-core figure (the reference's own bench reports threads=1 too): one thread over a bounded prefix of the same items
    n1 = max(1, min(n, 512 if decompress else 192))
    t1 = (C.c_double * 3)()
    b1 = lib.zo_mt_bench_dict(reflib.REF_SO.encode(), 1 if decompress else 0, blob.ctypes.data, offs.ctypes.data, n1, max_out, 3, 1, 3,
                              dict_data, len(dict_data) if dict_data else 0, t1)
    unc1 = unc_bytes * n1 / n
    threads1 = {"value": round(unc1 / sorted(t1)[1] / 1e9, 4), "unit": "GB/s", "items": n1} if b1 > 0 else None
    return {"value": by[top]["median"], "unit": "GB/s", "cores": top, "by_threads": {str(t): v for t, v in by.items()}, "passes": passes,
            "host_cores": ncpu, "threads1": threads1, "cpu_model": cpu_model()}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


USE_DIST = False      # set by main(): world > 1 (or forced, see there)


def emit(line):
    """rank 0's one JSON line. A run with ZHIP_BENCH_NO_VERIFY (diagnostic kernel variants that produce wrong bytes on purpose) skipped the
    full-size correctness gates: its line says so and carries no claim of exactness."""
    verified = not os.environ.get("ZHIP_BENCH_NO_VERIFY")
    line["verified"] = verified
    if not verified:
        line["metric"] = "UNVERIFIED DIAGNOSTIC RUN (ZHIP_BENCH_NO_VERIFY): " + line.get("metric", "")
        for k in ("round_trip_exact", "bit_exact_vs_libzstd"):
            for obj in [line] + [v for v in line.values() if isinstance(v, dict)]:
                if k in obj:
                    obj[k] = None
    global PENDING_LINE
    PENDING_LINE = json.dumps(ordered_for_the_driver(line))   # printed by main() as the process's LAST output (after the process group is gone: RCCL writes a version banner to stdout)


def _dig(obj, *path):
    for p in path:
        if not isinstance(obj, dict) or p not in obj:
            return None
        obj = obj[p]
    return obj


def ordered_for_the_driver(line):
    """The driver keeps the TAIL of stdout (~2 KB) next to the keys it parses, and the line is ~30 KB: the bulky
    sub-objects go first, the contract's scalars after them, and a flat `summary` of every figure BASELINE.json's metric is made of --
    both directions, the combined value, the host-buffer API, configs[3] / configs[4] -- goes LAST, so that the record shows them
    whatever is cut off in front. Same keys and values as before plus `summary`; only the order changes."""
    bulky = ("kernels", "setup_s", "dict", "roundtrip", "blocks", "compress", "host_api", "combined")
    out = {k: line[k] for k in bulky if k in line}
    for k, v in line.items():
        if k not in out and k not in ("roofline", "cpu_baseline", "verified"):
            out[k] = v
    for k in ("roofline", "cpu_baseline"):
        if k in line:
            out[k] = line[k]
    s = {"decompress_gbs": line.get("value"), "decompress_ms": line.get("ms_per_step"), "decompress_hbm_frac": _dig(line, "roofline", "frac"),
         "compress_gbs": _dig(line, "compress", "value"), "compress_ms": _dig(line, "compress", "ms_per_step"),
         "compress_hbm_frac": _dig(line, "compress", "roofline", "frac"),
         "decompress_line_transfer_frac": _dig(line, "roofline", "memory_requests", "frac_of_ceiling"),
         "compress_line_transfer_frac": _dig(line, "compress", "roofline", "memory_requests", "frac_of_ceiling"),
         "compress_match_kernel_ms": _dig(line, "compress", "regime", "match_kernel_ms_per_65536_frames"),
         "compress_regime": _dig(line, "compress", "regime", "class"), "combined_gbs": _dig(line, "combined", "value"),
         "host_api_8192_c": _dig(line, "host_api", "frames_8192", "compress"), "host_api_8192_d": _dig(line, "host_api", "frames_8192", "decompress"),
         "host_api_65536_c": _dig(line, "host_api", "frames_65536", "compress"), "host_api_65536_d": _dig(line, "host_api", "frames_65536", "decompress"),
         "host_api_devices": _dig(line, "host_api", "devices"),
         "host_api_all_devices": _dig(line, "host_api", "all_devices", "devices"),
         "host_api_all_devices_c": _dig(line, "host_api", "all_devices", "frames_%d" % (_dig(line, "config", "frames_per_gpu") or 65536), "compress"),
         "host_api_all_devices_d": _dig(line, "host_api", "all_devices", "frames_%d" % (_dig(line, "config", "frames_per_gpu") or 65536), "decompress"),
         "dict_c_gbs": _dig(line, "dict", "value"), "dict_d_gbs": _dig(line, "dict", "decompress", "value"),
         "roundtrip_gbs": _dig(line, "roundtrip", "value"), "roundtrip_c_gbs": _dig(line, "roundtrip", "compress", "value"),
         "roundtrip_d_gbs": _dig(line, "roundtrip", "decompress", "value"),
         "blocks_d_gbs": _dig(line, "blocks", "value"), "blocks_c_gbs": _dig(line, "blocks", "compress", "value"),
         "cpu_decompress_gbs": _dig(line, "cpu_baseline", "value"), "cpu_compress_gbs": _dig(line, "compress", "cpu_baseline", "value"),
         "n_gpus": line.get("n_gpus")}
    if line.get("metric", "").startswith("GB/s uncompressed throughput, batch compress"):       # --config compress: `value` IS the compress figure
        s = {"compress_gbs": line.get("value"), "compress_ms": line.get("ms_per_step"), "compress_hbm_frac": _dig(line, "roofline", "frac"),
             "compress_match_kernel_ms": _dig(line, "regime", "match_kernel_ms_per_65536_frames"), "compress_regime": _dig(line, "regime", "class"),
             "n_gpus": line.get("n_gpus")}
    out["summary"] = {k: v for k, v in s.items() if v is not None}
    out["verified"] = line.get("verified")
    return out


PENDING_LINE = None


class Job:
    """one direction's device buffers + the timed loop (barrier + synchronize on both sides, max over ranks)"""

    def __init__(self, world, dev):
        self.world, self.dev = world, dev

    def barrier(self):
        if USE_DIST:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, ctx, kernels, steps, warmup):
        for _ in range(warmup):
            fn()
        self.barrier()
        for k in kernels:
            if ctx.kernel_name(k):
                ctx.kernel_time(k)                                   # reset the per-kernel timers
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.barrier()
        elapsed = time.perf_counter() - t0
        ktimes = {k: ctx.kernel_time(k) for k in kernels if ctx.kernel_name(k)}
        if USE_DIST:
            import torch.distributed as dist
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, ktimes


def segs(offsets, lengths, dev):
    s = np.zeros((len(lengths), 2), dtype=np.int64)
    s[:, 0] = offsets
    s[:, 1] = lengths
    return torch.from_numpy(s).to(dev)


def roofline(ctx, ktimes, steps, algo_bytes_per_step, ms_per_step, frames_per_step=None):
    """The headline figures are the PIPELINE's: achieved = the step's algorithmic bytes / the sum of the average launch
    durations of the direction's kernels in one step (HIP events on the launch stream; the kernels of a step run one after the other),
    traffic = FETCH_SIZE + WRITE_SIZE summed over those kernels. `dominant_kernel` keeps the prescribed single-kernel formula (the kernel
    with the largest total time: the step's algorithmic bytes per launch / its average launch duration), `end_to_end` the same bytes over
    the step's wall time. Traffic comes from the separate rocprofv3 --pmc passes of the same build (tests/run_profiles.sh ->
    tests/prof_traffic.py -> profiles/traffic.json, bytes per 128 KiB frame), scaled to the step."""
    # (round 6: K1b runs BESIDE K2 on a side stream, so the decode direction's kernel durations no longer add up to the step; the library times a chunk's pipeline from
    # K1's start to K3's end as a pseudo-kernel, SPAN_NAME, and that span -- plus the generic kernel's launch behind it -- is the pipeline's time)
    span = {k: v for k, v in ktimes.items() if ctx.kernel_name(k) == SPAN_NAME and v[1]}
    ktimes = {k: v for k, v in ktimes.items() if ctx.kernel_name(k) != SPAN_NAME}
    kdom = max(ktimes, key=lambda k: ktimes[k][0] * ktimes[k][1])
    kernel_ms, launches = ktimes[kdom]
    launches_per_step = max(1, int(launches) // max(1, steps))
    algo_bytes = algo_bytes_per_step // launches_per_step
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    per_step = lambda ms, n: ms * max(1, int(n) // max(1, steps))
    pipe_ms = sum(per_step(ms, n) for ms, n in ktimes.values() if n)          # kernel time of ONE step, every launch of every kernel
    if span:
        pipe_ms = sum(per_step(ms, n) for ms, n in span.values()) + sum(per_step(ms, n) for k, (ms, n) in ktimes.items() if n and ctx.kernel_name(k) == "zhip_decode_frames_kernel")
    pipe = algo_bytes_per_step / (pipe_ms * 1e-3) / 1e9 if pipe_ms > 0 else 0.0
    traffic, ktraffic, tsrc = None, None, None
    if frames_per_step:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            per = tj["bytes_per_frame"]
            names = [ctx.kernel_name(k) for k, v in ktimes.items() if v[1]]
            if all(nm in per for nm in names if "frames_kernel" not in nm):
                # (K0, the lane-per-frame parser pass, runs inside K1's timer -- it has no timer of its own -- so its traffic rides with K1's)
                extra = per.get("zhip_decode_pre_kernel", 0) if "zhip_decode_lit_kernel" in names else 0
                traffic = int((sum(per.get(nm, 0) for nm in names) + extra) * frames_per_step)
                ktraffic = int(per[ctx.kernel_name(kdom)] * frames_per_step / launches_per_step) if ctx.kernel_name(kdom) in per else None
                tsrc = tj.get("round")
        except (OSError, ValueError, KeyError):
            pass
    e2e = algo_bytes_per_step / (ms_per_step * 1e-3) / 1e9
    # what the dominant kernel asks of the memory system in REQUESTS (the L2's memory-side counters TCC_EA0_RDREQ / WRREQ, one rocprofv3 --pmc pass, profiles/traffic.json):
    # the two kernels that bound this codec sit on a transfer-rate wall the bandwidth fraction does not show -- random 64-byte line transfers, fills and the evictions of
    # partly written lines alike, top out at ~50 G/s (tests/ubench/membench.hip, tablebench.hip); coalesced whole-line writes are not counted against it
    req = None
    if frames_per_step:
        try:
            rq = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("requests_per_frame", {}).get(ctx.kernel_name(kdom))
            if rq and kernel_ms > 0:
                n = frames_per_step / launches_per_step
                rd, wr = rq["read"] * n / (kernel_ms * 1e-3), rq["write"] * n / (kernel_ms * 1e-3)
                w64 = rq.get("write64_share") or 0.0
                req = {"kernel": ctx.kernel_name(kdom), "read_requests_per_s": round(rd / 1e9, 2), "write_requests_per_s": round(wr / 1e9, 2), "unit": "G/s",
                       "write_requests_of_64_bytes": rq.get("write64_share"), "random_line_transfers_per_s": round((rd + wr * (1.0 - w64)) / 1e9, 2), "ceiling": 50.0,
                       "frac_of_ceiling": round((rd + wr * (1.0 - w64)) / 50e9, 3),
                       "measured_in_run": False, "note": "requests per frame from profiles/traffic.json (separate --pmc pass) over this run's kernel time; the ceiling is the microbenchmarks' "
                                                          "rate of random 64-byte reads (49-54 G/s), which read fills and partial-line write-backs share"}
        except (OSError, ValueError, KeyError):
            pass
    return {"bound": "hbm", "kernel": "pipeline: " + " + ".join(ctx.kernel_name(k) for k, v in ktimes.items() if v[1] and v[0] >= 0.05),
            "achieved": round(pipe, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pipe / HBM_PEAK_GBS, 5),
            # traffic: read from profiles/traffic.json (separate rocprofv3 --pmc passes), scaled to the step
            "traffic": traffic, "traffic_source": tsrc, "traffic_measured_in_run": False,
            "kernel_ms_per_step": round(pipe_ms, 4), "algorithmic_bytes_per_step": int(algo_bytes_per_step), "memory_requests": req,
            "dominant_kernel": {"kernel": ctx.kernel_name(kdom), "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": ktraffic,
                                "kernel_ms": round(kernel_ms, 4), "launches": int(launches), "algorithmic_bytes_per_launch": int(algo_bytes),
                                "note": "the step's whole algorithmic bytes over ONE kernel's duration (the prescribed formula): it credits this kernel with bytes the other kernels move"},
            "end_to_end": {"achieved": round(e2e, 2), "frac": round(e2e / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": int(algo_bytes_per_step)}}, kdom


def frame_sections(frames, limit=8192):
    """what each pipeline stage owns of a frame (RFC 8878 3.1.1): per-frame MEANS over the first `limit` frames of
    hdr (frame / block / section headers + table descriptions), huf (compressed literal streams), lit (regenerated literals),
    rawlit (literals stored raw / RLE), seqs (sequence bitstream), nbseq, out. Plain header arithmetic on the host; used to give every
    kernel's roofline ITS OWN algorithmic bytes instead of crediting one kernel with the whole frame."""
    tot = dict(hdr=0, huf=0, lit=0, rawlit=0, seqs=0, nbseq=0, csize=0)
    n = min(len(frames), limit)
    for f in frames[:n]:
        tot["csize"] += len(f)
        fhd = f[4]
        single, dcode, fcode = (fhd >> 5) & 1, fhd & 3, fhd >> 6
        pos = 5 + (0 if single else 1) + (4 if dcode == 3 else dcode) + (single if fcode == 0 else 1 << fcode)
        while True:
            bh = f[pos] | (f[pos + 1] << 8) | (f[pos + 2] << 16); pos += 3
            last, btype, bs = bh & 1, (bh >> 1) & 3, bh >> 3
            if btype == 2:
                b0 = f[pos]; lt, sf = b0 & 3, (b0 >> 2) & 3
                if lt < 2:
                    hl = 1 if sf in (0, 2) else 2 if sf == 1 else 3
                    regen = (b0 >> 3) if hl == 1 else ((b0 >> 4) + (f[pos + 1] << 4) + ((f[pos + 2] << 12) if hl == 3 else 0))
                    body = regen if lt == 0 else 1
                    tot["rawlit"] += regen
                else:
                    hl = 3 if sf < 2 else 4 if sf == 2 else 5
                    h = int.from_bytes(f[pos:pos + hl], "little")
                    bits = 10 if hl == 3 else 14 if hl == 4 else 18
                    regen, body = (h >> 4) & ((1 << bits) - 1), (h >> (4 + bits)) & ((1 << bits) - 1)
                    tot["huf"] += body; tot["lit"] += regen
                sp = pos + hl + body
                nb = f[sp]
                if nb == 255: nb, sp = f[sp + 1] + (f[sp + 2] << 8) + 0x7F00, sp + 3
                elif nb > 127: nb, sp = ((nb - 128) << 8) + f[sp + 1], sp + 2
                else: sp += 1
                tot["nbseq"] += nb
                tot["seqs"] += pos + bs - sp
            pos += bs if btype != 1 else 1
            if last:
                break
    out = {k: v / n for k, v in tot.items()}
    out["hdr"] = out["csize"] - out["huf"] - out["seqs"]
    out["sample_frames"] = n
    return out


def per_kernel_roofline(ctx, ktimes, steps, own_bytes_per_step):
    """every kernel against the HBM peak with ITS OWN algorithmic bytes per launch (own_bytes_per_step: kernel name -> bytes per step)"""
    out = {}
    for k, (ms, launches) in ktimes.items():
        name = ctx.kernel_name(k)
        if not launches or name not in own_bytes_per_step or ms < 0.05:        # (a kernel that found an empty work list)
            continue
        per_launch = own_bytes_per_step[name] / max(1, int(launches) // max(1, steps))
        gbs = per_launch / (ms * 1e-3) / 1e9
        out[name] = {"algorithmic_bytes_per_launch": int(per_launch), "avg_ms": round(ms, 4), "achieved": round(gbs, 2), "frac": round(gbs / HBM_PEAK_GBS, 5)}
    return out


def decode_own_bytes(sec, F, item):
    """algorithmic bytes of the decode pipeline's kernels for F frames: K1 parses headers and table descriptions and
    copies raw literals; K1b reads the Huffman streams and writes the literals; K2 reads the sequence bitstream and writes 8-byte
    sequences; K3 reads sequences, literals and match sources and writes the output"""
    return {"zhip_decode_lit_kernel": F * (sec["hdr"] + sec["rawlit"]),
            "zhip_decode_huf_kernel": F * (sec["huf"] + sec["lit"]),
            "zhip_decode_seq_kernel": F * (sec["seqs"] + 8 * sec["nbseq"]),
            "zhip_decode_exec_kernel": F * (8 * sec["nbseq"] + 2 * item),
            "zhip_decode_exec_dict_kernel": F * (8 * sec["nbseq"] + 2 * item)}


def encode_own_bytes(sec, F, item):
    """the match kernels read the source and write 8-byte sequences; the entropy kernel reads sequences + source (literals) and writes the frame"""
    m = F * (item + 8 * sec["nbseq"])
    return {"zhip_encode_match_flat_kernel": m, "zhip_encode_match_kernel": m, "zhip_encode_match_lds_kernel": m,
            "zhip_encode_entropy_kernel": F * (8 * sec["nbseq"] + sec["lit"] + sec["rawlit"] + sec["csize"])}


def e1f_regime(ctx, ktimes, frames):
    """The flat match kernel's time for the same launch varies by up to 20 % with where the driver placed the context's hash tables (DESIGN.md 4.2:
    397 ... 488 ms per 65 536 frames, a property of one allocation, reproducible within a process). The line says which placement THIS process got:
    the kernel's average launch time scaled to 65 536 frames, against the best and worst the same build has measured (profiles/)."""
    for k, (ms, launches) in ktimes.items():
        if ctx.kernel_name(k) == "zhip_encode_match_flat_kernel" and launches:
            fpl = frames if frames <= 65536 else 131072 if frames >= 131072 else frames          # what one launch holds (zhip_compress_batch_device)
            per = ms * 65536.0 / fpl
            pick = ctx.table_pick()
            return {"match_kernel_ms_per_65536_frames": round(per, 2), "frames_per_launch": int(fpl), "known_range_ms": [397, 488],
                    "table_pick": {"candidates_ms": [round(x, 1) for x in pick[0] if x > 0], "kept": pick[1],
                                   "candidates_are": "probe launches over the first 8 KiB of every source (they rank the allocations like whole launches: profiles/r06zzr_pick_study.txt)"} if pick[0][0] > 0 else None,
                    "class": "fast" if per <= 425 else "slow" if per >= 455 else "middle",
                    "note": "placement of the context's tables, not the build: compare compress figures of equal class (launches of 131 072 frames cost ~9 % less per frame)"}
    return None


def kernels_obj(ctx, ktimes):
    out = {ctx.kernel_name(k): {"avg_ms": round(v[0], 4), "launches": int(v[1])} for k, v in ktimes.items() if v[1]}
    if SPAN_NAME in out and "zhip_decode_huf_kernel" in out:
        out["zhip_decode_huf_kernel"]["note"] = "runs on a side stream beside zhip_decode_seq_kernel, its first part inside that kernel's tail: timed from zhip_decode_seq_kernel's END to its own end, i.e. what it adds to the step (a kernel trace shows its whole residence)"
        out[SPAN_NAME]["note"] = "not a kernel: one chunk's pipeline from K1's start to K3's end (HIP events), what the overlapping kernels cost together"
    return out


DEC_KERNELS = (0, 2, 7, 3, 4, 9)      # generic, K1, K1b, K2, K3, the pipeline's span (K1b overlaps K2: see roofline())
SPAN_NAME = "zhip_decode_pipeline_span"
ENC_KERNELS = (1, 5, 6, 8)


def run_decompress(job, ctx, frames, csizes, raw, item, steps, warmup):
    """the decode direction on `frames` (list of bytes) whose originals are raw [F, item] in HBM. Returns (elapsed, ktimes)."""
    F = len(frames)
    dev = job.dev
    offs = np.zeros(F, dtype=np.int64)
    offs[1:] = np.cumsum(csizes)[:-1]
    src = torch.from_numpy(np.frombuffer(b"".join(frames), dtype=np.uint8).copy()).to(dev)
    src_segs = segs(offs, csizes, dev)
    dst_segs = segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64), dev)
    dst = torch.zeros(F * item, dtype=torch.uint8, device=dev)
    out_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
    status = torch.zeros(F, dtype=torch.int32, device=dev)
    elapsed, ktimes = job.timed(lambda: ctx.decompress(src, src_segs, dst, dst_segs, out_sizes, status), ctx, DEC_KERNELS, steps, warmup)
    if os.environ.get("ZHIP_BENCH_NO_VERIFY"):                     # diagnostic kernel variants that produce wrong bytes on purpose: the line says so (main())
        return elapsed, ktimes, out_sizes
    # correctness gate at full size: every frame decoded, every byte equals the original input
    assert int(status.abs().max().item()) == 0, "a frame failed to decode"
    assert bool((out_sizes == item).all().item())
    assert torch.equal(dst.view(F, item), raw), "round-trip mismatch"
    return elapsed, ktimes, out_sizes


def run_compress(job, ctx, raw, frames, item, steps, warmup):
    """the encode direction on raw [F, item]: every frame must be byte-identical to libzstd's (`frames`). Returns (elapsed, total, ktimes)."""
    F = raw.shape[0]
    dev = job.dev
    bound = (item + (item >> 8) + 64 + 15) & ~15
    src_segs = segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64), dev)
    dst_segs = segs(np.arange(F, dtype=np.int64) * bound, np.full(F, bound, dtype=np.int64), dev)
    dst = torch.zeros(F * bound, dtype=torch.uint8, device=dev)
    out_sizes = torch.zeros(F, dtype=torch.int64, device=dev)
    status = torch.zeros(F, dtype=torch.int32, device=dev)
    src = raw.reshape(-1)
    elapsed, ktimes = job.timed(lambda: ctx.compress(src, src_segs, dst, dst_segs, out_sizes, status), ctx, ENC_KERNELS, steps, warmup)
    assert int(status.abs().max().item()) == 0, "a frame failed to compress"
    sizes = out_sizes.cpu().numpy()
    out = dst.view(F, bound).cpu().numpy()
    for i in range(F):                                              # bit-exactness gate over every frame
        assert out[i, : sizes[i]].tobytes() == frames[i], "frame %d differs from libzstd 1.5.7" % i
    del dst
    return elapsed, int(sizes.sum()), ktimes


def sample_blob(frames, n):
    blob = np.frombuffer(b"".join(frames[:n]), dtype=np.uint8)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(f) for f in frames[:n]])
    return blob, offs


def bench_dict(args, rank, world, dev, steps=None, warmup=None, quiet=False):
    """BASELINE.json configs[3]: multi_compress_to_buffer with a shared trained dictionary (train_dictionary(112640, 10 000 JSON samples),
    the reference's default size, c-ext/compressiondict.c:56-61) over 262 144 x 4 KiB JSON-like documents. Returns the line (rank 0)."""
    steps = steps or args.steps; warmup = args.warmup if warmup is None else warmup
    from zstandard_amd.device import DeviceBatchContext
    from tests.corpus import Corpus
    F = args.docs
    dict_data = open(os.path.join(ROOT, "tests", "golden", "dict_json4k.bin"), "rb").read()
    raw = Corpus(frame_size=DOC, device=dev).json_docs(rank * F, F)
    torch.cuda.synchronize()
    ref, ref_kind = _ref_lib()
    raw_np = raw.cpu().numpy()
    frames, csizes = compress_on_host(raw_np, DOC, dict_data)
    ctotal = int(csizes.sum())
    job = Job(world, dev)
    ctx = DeviceBatchContext(dict_data=dict_data, level=3)
    ctx.set_size_hint(DOC)                                         # what the host API tells the library by itself: the batch's largest source (slots of 48 KiB instead of the attach cutoff's 192: one launch)
    elapsed, ctot2, ktimes = run_compress(job, ctx, raw, frames, DOC, steps, warmup)
    assert ctot2 == ctotal
    ms = elapsed / steps * 1e3
    sec = frame_sections(frames)
    line = {
        "metric": "GB/s uncompressed throughput, batch compress of 4 KiB inputs with a shared trained dictionary at level 3 (bit-exact vs libzstd 1.5.7)",
        "value": round(world * F * DOC * steps / elapsed / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "multi_compress_to_buffer (device-resident) with a shared ZstdCompressionDict (%d bytes, train_dictionary on 10 000 JSON samples): "
                               "%d x 4 KiB JSON-like documents per GPU, level 3" % (len(dict_data), F),
                   "docs_per_gpu": F, "doc_bytes": DOC, "level": 3, "dict_bytes": len(dict_data), "compression_ratio": round(F * DOC / ctotal, 3),
                   "parallelism": "documents sharded by rank, no data-path collective"},
    }
    d_elapsed, d_k, _ = run_decompress(job, ctx, frames, csizes, raw, DOC, steps, warmup)
    if rank == 0:
        line["kernels"] = kernels_obj(ctx, ktimes)
        pick = ctx.table_pick()
        line["table_pick"] = {"candidates_ms": [round(x, 2) for x in pick[0] if x > 0], "kept": pick[1]} if pick[0][0] > 0 else None
        line["roofline"], _ = roofline(ctx, ktimes, steps, F * DOC + ctotal, ms)
        line["roofline"]["per_kernel"] = per_kernel_roofline(ctx, ktimes, steps, encode_own_bytes(sec, F, DOC))
        d_ms = d_elapsed / steps * 1e3
        line["decompress"] = {"value": round(world * F * DOC * steps / d_elapsed / 1e9, 3), "unit": "GB/s", "ms_per_step": round(d_ms, 3),
                              "round_trip_exact": True, "kernels": kernels_obj(ctx, d_k)}
        line["decompress"]["roofline"], _ = roofline(ctx, d_k, steps, F * DOC + ctotal, d_ms)
        line["decompress"]["roofline"]["per_kernel"] = per_kernel_roofline(ctx, d_k, steps, decode_own_bytes(sec, F, DOC))
        if world == 1 and not args.no_cpu_baseline:
            n = min(F, 65536)
            offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(DOC)
            cb = cpu_baseline(False, np.ascontiguousarray(raw_np[:n]), offs, n, 0, n * DOC, dict_data, passes=3 if quiet else 5)
            cb.update({"kind": ref_kind, "sample": "libzstd 1.5.7 ZSTD_compressStream2(e_end) level 3 with the shared ZSTD_CDict over the first %d documents "
                                                   "of the same workload; median of %d passes at the better of 64 / all host threads" % (n, cb["passes"])})
            line["cpu_baseline"] = cb
            blob, boffs = sample_blob(frames, n)
            db = cpu_baseline(True, blob, boffs, n, DOC, n * DOC, dict_data, passes=3 if quiet else 5)
            db.update({"kind": ref_kind, "sample": "ZSTD_decompressStream with the shared ZSTD_DDict, first %d frames" % n})
            line["decompress"]["cpu_baseline"] = db
        if not quiet:
            emit(line)
    ctx.close()
    return line


def bench_roundtrip(args, rank, world, dev, steps=None, warmup=None, quiet=False):
    """BASELINE.json configs[4]: every rank generates its shard of the 1 048 576 x 128 KiB corpus in HBM, compresses it, decompresses
    the frames again and compares in HBM; a sample of frames is compared with libzstd's; N > 1: the compressed payload all-gathered."""
    from zstandard_amd.device import DeviceBatchContext
    from zstandard_amd import sharded
    from tests.corpus import Corpus
    steps = steps or args.steps; warmup = args.warmup if warmup is None else warmup
    F = args.frames if args.frames != 65536 else 131072            # BASELINE config: 1 048 576 / 8 per GPU
    raw = Corpus(device=dev, mix=args.mix).frames(rank * F, F, chunk=256)
    torch.cuda.synchronize()
    job = Job(world, dev)
    bound = (FRAME + (FRAME >> 8) + 64 + 15) & ~15
    src_segs = segs(np.arange(F, dtype=np.int64) * FRAME, np.full(F, FRAME, dtype=np.int64), dev)
    slot_segs = segs(np.arange(F, dtype=np.int64) * bound, np.full(F, bound, dtype=np.int64), dev)
    slots = torch.zeros(F * bound, dtype=torch.uint8, device=dev)
    csz = torch.zeros(F, dtype=torch.int64, device=dev)
    st = torch.zeros(F, dtype=torch.int32, device=dev)
    back = torch.zeros(F * FRAME, dtype=torch.uint8, device=dev)
    bsz = torch.zeros(F, dtype=torch.int64, device=dev)
    st2 = torch.zeros(F, dtype=torch.int32, device=dev)
    cctx, dctx = DeviceBatchContext(), DeviceBatchContext()
    src = raw.reshape(-1)
    c_el, c_k = job.timed(lambda: cctx.compress(src, src_segs, slots, slot_segs, csz, st), cctx, ENC_KERNELS, steps, warmup)
    assert int(st.abs().max().item()) == 0, "a frame failed to compress"
    frame_segs = torch.stack([slot_segs[:, 0], csz], dim=1).contiguous()                  # the frames where they lie, inside their slots
    d_el, d_k = job.timed(lambda: dctx.decompress(slots, frame_segs, back, src_segs, bsz, st2), dctx, DEC_KERNELS, steps, warmup)
    assert int(st2.abs().max().item()) == 0 and bool((bsz == FRAME).all().item()), "a frame failed to decode"
    assert torch.equal(back.view(F, FRAME), raw), "round-trip mismatch"
    ctotal = int(csz.sum().item())
    ns = min(F, 2048)                                              # sampled verification against libzstd (every 1 / (F / ns)-th frame)
    idx = np.linspace(0, F - 1, ns).astype(np.int64)
    sample_raw = raw[torch.from_numpy(idx).to(dev)].cpu().numpy()
    want, _ = compress_on_host(sample_raw, FRAME)
    got_sz = csz.cpu().numpy()
    slots_v = slots.view(F, bound)
    for j, i in enumerate(idx):
        assert bytes(slots_v[int(i), : int(got_sz[i])].cpu().numpy()) == want[j], "frame %d differs from libzstd 1.5.7" % i
    gather_ms = None
    if USE_DIST:                                                   # reassemble the compressed output on every rank (north_star's all-gatherv)
        import torch.distributed as dist
        dense, dsegs = sharded._compact(slots, slot_segs, csz, st, dev)
        sizes = sharded._exchange_sizes(csz, None)
        bounds = [(r * F, (r + 1) * F) for r in range(world)]
        res = sharded.ShardResult(rank, bounds, dense, dsegs, sizes, st)
        job.barrier()
        t0 = time.perf_counter()
        sharded.allgatherv_payload(res)
        job.barrier()
        gather_ms = (time.perf_counter() - t0) * 1e3
        assert int(res.full_arena.numel()) >= int(sizes.sum()) and res.slice_stride > 0
    step_s = (c_el + d_el) / steps
    sec = frame_sections(want)
    line = {"metric": "GB/s uncompressed throughput, compress + decompress round trip of 128 KiB buffers at level 3 (each step: both directions)",
            "value": round(world * F * FRAME / step_s / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "round trip on the GPU: %d x 128 KiB mixed-entropy buffers per GPU generated in HBM, multi_compress_to_buffer then "
                                   "multi_decompress_to_buffer (device-resident), level 3" % F, "frames_per_gpu": F, "frame_bytes": FRAME, "level": 3,
                       "compression_ratio": round(F * FRAME / ctotal, 3), "corpus_mix": args.mix,
                       "verification": "round trip compared in HBM for every frame; %d evenly spaced frames compared byte for byte with libzstd 1.5.7" % ns,
                       "parallelism": "frames sharded by rank; payload all-gatherv over RCCL after the timed steps" if world > 1 else "single GPU"}}
    if rank == 0:
        line["compress"] = {"value": round(world * F * FRAME * steps / c_el / 1e9, 3), "ms_per_step": round(c_el / steps * 1e3, 3), "kernels": kernels_obj(cctx, c_k),
                            "regime": e1f_regime(cctx, c_k, F)}
        line["decompress"] = {"value": round(world * F * FRAME * steps / d_el / 1e9, 3), "ms_per_step": round(d_el / steps * 1e3, 3), "kernels": kernels_obj(dctx, d_k)}
        allk = dict(c_k)
        line["roofline"], _ = roofline(cctx, allk, steps, 2 * (F * FRAME + ctotal), step_s * 1e3, F)
        line["roofline"]["per_kernel"] = per_kernel_roofline(cctx, c_k, steps, encode_own_bytes(sec, F, FRAME))
        line["roofline"]["per_kernel"].update(per_kernel_roofline(dctx, d_k, steps, decode_own_bytes(sec, F, FRAME)))
        if world == 1 and not args.no_cpu_baseline:
            # the same round trip on the host: libzstd compress + decompress over a bounded sample, harmonic combination of the two rates
            nsb = min(ns, 2048)
            offs_r = np.arange(nsb + 1, dtype=np.uint64) * np.uint64(FRAME)
            cbc = cpu_baseline(False, np.ascontiguousarray(sample_raw[:nsb]), offs_r, nsb, 0, nsb * FRAME, passes=3)
            blob, boffs = sample_blob(want, nsb)
            cbd = cpu_baseline(True, blob, boffs, nsb, FRAME, nsb * FRAME, passes=3)
            rt = 1.0 / (1.0 / cbc["value"] + 1.0 / cbd["value"])
            line["cpu_baseline"] = {"value": round(rt, 3), "unit": "GB/s", "cores": cbc["cores"], "kind": "reference", "compress": cbc["value"], "decompress": cbd["value"],
                                    "sample": "libzstd 1.5.7 level 3 compress then decompress of %d evenly spaced buffers of the same workload on host threads "
                                              "(median of 3 passes each at the better of 64 / all threads); value = 1 / (1 / compress + 1 / decompress)" % nsb}
        if gather_ms is not None:
            line["allgatherv"] = {"ms": round(gather_ms, 3), "bytes_per_rank": int(ctotal), "GBps_per_rank_received": round((world - 1) * ctotal / gather_ms / 1e6, 2)}
        if not quiet:
            emit(line)
    cctx.close(); dctx.close()
    del raw, slots, back
    torch.cuda.empty_cache()
    return line


def bench_host_api(raw_np, frames, csizes, counts=(8192, 65536)):
    """What a python-zstandard user sees (SURVEY 8(d) "so nobody is misled"): ZstdDecompressor.multi_decompress_to_buffer /
    ZstdCompressor.multi_compress_to_buffer through Python on HOST buffers -- packing, H2D, kernels, D2H, all inside the timed call -- for
    the first 8 192 and all 65 536 frames of the line's workload. Best of 3 / 2 calls after one warm-up call (arenas, pinned staging); a few
    frames of every call compared with the input / libzstd's frames. The reference libzstd on this host's threads over the same data is the
    line's cpu_baseline (decompress) and compress.cpu_baseline. Never `value`: the headline is the HBM-resident rate."""
    import zstandard_amd as pyz
    out = {"unit": "GB/s", "note": "uncompressed bytes / wall time of ONE Python call on host buffers, PCIe and host packing inclusive"}
    Fall = len(frames)
    offs = np.zeros(Fall + 1, dtype=np.uint64); offs[1:] = np.cumsum(csizes)
    blob = b"".join(frames)
    d, c = pyz.ZstdDecompressor(), pyz.ZstdCompressor(level=3)
    for F in counts:
        if F > Fall:
            continue
        segs_c = np.zeros((F, 2), dtype=np.uint64); segs_c[:, 0] = offs[:F]; segs_c[:, 1] = np.asarray(csizes[:F], dtype=np.uint64)
        bws = pyz.BufferWithSegments(memoryview(blob)[: int(offs[F])], segs_c.tobytes())
        sizes = np.full(F, FRAME, dtype=np.uint64).tobytes()
        probe = (0, F // 2 + 1, F - 1)
        r = d.multi_decompress_to_buffer(bws, decompressed_sizes=sizes); del r
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); r = d.multi_decompress_to_buffer(bws, decompressed_sizes=sizes); t1 = time.perf_counter()
            best = min(best, t1 - t0)
            assert len(r) == F and all(r[i].tobytes() == raw_np[i].tobytes() for i in probe), "host API decompress differs from the input"
            del r
        rec = {"decompress": round(F * FRAME / best / 1e9, 2)}
        segs_r = np.zeros((F, 2), dtype=np.uint64); segs_r[:, 0] = np.arange(F, dtype=np.uint64) * FRAME; segs_r[:, 1] = FRAME
        rb = pyz.BufferWithSegments(memoryview(raw_np[:F]).cast("B"), segs_r.tobytes())
        r = c.multi_compress_to_buffer(rb); del r
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); r = c.multi_compress_to_buffer(rb); t1 = time.perf_counter()
            best = min(best, t1 - t0)
            assert len(r) == F and all(r[i].tobytes() == frames[i] for i in probe), "host API compress differs from libzstd 1.5.7"
            del r
        rec["compress"] = round(F * FRAME / best / 1e9, 2)
        out["frames_%d" % F] = rec
    # the device slots the two calls fan a batch out over inside the call (zhip_compress_batch / zhip_decompress_batch: every visible device, or ZHIP_DEVICES)
    import ctypes
    try:
        devs = (ctypes.c_int * 64)()
        nd = pyz._lib.lib().zhip_batch_devices(devs, 64)
        out["devices"] = nd
        out["device_slots"] = list(devs[:nd])
    except AttributeError:                                          # (an older library under ZHIP_LIB, for A/B runs: one device, the current one)
        out["devices"] = 1
    return out


def host_api_child(args):
    """`python bench.py --host-api-child SLOTS`: the host-buffer calls (multi_compress_to_buffer / multi_decompress_to_buffer through Python) with the batch cut over
    several device slots INSIDE the call (zhip_compress_batch / zhip_decompress_batch: DESIGN.md section 6). A process of its own: the device list is read once per
    process, and a first run on real multi-GPU hardware must not be able to take the parent's line down. Prints one JSON object."""
    if args.host_api_child == "all":
        os.environ.pop("ZHIP_DEVICES", None)
    else:
        os.environ["ZHIP_DEVICES"] = args.host_api_child
    from tests.corpus import Corpus
    F = args.frames
    dev = torch.device("cuda", 0)
    raw_np = Corpus(device=dev, mix=args.mix).frames(0, F, chunk=256).cpu().numpy()
    torch.cuda.empty_cache()
    frames, csizes = compress_on_host(raw_np, FRAME)
    out = bench_host_api(raw_np, frames, csizes, counts=(F,))
    print(json.dumps(out), flush=True)


def host_api_over_slots(slots, frames, timeout=420):
    """the child above, with a time limit; its object, or what went wrong"""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-api-child", slots, "--frames", str(frames)], capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            return {"error": "exit code %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:200] if r.stderr.strip() else "")}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return {"error": "no result within %d s" % timeout}
    except (ValueError, IndexError) as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def bench_blocks(args, rank, world, dev, steps=None, warmup=None, quiet=False):
    """Frames of SEVERAL blocks (inputs above 128 KiB; not a BASELINE.json config -- configs[0]'s 1 MiB size as a batch): 2 048 x 1 MiB per GPU,
    each source eight consecutive 128 KiB corpus pieces. Decompression runs the phase-split kernels' several-block mode; compression of a batch
    this small stays in the generic kernel. Every frame compared with libzstd's, every byte with the input."""
    from zstandard_amd.device import DeviceBatchContext
    from tests.corpus import Corpus
    steps = steps or args.steps; warmup = args.warmup if warmup is None else warmup
    F, item = 2048, 1 << 20
    per = item // FRAME
    raw = Corpus(device=dev, mix=args.mix).frames(rank * F * per, F * per, chunk=256).reshape(F, item).contiguous()
    raw_np = raw.cpu().numpy()
    frames, csizes = compress_on_host(raw_np, item)
    job = Job(world, dev)
    ctx = DeviceBatchContext(); ctx.set_size_hint(item)
    d_el, d_k, _ = run_decompress(job, ctx, frames, csizes, raw, item, steps, warmup)
    ctx.close(); ctx = DeviceBatchContext(); ctx.set_size_hint(item)
    c_steps = min(steps, 2)
    c_el, ctotal, c_k = run_compress(job, ctx, raw, frames, item, c_steps, 1)
    line = {"metric": "GB/s uncompressed throughput, batch decompress of 1 MiB level-3 frames (several blocks each)", "value": round(world * F * item * steps / d_el / 1e9, 3),
            "unit": "GB/s", "n_gpus": world, "steps": steps, "ms_per_step": round(d_el / steps * 1e3, 3),
            "config": {"workload": "multi_decompress_to_buffer / multi_compress_to_buffer (device-resident): %d x 1 MiB sources per GPU, level 3, frames of several blocks" % F,
                       "frames_per_gpu": F, "frame_bytes": item, "level": 3, "compression_ratio": round(F * item / float(csizes.sum()), 3)}}
    if rank == 0:
        line["kernels"] = kernels_obj(ctx, d_k)
        line["compress"] = {"value": round(world * F * item * c_steps / c_el / 1e9, 3), "ms_per_step": round(c_el / c_steps * 1e3, 1), "bit_exact_vs_libzstd": True,
                            "kernel": "zhip_encode_frames_kernel (one wave per source: batches below ~8 192 sources per 256 KiB stay at the search's latency bound)"}
        if not quiet:
            emit(line)
    ctx.close()
    del raw
    torch.cuda.empty_cache()
    return line


def spawn_ranks(n):
    """re-run this command line as n ranks of one node: python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1
    --master-port <a free one> bench.py <the same arguments>. Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")               # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_launch(args, rank, local_rank, world):
    """--dry-launch: everything a multi-rank run does BEFORE it touches a kernel -- the process group (RCCL where there are GPUs, gloo where there are
    none: the CPU test of the launch path), every rank's shard of the workload (frames [rank * F, (rank + 1) * F), the reference's contiguous
    partition, c-ext/compressor.c:1127-1216), agreement on it across ranks, a barrier -- then rank 0 prints one JSON line and everybody leaves."""
    import torch.distributed as dist
    have_gpu = torch.cuda.is_available() and torch.cuda.device_count() > local_rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if have_gpu:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dev = torch.device("cuda", local_rank)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cpu")
    try:
        F = args.frames
        mine = torch.tensor([rank * F, (rank + 1) * F, HOST_THREADS], dtype=torch.int64, device=dev)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        bounds = [(int(t[0]), int(t[1])) for t in got]
        assert bounds[rank] == (rank * F, (rank + 1) * F)
        assert bounds[0][0] == 0 and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1)) and bounds[-1][1] == world * F, \
            "the ranks' shards do not tile the workload: %r" % (bounds,)
        t = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # the collective the timed region ends with
        assert int(t.item()) == world - 1
        dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_launch": True, "n_gpus": world, "backend": "nccl" if have_gpu else "gloo", "frames_per_gpu": F,
                              "partition": bounds, "host_threads_per_rank": int(got[0][2])}))
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=65536, help="frames per GPU (BASELINE config: 65536)")
    ap.add_argument("--docs", type=int, default=262144, help="--config dict: documents per GPU (BASELINE config: 262144)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compress-frames", type=int, default=65536,
                    help="after the timed decompress steps, also time multi_compress_to_buffer on this many of the same inputs (0 = skip)")
    ap.add_argument("--config", choices=["decompress", "compress", "dict", "roundtrip", "blocks"], default=None,
                    help="decompress (default) is the BASELINE.json headline; compress / dict / roundtrip are configs[2] / [3] / [4] as their own lines")
    ap.add_argument("--extra", action="store_true", help="carry the sub-objects at N > 1 too (default: N == 1 only -- the scaling runs keep to the headline)")
    ap.add_argument("--no-extra", action="store_true",
                    help="default config only: skip the 'dict' (configs[3]), 'roundtrip' (configs[4]) and 'blocks' (frames of several blocks) sub-objects the line otherwise carries")
    ap.add_argument("--no-host-api", action="store_true", help="skip the 'host_api' sub-object (the Python-visible calls on host buffers, PCIe inclusive)")
    ap.add_argument("--mix", choices=["silesia", "default"], default="silesia", help="class mix of the 128 KiB corpus (tests/corpus.py)")
    ap.add_argument("--direction", choices=["decompress", "compress"], default=None, help="older spelling of --config")
    ap.add_argument("--dry-launch", action="store_true", help="form the process group, agree on the partition, exit (no kernels; gloo where there is no GPU)")
    ap.add_argument("--host-api-child", default=None, metavar="SLOTS",
                    help="(internal) measure the host-buffer calls over these device slots ('all' = every visible device, or a ZHIP_DEVICES list) in this process and print the object")
    args = ap.parse_args()
    if args.host_api_child:
        return host_api_child(args)
    config = args.config or args.direction or "decompress"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: one rank per GPU through torch.distributed.run on this node (what the driver's own command line does);
        # rank 0 prints the one JSON line, the launcher's chatter goes to stderr, this process only waits and hands the exit code on
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE (%d) and --gpus (%d) disagree: launch with --nproc-per-node == --gpus, or let bench.py spawn its ranks" % (world, args.gpus)
    global HOST_THREADS
    HOST_THREADS = max(1, min(64, (os.cpu_count() or 1) // world))
    if args.dry_launch:
        return dry_launch(args, rank, local_rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # this process's host-buffer calls stay on its own device (the library would otherwise fan a large batch out over every visible device: measured in a process of
    # its own, host_api_child, so that nothing a first multi-device run may do can take the headline with it)
    os.environ.setdefault("ZHIP_DEVICES", str(local_rank))
    global USE_DIST
    # ZHIP_BENCH_FORCE_DIST=1: take the process-group path (RCCL init, barriers, max over ranks, the payload all-gatherv) with ONE rank too --
    # the only way to exercise it on a single-GPU box
    USE_DIST = world > 1 or bool(os.environ.get("ZHIP_BENCH_FORCE_DIST"))
    if USE_DIST:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        if config == "dict":
            bench_dict(args, rank, world, dev)
        elif config == "roundtrip":
            bench_roundtrip(args, rank, world, dev)
        elif config == "blocks":
            bench_blocks(args, rank, world, dev)
        else:
            bench_frames(args, config, rank, world, dev)
    finally:
        if USE_DIST:
            torch.distributed.destroy_process_group()
        if PENDING_LINE is not None:
            sys.stdout.flush()
            try:
                C.CDLL(None).fflush(None)                            # whatever native libraries still hold in stdio's buffer goes first
            except OSError:
                pass
            print(PENDING_LINE, flush=True)


def bench_frames(args, config, rank, world, dev):
    import zstandard_amd as zstd  # noqa: F401
    from zstandard_amd.device import DeviceBatchContext
    from tests.corpus import Corpus

    F = args.frames
    t0 = time.time()
    corpus = Corpus(device=dev, mix=args.mix)
    raw = corpus.frames(rank * F, F, chunk=256)                      # [F, FRAME] uint8 in HBM (this rank's shard)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    ref, ref_kind = _ref_lib()
    t0 = time.time()
    raw_np = raw.cpu().numpy()
    frames, csizes = compress_on_host(raw_np, FRAME)
    t_comp = time.time() - t0
    ctotal = int(csizes.sum())
    job = Job(world, dev)
    ctx = DeviceBatchContext()
    cfg = {"frames_per_gpu": F, "frame_bytes": FRAME, "level": 3, "compression_ratio": round(F * FRAME / ctotal, 3), "corpus_mix": args.mix,
           "parallelism": "frames sharded by rank, no data-path collective"}
    nsample = min(F, 16384)

    if config == "compress":
        if os.environ.get("ZHIP_BENCH_PREFRAG"):                    # experiment toggle (DESIGN 4.2, E1f's two regimes): VRAM the size of the decode arenas allocated and released first
            junk = [torch.empty(int(g) << 30, dtype=torch.uint8, device=dev) for g in os.environ["ZHIP_BENCH_PREFRAG"].split(",")]
            for t in junk: t.fill_(1)
            torch.cuda.synchronize(); del junk, t; torch.cuda.empty_cache()
        elapsed, ctot2, ktimes = run_compress(job, ctx, raw, frames, FRAME, args.steps, args.warmup)
        ms = elapsed / args.steps * 1e3
        cfg["workload"] = "multi_compress_to_buffer (device-resident): %d x 128 KiB Silesia-like inputs per GPU, level 3" % F
        line = {"metric": "GB/s uncompressed throughput, batch compress of 128 KiB inputs at level 3 (bit-exact vs libzstd 1.5.7)",
                "value": round(world * F * FRAME * args.steps / elapsed / 1e9, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u8", "data": "synthetic", "config": cfg}
        if rank == 0:
            line["kernels"] = kernels_obj(ctx, ktimes)
            line["regime"] = e1f_regime(ctx, ktimes, F)
            line["roofline"], _ = roofline(ctx, ktimes, args.steps, F * FRAME + ctotal, ms, F)
            if world == 1 and not args.no_cpu_baseline:
                offs = np.arange(nsample + 1, dtype=np.uint64) * np.uint64(FRAME)
                cb = cpu_baseline(False, np.ascontiguousarray(raw_np[:nsample]), offs, nsample, 0, nsample * FRAME)
                cb.update({"kind": ref_kind, "sample": "libzstd 1.5.7 ZSTD_compressStream2(e_end) level 3 over the first %d inputs of the same workload; "
                                                       "median of %d passes at the better of 64 / all host threads" % (nsample, cb["passes"])})
                line["cpu_baseline"] = cb
            emit(line)
        return

    elapsed, ktimes, out_sizes = run_decompress(job, ctx, frames, csizes, raw, FRAME, args.steps, args.warmup)
    if USE_DIST:
        import torch.distributed as dist
        # the only cross-rank exchange the path needs: the segment table of the sharded result (payload stays per GPU)
        gathered = [torch.zeros_like(out_sizes) for _ in range(world)]
        dist.all_gather(gathered, out_sizes)
    ms = elapsed / args.steps * 1e3
    cfg["workload"] = ("multi_decompress_to_buffer (device-resident): %d x 128 KiB Silesia-like frames per GPU, level 3, frames compressed by "
                       "libzstd 1.5.7" % F)
    line = {
        "metric": "GB/s uncompressed throughput, batch decompress of 128 KiB level-3 frames (the decompress half of BASELINE.json's metric; "
                  "the compress half on the same inputs is this line's 'compress' object)",
        "value": round(world * F * FRAME * args.steps / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": cfg,
    }
    sec = frame_sections(frames)
    if rank == 0:
        line["kernels"] = kernels_obj(ctx, ktimes)
        line["roofline"], _ = roofline(ctx, ktimes, args.steps, F * FRAME + ctotal, ms, F)
        line["roofline"]["per_kernel"] = per_kernel_roofline(ctx, ktimes, args.steps, decode_own_bytes(sec, F, FRAME))
        line["roofline"]["sections_per_frame"] = {k: round(v, 1) for k, v in sec.items()}
        if world == 1 and not args.no_cpu_baseline:
            blob, boffs = sample_blob(frames, nsample)
            cb = cpu_baseline(True, blob, boffs, nsample, FRAME, nsample * FRAME)
            cb.update({"kind": ref_kind, "sample": "libzstd 1.5.7 ZSTD_decompressStream over the first %d frames of the same workload; median of %d "
                                                   "passes at the better of 64 / all host threads" % (nsample, cb["passes"])})
            line["cpu_baseline"] = cb
        line["setup_s"] = {"generate": round(t_gen, 1), "host_compress": round(t_comp, 1)}
    if args.compress_frames > 0:
        # the other half of BASELINE.json's metric, on the same inputs (the timed decompress region above is over). The compressor is
        # its own object in the reference API: release the decode direction's working set and start from a fresh context
        Fc = min(F, args.compress_frames)
        if os.environ.get("ZHIP_BENCH_KEEP_DECODE_CTX"):           # experiment toggle (DESIGN 4.2, E1f's two regimes): the tables come from VRAM the decode arenas never used
            keep_ctx = ctx
        else:
            ctx.close()
        torch.cuda.empty_cache()
        ctx = DeviceBatchContext()
        c_elapsed, c_total, c_k = run_compress(job, ctx, raw[:Fc], frames[:Fc], FRAME, 3, 2)
        if rank == 0:
            c_ms = c_elapsed / 3 * 1e3
            line["compress"] = {"value": round(world * Fc * FRAME * 3 / c_elapsed / 1e9, 3), "unit": "GB/s", "frames_per_gpu": Fc, "steps": 3,
                                "ms_per_step": round(c_ms, 3), "bit_exact_vs_libzstd": True, "kernels": kernels_obj(ctx, c_k)}
            line["compress"]["regime"] = e1f_regime(ctx, c_k, Fc)
            line["compress"]["roofline"], _ = roofline(ctx, c_k, 3, Fc * FRAME + c_total, c_ms, Fc)
            line["compress"]["roofline"]["per_kernel"] = per_kernel_roofline(ctx, c_k, 3, encode_own_bytes(sec, Fc, FRAME))
            if world == 1 and not args.no_cpu_baseline:
                ns = min(Fc, nsample)
                offs = np.arange(ns + 1, dtype=np.uint64) * np.uint64(FRAME)
                cb = cpu_baseline(False, np.ascontiguousarray(raw_np[:ns]), offs, ns, 0, ns * FRAME)
                cb.update({"kind": ref_kind, "sample": "libzstd 1.5.7 level 3, first %d inputs; median of %d passes at the better of 64 / all host threads"
                                                       % (ns, cb["passes"])})
                line["compress"]["cpu_baseline"] = cb
    ctx.close()
    if rank == 0 and isinstance(line.get("compress"), dict) and line["compress"].get("frames_per_gpu") == F:
        # BASELINE.json's metric is "compress+decompress": both directions over the same bytes, one after the other -- the harmonic combination
        cv, dv = line["compress"]["value"], line["value"]
        line["combined"] = {"value": round(1.0 / (1.0 / cv + 1.0 / dv), 3), "unit": "GB/s", "compress": cv, "decompress": dv,
                            "formula": "1 / (1 / compress + 1 / decompress): uncompressed GB/s of a batch compressed and then decompressed on the same GPU(s)"}
    if world == 1 and rank == 0 and not args.no_host_api and F >= 8192:
        torch.cuda.empty_cache()
        t0 = time.time()
        try:
            line["host_api"] = bench_host_api(raw_np, frames, csizes)
            line["host_api"]["wall_s"] = round(time.time() - t0, 1)
        except Exception as e:                                      # noqa: BLE001 -- keep the headline
            line["host_api"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # the same two calls with the batch cut over every device of the node inside the call -- where there is more than one (ZHIP_BENCH_HOST_API_SLOTS=0,0 forces
        # two slots on one GPU: the plumbing's price where it cannot help)
        slots = os.environ.get("ZHIP_BENCH_HOST_API_SLOTS") or ("all" if torch.cuda.device_count() > 1 else None)
        if slots and isinstance(line.get("host_api"), dict):
            line["host_api"]["all_devices"] = host_api_over_slots(slots, F)
    if not args.no_extra and F >= 65536 and (world == 1 or args.extra):
        # BASELINE.json configs[3] and configs[4] ride on the default line as sub-objects (each with its own roofline and cpu_baseline), so
        # that the driver's one run records every config; their own timed regions start after this line's is over. A failure there is
        # reported in the sub-object and never loses the headline.
        del raw, raw_np, frames
        torch.cuda.empty_cache()
        for key, fn in (("dict", bench_dict), ("roundtrip", bench_roundtrip), ("blocks", bench_blocks)):
            t0 = time.time()
            try:
                sub = fn(args, rank, world, dev, steps=3, warmup=1, quiet=True)
                sub["wall_s"] = round(time.time() - t0, 1)
            except Exception as e:                                  # noqa: BLE001 -- keep the headline
                if USE_DIST and world > 1:                          # the sub-benchmarks contain collectives: a rank that swallowed its error would leave the others waiting
                    raise
                sub = {"error": "%s: %s" % (type(e).__name__, e)}
            if rank == 0:
                line[key] = sub
            torch.cuda.empty_cache()
    if rank == 0:
        emit(line)


if __name__ == "__main__":
    main()
