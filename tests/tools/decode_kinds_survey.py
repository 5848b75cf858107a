"""The decode step over frames of kinds no bench line times (device-resident, 128 KiB sources): levels 1 / 3 / 9 / 19, frames without the content size in the header,
incompressible sources (raw blocks), byte runs (RLE blocks), small frames of 16 KiB. A survey for cliffs: ms per step and GB/s of output.  Usage: python tests/tools/decode_kinds_survey.py [frames]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from zstandard_amd.device import DeviceBatchContext
from tests.corpus import Corpus
from tests import reflib
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
raw = Corpus(device=dev, mix="silesia").frames(0, F, chunk=256)
raw_np = raw.cpu().numpy()
ref = reflib.RefZstd()
job = bench.Job(1, dev)
out = {"frames": F}


def run(name, frames, rawt, item):
    cs = np.array([len(x) for x in frames], dtype=np.int64)
    ctx = DeviceBatchContext()
    el, kt, _ = bench.run_decompress(job, ctx, frames, cs, rawt, item, 5, 2)
    out[name] = {"ms": round(el / 5 * 1e3, 3), "GBps": round(len(frames) * item * 5 / el / 1e9, 1),
                 "kernels": {ctx.kernel_name(k).replace("zhip_decode_", "").replace("_kernel", ""): round(v[0], 3) for k, v in kt.items() if v[1]}}
    ctx.close()


for level in (3, 1, 9, 19):
    n = F if level < 19 else min(F, 2048)
    frames, _ = bench.compress_on_host(raw_np[:n], bench.FRAME, None, level)
    run("level_%d" % level + ("" if n == F else "_%d_frames" % n), frames, raw[:n], bench.FRAME)
frames = [ref.compress(raw_np[i].tobytes(), level=3, flags=reflib.F_DICTID) for i in range(min(F, 4096))]            # no content size in the header
run("no_content_size_4096_frames", frames, raw[:len(frames)], bench.FRAME)
rng = np.random.default_rng(1)
rnd = torch.from_numpy(rng.integers(0, 256, (min(F, 4096), bench.FRAME), dtype=np.uint8)).to(dev)
frames, _ = bench.compress_on_host(rnd.cpu().numpy(), bench.FRAME)
run("incompressible_4096_frames", frames, rnd, bench.FRAME)
runs = torch.zeros((min(F, 4096), bench.FRAME), dtype=torch.uint8, device=dev) + 7
frames, _ = bench.compress_on_host(runs.cpu().numpy(), bench.FRAME)
run("one_byte_runs_4096_frames", frames, runs, bench.FRAME)
small = raw.reshape(-1, 16384)[: 8 * F].contiguous()
frames, _ = bench.compress_on_host(small.cpu().numpy(), 16384)
run("16KiB_frames_x8", frames, small, 16384)
print(json.dumps(out))
