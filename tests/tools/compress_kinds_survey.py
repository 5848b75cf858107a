"""The compress step (level 3, device-resident) over source kinds no bench line times: small sources without a dictionary, incompressible, all-zero, mixed sizes. A survey for cliffs:
ms per step, GB/s of input; a sample of frames against libzstd's.  Usage: python tests/tools/compress_kinds_survey.py"""
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

dev = torch.device("cuda", 0)
ref = reflib.checker()
job = bench.Job(1, dev)
out = {}


def run(name, raw2d, lens=None):
    F, item = raw2d.shape
    lens = np.full(F, item, dtype=np.int64) if lens is None else lens
    bound = (item + (item >> 8) + 64 + 15) & ~15
    src_segs = bench.segs(np.arange(F, dtype=np.int64) * item, lens, dev)
    dst_segs = bench.segs(np.arange(F, dtype=np.int64) * bound, np.full(F, bound, dtype=np.int64), dev)
    dst = torch.zeros(F * bound, dtype=torch.uint8, device=dev); osz = torch.zeros(F, dtype=torch.int64, device=dev); st = torch.zeros(F, dtype=torch.int32, device=dev)
    ctx = DeviceBatchContext(level=3); ctx.set_size_hint(item)
    el, kt = job.timed(lambda: ctx.compress(raw2d.reshape(-1), src_segs, dst, dst_segs, osz, st), ctx, bench.ENC_KERNELS, 2, 1)
    assert int(st.abs().max().item()) == 0
    sizes = osz.cpu().numpy(); n = min(F, 64); o = dst.view(F, bound)[:n].cpu().numpy(); r = raw2d[:n].cpu().numpy()
    assert all(o[i, : sizes[i]].tobytes() == ref.compress(r[i, : lens[i]].tobytes(), level=3) for i in range(n)), name
    total = float(lens.sum())
    out[name] = {"sources": F, "ms": round(el / 2 * 1e3, 2), "GBps": round(total * 2 / el / 1e9, 2), "ratio": round(total / float(sizes.sum()), 2),
                 "kernels": {ctx.kernel_name(k).replace("zhip_encode_", "").replace("_kernel", ""): round(v[0], 2) for k, v in kt.items() if v[1]}}
    ctx.close(); del dst


c = Corpus(device=dev, mix="silesia")
base = c.frames(0, 16384, chunk=256)
run("128KiB_x16384", base)
run("4KiB_x262144_no_dict", base.reshape(-1, 4096)[:262144].contiguous())
run("512B_x262144", base.reshape(-1, 512)[:262144].contiguous())
rng = np.random.default_rng(2)
run("incompressible_128KiB_x8192", torch.from_numpy(rng.integers(0, 256, (8192, 131072), dtype=np.uint8)).to(dev))
run("zeros_128KiB_x8192", torch.zeros((8192, 131072), dtype=torch.uint8, device=dev))
lens = rng.integers(64, 131073, 16384).astype(np.int64)
run("mixed_sizes_x16384", base, lens)
print(json.dumps(out))
