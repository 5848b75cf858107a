"""The compress step by level (device-resident, 128 KiB sources): levels 3 (double-fast: the flat search) and 1 / 2 / -1 (fast strategy: the lane-serial match kernel), a sample of
frames against libzstd's.  Usage: python tests/tools/compress_levels_rate.py [frames]"""
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

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
raw = Corpus(device=dev, mix="silesia").frames(0, F, chunk=256)
raw_np = raw.cpu().numpy()
ref = reflib.checker()
job = bench.Job(1, dev)
item = bench.FRAME
bound = (item + (item >> 8) + 64 + 15) & ~15
src_segs = bench.segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64), dev)
dst_segs = bench.segs(np.arange(F, dtype=np.int64) * bound, np.full(F, bound, dtype=np.int64), dev)
out = {"frames": F}
for level in (3, 2, 1, -1):
    n = min(F, 128)
    want = [ref.compress(raw_np[i].tobytes(), level=level) for i in range(n)]
    ctx = DeviceBatchContext(level=level)
    dst = torch.zeros(F * bound, dtype=torch.uint8, device=dev); osz = torch.zeros(F, dtype=torch.int64, device=dev); st = torch.zeros(F, dtype=torch.int32, device=dev)
    el, kt = job.timed(lambda: ctx.compress(raw.reshape(-1), src_segs, dst, dst_segs, osz, st), ctx, bench.ENC_KERNELS, 2, 1)
    assert int(st.abs().max().item()) == 0
    sizes = osz.cpu().numpy(); o = dst.view(F, bound)[:n].cpu().numpy()
    assert all(o[i, : sizes[i]].tobytes() == want[i] for i in range(n)), "level %d: frames differ from libzstd's" % level
    out["level_%d" % level] = {"ms": round(el / 2 * 1e3, 1), "GBps": round(F * item * 2 / el / 1e9, 2), "ratio": round(F * item / float(sizes.sum()), 3),
                              "kernels": {ctx.kernel_name(k).replace("zhip_encode_", "").replace("_kernel", ""): round(v[0], 2) for k, v in kt.items() if v[1]}}
    ctx.close(); del dst
print(json.dumps(out))
