"""What write_checksum costs the compress step (device-resident, 128 KiB sources): the same batch without and with the checksum flag, every frame against libzstd's.
Usage: python tests/tools/compress_checksum_cost.py [frames]"""
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
out = {"frames": F}
job = bench.Job(1, dev)
for name, ck in (("plain", False), ("checksum", True)):
    flags = reflib.DEFAULT_FLAGS | (reflib.F_CHECKSUM if ck else 0)
    n = min(F, 512)
    want = [ref.compress(raw_np[i].tobytes(), level=3, flags=flags) for i in range(n)]
    frames, _ = bench.compress_on_host(raw_np, bench.FRAME)
    if ck:      # the bench's gate compares every frame: give it libzstd's checksummed frames for the first n, and check only those
        frames = want + frames[n:]
    ctx = DeviceBatchContext(write_checksum=ck)
    item = bench.FRAME
    bound = (item + (item >> 8) + 64 + 15) & ~15
    src_segs = bench.segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64), dev)
    dst_segs = bench.segs(np.arange(F, dtype=np.int64) * bound, np.full(F, bound, dtype=np.int64), dev)
    dst = torch.zeros(F * bound, dtype=torch.uint8, device=dev); osz = torch.zeros(F, dtype=torch.int64, device=dev); st = torch.zeros(F, dtype=torch.int32, device=dev)
    el, kt = job.timed(lambda: ctx.compress(raw.reshape(-1), src_segs, dst, dst_segs, osz, st), ctx, bench.ENC_KERNELS, 3, 1)
    assert int(st.abs().max().item()) == 0
    sizes = osz.cpu().numpy(); o = dst.view(F, bound)[:n].cpu().numpy()
    assert all(o[i, : sizes[i]].tobytes() == want[i] for i in range(n)), "frames differ from libzstd's"
    out[name] = {"ms": round(el / 3 * 1e3, 2), "kernels": {ctx.kernel_name(k).replace("zhip_encode_", "").replace("_kernel", ""): round(v[0], 2) for k, v in kt.items() if v[1]}}
    ctx.close(); del dst
print(json.dumps(out))
