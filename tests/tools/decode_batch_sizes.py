"""The decode step by batch size (device-resident, 128 KiB frames): where the fixed costs of the pipeline's small kernels -- K0's lane-serial walks, the bin pass --
stop paying. Usage: python tests/tools/decode_batch_sizes.py [sizes ...]   (ZHIP_LIB selects the build)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zstandard_amd.device import DeviceBatchContext
from tests.corpus import Corpus
import bench

sizes = [int(x) for x in sys.argv[1:]] or [1, 64, 512, 2048, 8192, 32768]
dev = torch.device("cuda", 0)
F = max(sizes)
raw = Corpus(device=dev, mix="silesia").frames(0, F, chunk=256)
frames, csizes = bench.compress_on_host(raw.cpu().numpy(), bench.FRAME)
job = bench.Job(1, dev)
out = {}
for n in sizes:
    ctx = DeviceBatchContext()
    steps = 50 if n <= 2048 else 10
    el, kt, _ = bench.run_decompress(job, ctx, frames[:n], csizes[:n], raw[:n], bench.FRAME, steps, 3)
    out[n] = {"ms": round(el / steps * 1e3, 3), "k0_k1_ms": round(kt[2][0], 3)}
    ctx.close()
print(json.dumps(out))
