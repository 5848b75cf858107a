"""Throughput of MULTI-BLOCK frames (inputs above 128 KiB), device-resident. Decompression: the phase-split kernels in their several-block
mode (rounds 1-2: one wave per frame in the generic kernel); compression: the generic kernel (DESIGN.md 4.1 / 4.2,
VERDICT r02 "missing" 3). 1 MiB inputs = BASELINE configs[0]'s size, built from 8 consecutive
128 KiB corpus frames; frames compared with libzstd's (compress) and with the inputs (decompress).
Usage: python tests/multiblock_rate.py [frames] [KiB per frame]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zstandard_amd.device import DeviceBatchContext
from tests.corpus import Corpus
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
KIB = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
item = KIB * 1024
dev = torch.device("cuda", 0)
per = item // 131072
raw = Corpus(device=dev, mix="silesia").frames(0, F * per, chunk=256).reshape(F, item).contiguous()
raw_np = raw.cpu().numpy()
frames, csizes = bench.compress_on_host(raw_np, item)
out = {"frames": F, "frame_KiB": KIB, "ratio": round(F * item / float(csizes.sum()), 3)}
job = bench.Job(1, dev)
ctx = DeviceBatchContext(); ctx.set_size_hint(item)
el, kt, _ = bench.run_decompress(job, ctx, frames, csizes, raw, item, 3, 1)
out["decompress_GBps"] = round(F * item * 3 / el / 1e9, 2); out["decompress_ms"] = round(el / 3 * 1e3, 2)
out["decompress_kernels"] = bench.kernels_obj(ctx, kt)
ctx.close(); ctx = DeviceBatchContext(); ctx.set_size_hint(item)
el, total, _ = bench.run_compress(job, ctx, raw, frames, item, 2, 1)
out["compress_GBps"] = round(F * item * 2 / el / 1e9, 3); out["compress_ms"] = round(el / 2 * 1e3, 1); out["bit_exact_vs_libzstd"] = True
ctx.close()
print(json.dumps(out))
