"""Frames of one block's size that libzstd's block splitter cut into several blocks (level 19), through the host-buffer API: ms per call and GB/s of output.
Usage: python tests/tools/split_block_frames_rate.py [frames]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import zstandard_amd as pyz
from tests.corpus import Corpus
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
raw = Corpus(device=torch.device("cuda", 0), mix="silesia").frames(0, F, chunk=256).cpu().numpy()
out = {"frames": F}
for level in (3, 19):
    frames, _ = bench.compress_on_host(raw, bench.FRAME, None, level)
    d = pyz.ZstdDecompressor()
    r = d.multi_decompress_to_buffer(frames); del r
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = d.multi_decompress_to_buffer(frames); best = min(best, time.perf_counter() - t0)
        assert r[F - 1].tobytes() == raw[F - 1].tobytes() and r[F // 2].tobytes() == raw[F // 2].tobytes()
        del r
    out["level_%d" % level] = {"ms": round(best * 1e3, 2), "GBps": round(F * bench.FRAME / best / 1e9, 2)}
print(json.dumps(out))
