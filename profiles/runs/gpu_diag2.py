"""bring-up probe for the decode pipeline: the edge-frame batch, alone and together, with per-frame pipeline records"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import zstandard_amd as zstd
from tests import reflib
enc = reflib.RefZstd()
raws = [b"", b"foo", b"foo" * 4, b"bar" * 6, b"a" * 1000, b"a" * 131072, bytes(range(256)) * 40,
        b"hello world, hello world, hello there world! " * 500, np.random.default_rng(1).bytes(1 << 17)]
frames = [enc.compress(r) for r in raws]
d = zstd.ZstdDecompressor()
os.environ["ZHIP_DEBUG_PIPE"] = "1"
for i, (f, r) in enumerate(zip(frames, raws)):
    if not r:
        continue
    try:
        out = d.multi_decompress_to_buffer([f])
        print("single", i, "ok" if out[0].tobytes() == r else "WRONG BYTES", flush=True)
    except Exception as e:
        print("single", i, "ERR", e, flush=True)
try:
    out = d.multi_decompress_to_buffer(frames)
    print("batch", [out[i].tobytes() == raws[i] for i in range(len(raws))])
except Exception as e:
    print("batch ERR", e)
