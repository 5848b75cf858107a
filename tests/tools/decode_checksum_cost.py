"""What a content checksum costs the decode step (device-resident, 128 KiB frames): the same frames without and with the checksum flag + XXH64 trailer (added on the host with the
xxhash module: flag bit 2 of the frame header descriptor, low 32 bits of XXH64 of the content behind the last block -- RFC 8878 3.1.1.1.1.4 / 3.1.1.2).
Usage: python tests/tools/decode_checksum_cost.py [frames]"""
import json
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import xxhash
from zstandard_amd.device import DeviceBatchContext
from tests.corpus import Corpus
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
raw = Corpus(device=dev, mix="silesia").frames(0, F, chunk=256)
raw_np = raw.cpu().numpy()
frames, csizes = bench.compress_on_host(raw_np, bench.FRAME)
with_ck = []
for i, f in enumerate(frames):
    b = bytearray(f); b[4] |= 4
    with_ck.append(bytes(b) + struct.pack("<I", xxhash.xxh64(raw_np[i].tobytes()).intdigest() & 0xFFFFFFFF))
import numpy as np
out = {"frames": F}
job = bench.Job(1, dev)
for name, fr in (("plain", frames), ("checksum", with_ck)):
    cs = np.array([len(x) for x in fr], dtype=np.int64)
    ctx = DeviceBatchContext()
    el, kt, _ = bench.run_decompress(job, ctx, fr, cs, raw, bench.FRAME, 5, 2)
    out[name] = {"ms": round(el / 5 * 1e3, 3), "kernels": {ctx.kernel_name(k).replace("zhip_decode_", "").replace("_kernel", ""): round(v[0], 3) for k, v in kt.items() if v[1]}}
    ctx.close()
print(json.dumps(out))
