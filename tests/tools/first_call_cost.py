"""What the first large compress call of a context costs beside the later ones (the table placement pick: eight probe launches on eight candidate allocations, zhip_compress_batch_device):
python tests/tools/first_call_cost.py [sources] -- device-resident sources of 128 KiB, wall time of calls 1, 2, 3 of one context (the first also reserves the tables and arenas), and of a
second context's first call in the same process with ZHIP_E1F_PICK=0 (reservations only). Measurement aid; run on the GPU box."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from tests.corpus import Corpus
from zstandard_amd.device import DeviceBatchContext

F = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
item = 131072
dev = torch.device("cuda", 0)
raw = Corpus(device=dev, mix="silesia").frames(0, F, chunk=256)
bound = item + 1024
src_segs = bench.segs(np.arange(F, dtype=np.int64) * item, np.full(F, item, dtype=np.int64), dev)
dst_segs = bench.segs(np.arange(F, dtype=np.int64) * bound, np.full(F, bound, dtype=np.int64), dev)
dst = torch.zeros(F * bound, dtype=torch.uint8, device=dev)
out_sizes = torch.zeros(F, dtype=torch.int64, device=dev); status = torch.zeros(F, dtype=torch.int32, device=dev)
out = {"sources": F}
for name, env in (("with_pick", None), ("without_pick", "0")):
    if env is not None: os.environ["ZHIP_E1F_PICK"] = env
    ctx = DeviceBatchContext()
    os.environ.pop("ZHIP_E1F_PICK", None)
    ms = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.compress(raw.view(-1), src_segs, dst, dst_segs, out_sizes, status)
        torch.cuda.synchronize(); ms.append(round((time.perf_counter() - t0) * 1e3, 1))
    assert int(status.abs().max().item()) == 0
    out[name + "_calls_ms"] = ms
    out[name + "_table_pick"] = ctx.table_pick()
    ctx.close(); torch.cuda.empty_cache()
print(json.dumps(out))
