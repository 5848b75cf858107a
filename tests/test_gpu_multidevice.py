"""One call, every device: zhip_compress_batch / zhip_decompress_batch -- what multi_compress_to_buffer / multi_decompress_to_buffer call --
cut a batch over the node's devices inside the call, like the reference cuts it over its worker threads (c-ext/compressor.c:1127-1298). A one-GPU box tests the
split with ZHIP_DEVICES=0,0: two device slots (two host threads, two contexts, two staging areas) on the one GPU. The collections must be byte-identical to the
single-device call's and to libzstd's frames; the first failing item must be the lowest index whichever slot found it. The device list is read once per
process, so every configuration runs in a process of its own."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, json, sys
sys.path.insert(0, %(root)r)
import numpy as np
import zstandard_amd as zstd
from tests.corpus import Corpus
from tests import reflib
corpus = Corpus()
rng = np.random.default_rng(11)
raws = []
for i in range(700):
    f = corpus.frame_bytes(i %% 97)
    k = int(rng.integers(1, 131072)) if i %% 5 else 131072
    raws.append(f[:k])
raws += [b"", b"x", corpus.frame_bytes(3) * 3]                     # an empty source, one byte, a source of several blocks
n_dev = zstd._lib.lib().zhip_batch_devices(None, 0)
c = zstd.ZstdCompressor(level=3, write_checksum=True)
res = c.multi_compress_to_buffer(raws)
frames = [res[i].tobytes() for i in range(len(res))]
ref = reflib.checker()
ok_ref = all(frames[i] == ref.compress(raws[i], flags=3) for i in range(0, len(raws), 7))       # (flags: content size + checksum)
back = zstd.ZstdDecompressor().multi_decompress_to_buffer(frames)
ok_back = len(back) == len(raws) and all(back[i].tobytes() == raws[i] for i in range(len(raws)))
# the first failing item is the lowest index, whichever device slot holds it
bad = list(frames)
hit = []
for where in (len(bad) - 9, len(bad) // 2 + 5, 17):
    b = bytearray(bad[where]); b[len(b) // 2] ^= 0x5A; b[-3] ^= 0x11; bad[where] = bytes(b)
    try:
        zstd.ZstdDecompressor().multi_decompress_to_buffer(bad)
        hit.append(None)
    except zstd.ZstdError as e:
        hit.append(str(e))
h = hashlib.sha256()
for f in frames:
    h.update(len(f).to_bytes(8, "little")); h.update(f)
print(json.dumps({"devices": n_dev, "segments": len(res), "sha": h.hexdigest(),
                  "ok_ref": ok_ref, "ok_back": ok_back, "errors": hit}))
'''


def _run(devices):
    env = dict(os.environ)
    env.pop("ZHIP_DEVICES", None)
    if devices:
        env["ZHIP_DEVICES"] = devices
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_two_device_slots_give_the_single_device_collection():
    one = _run("0")
    two = _run("0,0")
    three = _run("0,0,0")
    assert one["devices"] == 1 and two["devices"] == 2 and three["devices"] == 3
    for r in (one, two, three):
        assert r["ok_ref"] and r["ok_back"], r
    assert one["sha"] == two["sha"] == three["sha"] and one["segments"] == two["segments"] == three["segments"] == 703
    # the damaged frames: item 694 fails first, then 356 (in the other half), then 17 -- always the lowest damaged index is the one reported
    assert one["errors"] == two["errors"] == three["errors"], (one["errors"], two["errors"])
    assert all(e for e in one["errors"]) and "item 694" in one["errors"][0] and "item 356" in one["errors"][1] and "item 17" in one["errors"][2], one["errors"]
