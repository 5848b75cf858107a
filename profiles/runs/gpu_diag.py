"""Step-by-step GPU bring-up probe (prints before every step so a hang is attributable)."""
import faulthandler
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(100, exit=True)


def say(*a):
    print("[diag %.1fs]" % (time.time() - T0), *a, flush=True)


T0 = time.time()
say("import torch")
import torch
say("torch", torch.__version__, "cuda available:", torch.cuda.is_available())
import zstandard_amd as zstd
L = zstd._lib.lib()
say("lib loaded; device count:", L.zhip_device_count(), "last error:", zstd._lib.last_error())
say("selftest:", L.zhip_selftest(), zstd._lib.last_error())
from tests import reflib
enc = reflib.RefZstd() if reflib.have_ref() else reflib.Oracle()
say("encoder:", type(enc).__name__)
d = zstd.ZstdDecompressor()
for name, raw in [("empty", b""), ("foo", b"foo"), ("raw-block", os.urandom(1000)), ("rle", b"a" * 5000),
                  ("text", b"hello world, hello there world! " * 300)]:
    f = enc.compress(raw)
    say("decode", name, len(f), "->", len(raw))
    r = d.multi_decompress_to_buffer([f]) if raw else None
    if raw:
        assert r[0].tobytes() == raw, name
    say("  ok")
from tests.corpus import Corpus
c = Corpus()
raws = [c.frame_bytes(i) for i in range(16)]
frames = [enc.compress(r) for r in raws]
say("decode 16 corpus frames")
t = time.time()
r = d.multi_decompress_to_buffer(frames)
say("  done in %.3fs" % (time.time() - t), all(r[i].tobytes() == raws[i] for i in range(16)))
say("all good")
