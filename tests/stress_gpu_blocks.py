"""GPU stress of the several-block modes through the Python API against libzstd 1.5.7: python tests/stress_gpu_blocks.py SEED [SOURCES].
Sources of 1-9 blocks built to break the flat search's assumption now and then (tests/stress_emu_encode_blocks.make) plus small neighbours:
multi_compress_to_buffer with the flat search forced on (ZHIP_MBC_MIN=0, read when the thread's context is created) -- every frame must
be libzstd's --, then multi_decompress_to_buffer of libzstd's frames at several levels (the decode kernels' several-block mode) and one-shot
calls on the largest source. The emulator twins are stress_emu_encode_blocks.py / stress_emu_decode_any.py. Not collected by pytest."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import zstandard_amd as zstd
from tests import reflib
from tests.corpus import Corpus
from tests.stress_emu_encode_blocks import make

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 48
ref = reflib.RefZstd(); corpus = Corpus(); rng = np.random.default_rng(seed)
raws = [make(rng, corpus) for _ in range(count)] + [corpus.frame_bytes(int(rng.integers(0, 500)))[:int(rng.integers(1, 131073))] for _ in range(8)]
box = {}


def run():
    try:
        t0 = time.time()
        flags = int(rng.choice([5, 7]))
        res = zstd.ZstdCompressor(level=3, write_checksum=flags == 7).multi_compress_to_buffer(raws)
        box["c"] = sum(1 for i, r in enumerate(raws) if res[i].tobytes() != ref.compress(r, level=3, flags=flags))
        frames = [ref.compress(r, level=int(rng.choice([1, 3, 3, 5, 9, 19])), flags=int(rng.choice([5, 7]))) for r in raws]
        back = zstd.ZstdDecompressor().multi_decompress_to_buffer(frames)
        box["d"] = sum(1 for i, r in enumerate(raws) if back[i].tobytes() != r)
        k = max(range(len(raws)), key=lambda i: len(raws[i]))
        box["o"] = int(zstd.ZstdDecompressor().decompress(frames[k]) != raws[k]) + int(zstd.ZstdCompressor(level=3).compress(raws[k]) != ref.compress(raws[k], level=3))
        # with a dictionary (trained / raw content by turns): frames of several blocks whose matches reach below the frame's first byte from any
        # block -- the dictionary instantiation of the several-block K3 -- and the encoder's table-copy mode over several blocks
        from tests.test_oracle_vs_golden import _dict_vectors
        dicts, _ = _dict_vectors()
        dd = dicts["trained"] if seed % 2 else corpus.frame_bytes(950 + seed)[:60000]
        zd = zstd.ZstdCompressionDict(dd)
        draws = [r[:int(rng.integers(131073, 700000))] for r in raws[:12] if len(r) > 140000] + [dd[300:9000] * 30, raws[-1]]
        dframes = [ref.compress(r, level=3, flags=7, dict_data=dd) for r in draws]
        dback = zstd.ZstdDecompressor(dict_data=zd).multi_decompress_to_buffer(dframes)
        box["dd"] = sum(1 for i, r in enumerate(draws) if dback[i].tobytes() != r)
        dres = zstd.ZstdCompressor(level=3, dict_data=zd, write_checksum=True).multi_compress_to_buffer(draws)
        box["dc"] = sum(1 for i in range(len(draws)) if dres[i].tobytes() != dframes[i])
        box["t"] = time.time() - t0
    except Exception as e:      # noqa: BLE001
        box["error"] = repr(e)


os.environ["ZHIP_MBC_MIN"] = "0"
t = threading.Thread(target=run); t.start(); t.join()
bad = box.get("c", 1) + box.get("d", 1) + box.get("o", 1) + box.get("dd", 1) + box.get("dc", 1) + (1 if "error" in box else 0)
print("gpu blocks stress", seed, "sources", len(raws), "MiB", round(sum(map(len, raws)) / 2**20, 1), "compress mismatches", box.get("c"), "decompress mismatches", box.get("d"),
      "one-shot mismatches", box.get("o"), "dictionary decompress / compress mismatches", box.get("dd"), box.get("dc"), box.get("error", ""), "%.1fs" % box.get("t", 0))
sys.exit(1 if bad else 0)
