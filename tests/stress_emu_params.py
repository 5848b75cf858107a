"""Emulator stress of explicit compression parameters against libzstd 1.5.7: python tests/stress_emu_params.py SEED [N] [blocks].
Random ZstdCompressionParameters fields (window / hash / chain log, minimum match, target length, strategy fast or double-fast) on top
of random levels, sources below and above one block: an accepted frame is libzstd's with the same parameters
(c-ext/compressionparams.c -> ZSTD_CCtx_setParameter), what is not implemented is refused with parameter_unsupported (40).
Not collected by pytest; the bounded version is tests/test_emu_kernels.py::test_explicit_parameters_and_magicless_bit_exact."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import emulib, reflib
from tests.corpus import Corpus

emu = emulib.Emu(); ref = reflib.RefZstd(); corpus = Corpus()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
focus = len(sys.argv) > 3 and sys.argv[3] == 'blocks'      # bias towards windows of a few blocks and sources around whole blocks
rng = np.random.default_rng(seed)
pool = corpus.frame_list(24 * (seed % 30), 12)
bad = refused = ok = 0
t0 = time.time()
try:
    for it in range(rounds):
        kw = {}
        if rng.integers(0, 2): kw["window_log"] = int(rng.integers(10, 22))
        if focus and rng.integers(0, 2): kw["window_log"] = int(rng.integers(17, 20))          # windows of one to four blocks
        if rng.integers(0, 2): kw["hash_log"] = int(rng.integers(6, 19))
        if rng.integers(0, 2): kw["chain_log"] = int(rng.integers(6, 18))
        if rng.integers(0, 2): kw["min_match"] = int(rng.integers(3, 8))
        if rng.integers(0, 3) == 0: kw["target_length"] = int(rng.integers(0, 64))
        if rng.integers(0, 2): kw["strategy"] = int(rng.integers(1, 3))
        level = int(rng.choice([3, 3, 1, 2, -1, -5, 4]))
        flags = int(rng.choice([5, 5, 7, 6, 4, 1]))               # content size / checksum / dictionary-id flags
        raws = []
        for i in range(8):
            n = int(rng.choice([rng.integers(1, 2000), rng.integers(2000, 20000), rng.integers(20000, 131073), rng.integers(131073, 280000)]))
            if focus and rng.integers(0, 2): n = int(rng.integers(1, 6)) * 131072 + int(rng.choice([-1, 0, 1, 2, 4097, 70000]))    # around whole blocks
            k = int(rng.integers(0, 4))
            if k == 0: r = (pool[int(rng.integers(0, 12))] + pool[int(rng.integers(0, 12))] + pool[int(rng.integers(0, 12))])[:n]
            elif k == 1: r = rng.bytes(n)
            elif k == 2:
                blk = rng.bytes(int(rng.integers(20, 900))); r = ((blk + rng.bytes(int(rng.integers(1, 3000)))) * (n // 20 + 1))[:n]
            else: r = bytes(rng.integers(0, 5, n, dtype=np.uint8))
            raws.append(r)
        want = []
        for r in raws:
            try: want.append(ref.compress_advanced(r, level=level, flags=flags, **kw))
            except RuntimeError: want.append(None)
        emu.set_cparams(**kw)
        for pipe in (True, False):
            outs, st = emu.compress_batch(raws, level=level, flags=flags, n_blocks=2, pipeline=pipe)
            for i, (o, w) in enumerate(zip(outs, want)):
                if st[i] == 40:
                    refused += 1
                    if os.environ.get('SHOW_REFUSED') and pipe: print('refused', level, kw, len(raws[i]))
                elif st[i] != 0 or w is None or o != w:
                    bad += 1; print("MISMATCH", seed, it, level, kw, pipe, i, len(raws[i]), st[i], len(o), None if w is None else len(w))
                else: ok += 1
finally:
    emu.set_cparams()
print("params stress", seed, "ok", ok, "refused", refused, "bad", bad, "searched by the flat kernel over several blocks", emu.stat(8), "redone", emu.stat(9), "%.1fs" % (time.time() - t0))
sys.exit(1 if bad else 0)
