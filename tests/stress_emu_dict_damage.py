"""Emulator fuzz of dictionary digestion on DAMAGED dictionaries: python tests/stress_emu_dict_damage.py SEED [N].
A trained dictionary with a bit flipped / bytes overwritten / cut in its entropy header goes through the device-side compression and
decompression dictionary digestion (ze_cdict_body, zhip_parse_dict) and libzstd 1.5.7 (ZSTD_loadZstdDictionary zstd.c:28115,
ZSTD_loadDEntropy :44673): both refuse or both accept, and frames made with an accepted one are identical and round-trip.
Run under the AddressSanitizer build for the bounds (tests/emu/build_asan.sh). Not collected by pytest."""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from tests import emulib, reflib
from tests.corpus import Corpus
emu = emulib.Emu(); ref = reflib.RefZstd(); corpus = Corpus()
seed = int(sys.argv[1]); rng = np.random.default_rng(seed)
fr = corpus.frame_list(900, 24)
trained = ref.train_dictionary(8192, [f[j*4096:(j+1)*4096] for f in fr for j in range(16)])
# entropy header length: find content offset via our own parser
_, content, _ = emu.parse_dict(trained)
hdr = len(trained) - len(content)
raws = [fr[i][j*4096:(j+1)*4096][:int(rng.integers(100, 4097))] for i in range(2) for j in range(3)]
res = dict(n=0, both_err=0, both_ok=0, ours_only_err=0, theirs_only_err=0, diff=0, dd_mismatch=0)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    b = bytearray(trained)
    kind = int(rng.choice([0,0,0,0,0,1,2,3]))
    if kind == 0:
        for _ in range(1): b[int(rng.integers(8, hdr + 4))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        at = int(rng.integers(8, hdr)); b[at:at+int(rng.integers(1, 12))] = rng.bytes(int(rng.integers(1, 12)))
    elif kind == 2:
        del b[int(rng.integers(8, hdr + 40)):]                  # truncated in / just after the header
    else:
        at = int(rng.integers(8, hdr)); del b[at:at+int(rng.integers(1, 6))]
    mut = bytes(b); res["n"] += 1
    try:
        want = [ref.compress(r, level=3, dict_data=mut) for r in raws]
    except (RuntimeError, AssertionError):
        want = None
    try:
        outs, st = emu.compress_batch(raws, level=3, flags=5, dict_data=mut, pipeline=bool(it & 1))
        ours = outs if not any(st) else None
        if any(st): print("per-frame status", st)
    except RuntimeError as e:
        ours = None
    if want is None and ours is None: res["both_err"] += 1
    elif want is None: res["ours_only_err".replace("ours_only_err", "theirs_only_err")] += 1; print(seed, it, "libzstd refuses, we accept", kind)
    elif ours is None: res["ours_only_err"] += 1; print(seed, it, "we refuse, libzstd accepts", kind)
    elif ours != want: res["diff"] += 1; print(seed, it, "frames differ", kind)
    else: res["both_ok"] += 1
    # decompression dictionary digestion
    st = emu.set_ddict(mut)
    try:
        ref.decompress(ref.compress(raws[0], level=3), len(raws[0]), dict_data=mut); their_dd = 0
    except RuntimeError: their_dd = 1
    # libzstd's ZSTD_DCtx_loadDictionary failure is silent in reflib.decompress (return value ignored) -- check directly
    import ctypes as C
    d = ref.lib.ZSTD_createDCtx(); r = ref.lib.ZSTD_DCtx_loadDictionary(d, mut, len(mut)); their_dd = 1 if ref.lib.ZSTD_isError(r) else 0; ref.lib.ZSTD_freeDCtx(d)
    if (st != 0) != (their_dd != 0): res["dd_mismatch"] += 1; print(seed, it, "ddict digestion differs: ours", st, "theirs", their_dd, kind)
    elif st == 0 and want is not None:
        outs, st2, nfb = emu.decompress_pipeline(want, [len(r) for r in raws], n_blocks=2)
        if any(st2) or outs != raws: res["dd_mismatch"] += 1; print(seed, it, "round trip with damaged dict failed", st2)
emu.set_ddict(None)
print("dict fuzz", seed, res)
