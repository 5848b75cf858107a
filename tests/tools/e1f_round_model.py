"""What a WAVE of the flat match kernel pays per trip, counted under the emulator (round 6; analysis aid, test infrastructure).

The flat double-fast search is one lane per source, 64 sources per wave; lanes re-converge at the top of every trip, so a trip costs the
wave, for every KIND of dependent memory round, the most any of its lanes needed -- and some lane of 64 always has a match. The search
functions log their rounds (ZE_RND, ZHIP_EMU builds only); this tool runs whole waves of bench-corpus sources through both forms -- the
form of rounds 1-5 that re-reads a lane's own bytes from memory (ze_dfast_flat_np) and round 6's LDS-window form (ze_dfast_flat_w) --
checks every frame against libzstd and prints the per-wave round totals: the length of the chain that bounds launches of <= 32 768
sources (every host-API call).   usage: python tests/tools/e1f_round_model.py [waves] [probes ...]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.corpus import Corpus
from tests import reflib, emulib

KINDS = ["own bytes from memory (round 0)", "table cells (round 1)", "candidates (round 2)", "count (round 3)", "count goes on", "catch-up goes on",
         "own / insertion bytes not in the window", "repeat-offset bytes", "trips that only fetched"]
waves = int(sys.argv[1]) if len(sys.argv) > 1 else 1
probes = [int(x) for x in sys.argv[2:]] or [2, 3, 4]
corpus = Corpus(mix="silesia")
ref = reflib.checker()
emu = emulib.Emu()
raws = corpus.frame_list(0, 64 * waves)
want = [ref.compress(r) for r in raws]
out = (C.c_uint64 * 20)()
for np_ in probes:
    emu.lib.emu_set_probes(C.c_uint32(np_))
    res = {}
    for win in (0, 1):
        emu.lib.emu_set_e1f_window(C.c_uint32(win))
        emu.lib.emu_rnd_log(1)
        got, st = emu.compress_batch(raws, level=3, flags=1, n_blocks=4, pipeline=True)
        emu.lib.emu_rnd_report(out)
        emu.lib.emu_rnd_log(0)
        assert not any(st) and got == want, "frames differ from libzstd (window=%d, probes=%d)" % (win, np_)
        res[win] = list(out)
    print("== %d probes per trip, %d wave(s) of 64 bench sources, frames identical to libzstd in both forms" % (np_, waves))
    print("%-44s %12s %12s   %s" % ("rounds the waves paid", "rounds 1-5", "LDS window", "(lane-trips that needed it: before / after)"))
    for k, name in enumerate(KINDS):
        print("%-44s %12d %12d   %d / %d" % (name, res[0][k], res[1][k], res[0][10 + k], res[1][10 + k]))
    t0, t1 = sum(res[0][:8]), sum(res[1][:8])
    print("%-44s %12d %12d   %.1f %%" % ("all dependent rounds", t0, t1, 100.0 * (t1 - t0) / t0))
    print("%-44s %12d %12d   lane-trips %d / %d" % ("trips (slowest lane of each wave)", res[0][9], res[1][9], res[0][19], res[1][19]))
emu.lib.emu_set_e1f_window(C.c_uint32(1))
