"""Offline go / no-go for intra-frame parallel tANS sequence decoding (analysis aid, test infrastructure).
Compresses `frames` frames of the bench corpus with the reference libzstd, runs tests/tools/tans_sync_study.c over them and prints
the distribution of "sequences decoded from an arbitrary (bit position, states) start until the decoder is on the true trajectory".
usage: python tests/tools/tans_sync_study.py [frames=1000] [trials_per_block=15] [maxSteps=4096]"""
import os, struct, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.corpus import Corpus
from tests import reflib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 15
max_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "tans_sync_study")
subprocess.check_call(["gcc", "-O2", "-w", "-o", exe, os.path.join(ROOT, "tests", "tools", "tans_sync_study.c")])
ref = reflib.RefZstd() if reflib.have_ref() else reflib.Oracle()
corpus = Corpus(mix="silesia")
path = os.path.join(tmp, "frames.bin")
with open(path, "wb") as f:
    for s in range(0, n, 64):
        for r in corpus.frame_list(s, min(64, n - s)):
            c = ref.compress(r)
            f.write(struct.pack("<I", len(c))); f.write(c)
out = os.path.join(tmp, "out.txt")
subprocess.check_call([exe, path, out, str(trials), str(max_steps)])
a = np.loadtxt(out, dtype=np.int64)
steps, nbseq = a[:, 0], a[:, 2]
ok = steps >= 0
print("trials %d (frames %d, %d cut points per block, 2 start-state variants each), block nbSeq median %d" % (len(a), n, trials, np.median(nbseq)))
print("synchronised within %d steps: %.4f" % (max_steps, ok.mean()))
if ok.any():
    s = steps[ok]
    print("steps to synchronise (of those): median %d  p90 %d  p99 %d  p99.9 %d  max %d" % (np.median(s), *np.percentile(s, [90, 99, 99.9]).astype(int), s.max()))
for lim in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    print("  <= %4d steps: %.4f" % (lim, ((steps >= 0) & (steps <= lim)).mean()))
