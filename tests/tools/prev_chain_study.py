"""Offline evidence for a table-free double-fast search (analysis aid, test infrastructure).
Runs tests/tools/prev_chain_study.c over `frames` sources of the bench corpus and prints what the link-following formulation would
cost per probed position, next to what the table formulation costs (two random reads + two random write-backs).
usage: python tests/tools/prev_chain_study.py [frames=512] [mix=silesia]"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.corpus import Corpus, CLASS_NAMES

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mix = sys.argv[2] if len(sys.argv) > 2 else "silesia"
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "prev_chain_study")
subprocess.check_call(["gcc", "-O2", "-w", "-o", exe, os.path.join(ROOT, "tests", "tools", "prev_chain_study.c"), os.path.join(ROOT, "oracle", "zo_decode.c")])
corpus = Corpus(mix=mix)
path = os.path.join(tmp, "sources.bin")
with open(path, "wb") as f:
    for s in range(0, n, 64):
        for r in corpus.frame_list(s, min(64, n - s)):
            f.write(r)
out = subprocess.run([exe, path, "131072"], check=True, capture_output=True, text=True)
sys.stderr.write(out.stderr)
a = np.array([[int(x) for x in l.split()] for l in out.stdout.splitlines()], dtype=np.int64)
probes, lookL, hopL, maxL, lookS, hopS, maxS, emptyL, emptyS, nseq, excess2, over16, over64 = a.T
print("frames %d (mix %s): probed positions per frame median %d, sequences median %d" % (len(a), mix, np.median(probes), np.median(nseq)))
print("long  table: %.2f lookups per probed position, %.2f random link reads per lookup (frame median %.2f, p90 %.2f, worst frame %.2f; longest walk %d); %.1f %% of lookups find the cell empty"
      % (lookL.sum() / probes.sum(), hopL.sum() / lookL.sum(), np.median(hopL / lookL), np.percentile(hopL / lookL, 90), (hopL / lookL).max(), maxL.max(), 100.0 * emptyL.sum() / lookL.sum()))
print("short table: %.2f lookups per probed position, %.2f random link reads per lookup (frame median %.2f, p90 %.2f, worst frame %.2f; longest walk %d); %.1f %% of lookups find the cell empty"
      % (lookS.sum() / probes.sum(), hopS.sum() / lookS.sum(), np.median(hopS / lookS), np.percentile(hopS / lookS, 90), (hopS / lookS).max(), maxS.max(), 100.0 * emptyS.sum() / lookS.sum()))
per = (hopL + hopS) / probes
print("random reads per probed position, links formulation: mean %.2f (frame median %.2f, p90 %.2f, p99 %.2f, max %.2f)  |  table formulation: 2 reads + 2 write-backs"
      % ((hopL.sum() + hopS.sum()) / probes.sum(), np.median(per), np.percentile(per, 90), np.percentile(per, 99), per.max()))

cls = [CLASS_NAMES[int(c)] for c in corpus.classes(__import__("torch").arange(len(a)))]
print("link reads beyond a lookup's second (a lane-per-frame kernel's wave waits for them), per frame: median %d, p90 %d, p99 %d, max %d; sum over frames %d = %.2f per probed position"
      % (np.median(excess2), np.percentile(excess2, 90), np.percentile(excess2, 99), excess2.max(), excess2.sum(), excess2.sum() / probes.sum()))
order = np.argsort(-excess2)
print("frames by excess (index: excess, probes, lookups over 16 / 64 reads, longest):")
for i in order[:12]:
    print("   frame %5d: %8d  probes %6d  over16 %5d over64 %5d  longest %d  class %s" % (i, excess2[i], probes[i], over16[i], over64[i], max(maxL[i], maxS[i]), cls[i]))
for cap in (16, 64, 256):
    bad = np.maximum(maxL, maxS) > cap
    print("hop cap %3d: %.1f %% of the frames have a lookup beyond it" % (cap, 100.0 * bad.mean()))
