"""Sequence statistics of the bench corpus (analysis aid for the decode kernels' design; test infrastructure).
usage: python tests/tools/seq_stats.py [frames] [mix]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.corpus import Corpus
from tests import reflib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mix = sys.argv[2] if len(sys.argv) > 2 else "silesia"
corpus = Corpus(mix=mix)
ref = reflib.RefZstd() if reflib.have_ref() else reflib.Oracle()
L = C.CDLL(reflib.ORACLE_SO)
L.zo_set_seq_trace.argtypes = [C.c_void_p, C.c_size_t]
L.zo_seq_trace_count.restype = C.c_size_t
L.zo_decompress_frame.restype = C.c_int64
L.zo_decompress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
raws = corpus.frame_list(0, n)
buf = np.zeros((1 << 18, 3), dtype=np.uint32)
out = C.create_string_buffer(131072)
allseq = []
csz = 0
for r in raws:
    f = ref.compress(r)
    csz += len(f)
    L.zo_set_seq_trace(buf.ctypes.data, buf.shape[0])
    got = L.zo_decompress_frame(out, 131072, f, len(f), None, 0, None)
    assert got == len(r)
    k = L.zo_seq_trace_count()
    allseq.append(buf[:k].copy())
L.zo_set_seq_trace(None, 0)
ns = np.array([len(a) for a in allseq])
print("frames", n, "ratio %.2f" % (n * 131072 / csz), "nbSeq mean %.0f  p10 %d p50 %d p90 %d max %d" % (ns.mean(), *np.percentile(ns, [10, 50, 90]).astype(int), ns.max()))
S = np.concatenate([a for a in allseq if len(a)])
ll, ml, of = S[:, 0].astype(np.int64), S[:, 1].astype(np.int64), S[:, 2].astype(np.int64)
def hist(name, v, edges):
    h = np.histogram(v, bins=edges)[0] / len(v)
    print(name, " ".join("%s:%.3f" % (("<%d" % e), x) for e, x in zip(edges[1:], h)))
print("mean ll %.2f ml %.2f  bytes/seq %.2f  literal share %.3f" % (ll.mean(), ml.mean(), (ll + ml).mean(), ll.sum() / (ll + ml).sum()))
hist("ll", ll, [0, 1, 2, 4, 8, 16, 32, 64, 1 << 20])
hist("ml", ml, [0, 4, 5, 8, 16, 32, 64, 128, 1 << 20])
hist("of", of, [0, 4, 16, 64, 256, 1024, 4096, 16384, 65536, 1 << 20])
print("overlapping (of < ml): %.4f" % (of < ml).mean())
# distance of the match source behind the current 64-sequence batch start
for B in (64,):
    near = 0; tot = 0; pre = 0
    depth_hist = np.zeros(66)
    for a in allseq:
        if not len(a): continue
        l, m, o = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].astype(np.int64)
        end = np.cumsum(l + m); mstart = end - m
        for b0 in range(0, len(a), B):
            sl = slice(b0, min(len(a), b0 + B))
            base = end[b0 - 1] if b0 else 0
            src = mstart[sl] - o[sl]
            inb = src + m[sl] > base
            near += inb.sum(); tot += inb.size
            pre += ((src < base) & inb).sum()
    print("batch %d: near (reads the batch's own output) %.3f, of which straddle the batch start %.3f" % (B, near / tot, pre / max(near, 1)))
