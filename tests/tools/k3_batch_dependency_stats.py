"""How K3's 64-sequence batches depend on themselves (analysis aid for DESIGN.md 4.1, round 4; test infrastructure).
For `frames` frames of the bench corpus: per batch of B sequences, the share of matches that read the batch's own output ("near"), the
dependency depth (rounds a need-mask scheme takes), the rounds of a conservative frontier rule, and the long items above 16 / 32 bytes.
usage: python tests/tools/k3_batch_dependency_stats.py [frames=48]"""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.corpus import Corpus
from tests import reflib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ref = reflib.checker()
L = C.CDLL(reflib.ORACLE_SO)
L.zo_set_seq_trace.argtypes = [C.c_void_p, C.c_size_t]
L.zo_seq_trace_count.restype = C.c_size_t
L.zo_decompress_frame.restype = C.c_int64
L.zo_decompress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
buf = np.zeros((1 << 18, 3), dtype=np.uint32)
out = C.create_string_buffer(131072)
allseq = []
for r in Corpus(mix="silesia").frame_list(0, n):
    f = ref.compress(r)
    L.zo_set_seq_trace(buf.ctypes.data, buf.shape[0])
    assert L.zo_decompress_frame(out, 131072, f, len(f), None, 0, None) == len(r)
    allseq.append(buf[: L.zo_seq_trace_count()].astype(np.int64).copy())
L.zo_set_seq_trace(None, 0)
for B in (64, 32, 16):
    st = collections.Counter(); dh = collections.Counter()
    for a in allseq:
        if not len(a):
            continue
        ll, ml, of = a[:, 0], a[:, 1], a[:, 2]
        end = np.cumsum(ll + ml); mst = end - ml
        for b0 in range(0, len(a), B):
            sl = slice(b0, min(len(a), b0 + B))
            base = end[b0 - 1] if b0 else 0
            m0 = mst[sl] - base; m1 = end[sl] - base
            s0 = m0 - of[sl]; s1 = np.minimum(s0 + ml[sl], m0)
            near = s1 > 0
            nn = len(m0)
            depth = np.zeros(nn, dtype=int)
            for j in range(nn):
                if not near[j]:
                    continue
                a0 = max(s0[j], 0); d = 0
                for i in range(j):
                    if near[i] and m1[i] > a0 and m0[i] < s1[j]:
                        d = max(d, depth[i])
                depth[j] = d + 1
            pend = near.copy(); fr = 0
            while pend.any():                                       # frontier rule: ready when the source ends at or below the first pending match's start
                first = int(np.argmax(pend)); ready = pend & (s1 <= m0[first]); ready[first] = True
                pend &= ~ready; fr += 1
            st["batches"] += 1; st["seqs"] += nn; st["near"] += int(near.sum()); st["rounds"] += int(depth.max()) if nn else 0; st["frontier"] += fr
            st["long16"] += int((ll[sl] > 16).sum() + (~near & (ml[sl] > 16)).sum()); st["long32"] += int((ll[sl] > 32).sum() + (~near & (ml[sl] > 32)).sum())
            for d in depth[near]:
                dh[int(d)] += 1
    b = st["batches"]; tot = max(1, sum(dh.values()))
    print("B %2d: near %.3f of the sequences, need-mask rounds %.2f per batch (%.2f per 64 sequences), frontier rounds %.2f, long items > 16 B %.2f, > 32 B %.2f per batch; depth shares %s"
          % (B, st["near"] / st["seqs"], st["rounds"] / b, st["rounds"] / st["seqs"] * 64, st["frontier"] / b, st["long16"] / b, st["long32"] / b,
             " ".join("%d:%.2f" % (k, v / tot) for k, v in sorted(dh.items())[:6])))
