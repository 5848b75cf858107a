"""Which of K3's batches a restricted fast path would take (analysis aid for round 5's hand-written batch body; test infrastructure).
Simulates zp_exec_block's batching exactly (<= 64 sequences, output + carried bytes <= 4 096, flush in whole 16-byte units) over frames of the
bench corpus and counts, per batch, the features that matter to a fast path: a "big" sequence, in-batch ("near") matches longer than 32
bytes, near matches that overlap their own output, 16-byte units of long items, dependency depth.
usage: python tests/tools/k3_fastpath_stats.py [frames=128]"""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.corpus import Corpus
from tests import reflib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ref = reflib.checker()
L = C.CDLL(reflib.ORACLE_SO)
L.zo_set_seq_trace.argtypes = [C.c_void_p, C.c_size_t]
L.zo_seq_trace_count.restype = C.c_size_t
L.zo_decompress_frame.restype = C.c_int64
L.zo_decompress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
buf = np.zeros((1 << 18, 3), dtype=np.uint32)
out = C.create_string_buffer(131072)
st = collections.Counter(); frames_with = collections.Counter(); uh = collections.Counter(); dh = collections.Counter(); nearh = collections.Counter()
for r in Corpus(mix="silesia").frame_list(0, n):
    f = ref.compress(r)
    L.zo_set_seq_trace(buf.ctypes.data, buf.shape[0])
    assert L.zo_decompress_frame(out, 131072, f, len(f), None, 0, None) == len(r)
    a = buf[: L.zo_seq_trace_count()].astype(np.int64).copy()
    ll, ml, of = a[:, 0], a[:, 1], a[:, 2]
    done, op, carry = 0, 0, 0
    flags = set()
    while done < len(a):
        sl = slice(done, min(len(a), done + 64))
        incT = np.cumsum(ll[sl] + ml[sl])
        cnt = int((incT + carry <= 4096).sum())
        st["batches"] += 1
        if cnt == 0:
            st["big"] += 1; flags.add("big")
            op += int(ll[done] + ml[done]); carry = 0; done += 1
            continue
        sl = slice(done, done + cnt)
        l, m, o = ll[sl], ml[sl], of[sl]
        tot = l + m
        ob = op - carry
        oRel = np.cumsum(tot) - tot + carry
        mRel = oRel + l
        sAbs = ob + mRel - o
        hasM = m > 0
        far = hasM & (sAbs + m <= ob)
        near = hasM & ~far
        pre = near & (sAbs < ob)
        preLen = np.where(pre, ob - sAbs, 0)
        nLen = m - preLen
        st["seqs"] += cnt; st["near"] += int(near.sum()); nearh[min(int(near.sum()), 40)] += 1
        longnear = near & (m > 32)
        overlap = near & (o < nLen)
        lenMi = np.where(far, m, preLen)
        uL = np.where(l > 16, (l + 15) >> 4, 0); uM = np.where((far | pre) & (lenMi > 16), (lenMi + 15) >> 4, 0)
        U = int(uL.sum() + uM.sum())
        uh[min(U, 80)] += 1
        declined = False
        if longnear.any(): st["b_longnear"] += 1; declined = True
        if (overlap & ~longnear).any(): st["b_overlap_short"] += 1; declined = True
        if U > 64: st["b_U>64"] += 1; declined = True
        if declined: st["declined"] += 1
        # dependency depth
        m0 = mRel; m1 = mRel + m; s0 = sAbs - ob; s1 = np.minimum(s0 + m, m0)
        depth = np.zeros(cnt, dtype=int)
        for j in range(cnt):
            if not near[j]:
                continue
            a0 = max(s0[j], 0); d = 0
            for i in range(j):
                if near[i] and m1[i] > a0 and m0[i] < s1[j]:
                    d = max(d, depth[i])
            depth[j] = d + 1
        dh[int(depth.max())] += 1; st["rounds"] += int(depth.max())
        totB = int(tot.sum()) + carry
        op += int(tot.sum()); carry = totB & 15; done += cnt
        st["bytes"] += int(tot.sum())
    for k in flags: frames_with[k] += 1
L.zo_set_seq_trace(None, 0)
B = st["batches"]
print("frames %d  batches %d (%.1f per frame)  sequences per batch %.1f  bytes per batch %.0f" % (n, B, B / n, st["seqs"] / (B - st["big"]), st["bytes"] / (B - st["big"])))
print("big-sequence batches %d (%.2f %%), frames with one: %d" % (st["big"], 100 * st["big"] / B, frames_with["big"]))
print("near matches per batch %.1f, rounds per batch %.2f" % (st["near"] / B, st["rounds"] / B))
for k in ("b_longnear", "b_overlap_short", "b_U>64", "declined"):
    print("%-18s %6d batches (%.2f %%)" % (k, st[k], 100 * st[k] / B))
print("units per batch:", " ".join("%d:%.1f%%" % (k, 100 * v / B) for k, v in sorted(uh.items()) if v / B > 0.01))
print("depth:", " ".join("%d:%.1f%%" % (k, 100 * v / B) for k, v in sorted(dh.items())))
