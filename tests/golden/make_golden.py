"""Generates tests/golden/golden.json + frames_*.bin from the REFERENCE libzstd 1.5.7 (oracle/_ref/libzstd_ref.so, built
from /root/reference/zstd/zstd.c). Run in the authoring container only:  python tests/golden/make_golden.py

Inputs are deterministic (tests/corpus.py seeds or literals below), so only the reference's OUTPUT is stored:
small frames verbatim (hex), large ones as sha256 + size, plus a handful of full frames for decoder-only tests.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np

from tests.corpus import Corpus
from tests import reflib


def inputs(corpus):
    rng = np.random.default_rng(20260924)
    cases = {
        "empty": b"", "foo": b"foo", "foo4": b"foo" * 4, "bar6": b"bar" * 6, "x6": b"x" * 6, "x7": b"x" * 7, "x8": b"x" * 8,
        "a1000": b"a" * 1000, "a131072": b"a" * 131072, "bytes256x40": bytes(range(256)) * 40,
        "hello500": b"hello world, hello world, hello there world! " * 500,
        "random128k": rng.bytes(1 << 17), "random300": rng.bytes(300),
        "quat70000": bytes(rng.integers(0, 4, 70000, dtype=np.uint8)),
    }
    for i in range(16):
        cases["corpus%d" % i] = corpus.frame_bytes(i)
    for i in range(16):
        cases["corpus%d_cut" % (100 + i)] = corpus.frame_bytes(100 + i)[: 997 * (i + 1)]
    return cases


DICT_SIZES = (0, 6, 7, 63, 64, 65, 300, 1024, 1025, 2048, 5000, 9999, 16383, 16384)


def dict_inputs(corpus):
    """(raw-content dictionary, list of sources) used by the dictionary-compression vectors"""
    raw_dict = b"".join(corpus.frame_bytes(600 + i)[:4096] for i in range(12))
    base = [b"".join(corpus.frame_bytes(640 + 4 * j + k)[:4096] for k in range(4)) + corpus.frame_bytes(40 + j)[:2000] for j in range(len(DICT_SIZES))]
    return raw_dict, [b[:n] for b, n in zip(base, DICT_SIZES)]


def dict_compress_cases(ref, corpus, trained):
    raw_dict, srcs = dict_inputs(corpus)
    out = {"sizes": list(DICT_SIZES), "raw_dict_sha256": hashlib.sha256(raw_dict).hexdigest(), "frames": {}}
    for name, dd in (("trained", trained), ("raw", raw_dict)):
        for tag, flags in (("default", reflib.DEFAULT_FLAGS), ("checksum_nodictid", reflib.F_CONTENTSIZE | reflib.F_CHECKSUM)):
            recs = []
            for s in srcs:
                fr = ref.compress(s, level=3, flags=flags, dict_data=dd)
                recs.append({"size": len(fr), "sha256": hashlib.sha256(fr).hexdigest()})
            out["frames"]["%s/%s" % (name, tag)] = recs
    return out


LEVELS = (1, 2, -1, -5, -100)
MB_LEVELS = (3, 1, -3)


def multiblock_inputs(corpus):
    def cb(i, n):
        out = b""
        k = 0
        while len(out) < n:
            out += corpus.frame_bytes(2000 + i * 16 + k)
            k += 1
        return out[:n]
    rng = np.random.default_rng(1)
    return {"random1M": rng.bytes(1 << 20), "zeros600k": bytes(600000), "corpus128k+1": cb(0, 131073), "corpus300k": cb(1, 300000),
            "corpus1M": cb(2, 1 << 20), "mixed": cb(3, 200000) + np.random.default_rng(3).bytes(200000) + cb(4, 150000),
            "rle_tail": cb(5, 131072) + b"x" * 131072 + b"y" * 3, "beyond_window_l1": cb(6, 700000)}


def multiblock_cases(ref, corpus):
    out = {"levels": list(MB_LEVELS), "frames": {}}
    for name, data in multiblock_inputs(corpus).items():
        rec = {"size": len(data), "input_sha256": hashlib.sha256(data).hexdigest()}
        for lvl in MB_LEVELS:
            for tag, flags in (("default", reflib.DEFAULT_FLAGS), ("checksum", reflib.DEFAULT_FLAGS | reflib.F_CHECKSUM)):
                fr = ref.compress(data, level=lvl, flags=flags)
                rec["%d/%s" % (lvl, tag)] = {"size": len(fr), "sha256": hashlib.sha256(fr).hexdigest()}
        out["frames"][name] = rec
    return out


def level_cases(ref, cases):
    out = {"levels": list(LEVELS), "frames": {}}
    for name, data in cases.items():
        out["frames"][name] = {}
        for lvl in LEVELS:
            fr = ref.compress(data, level=lvl, flags=reflib.DEFAULT_FLAGS)
            out["frames"][name][str(lvl)] = {"size": len(fr), "sha256": hashlib.sha256(fr).hexdigest()}
    return out


def main():
    ref = reflib.RefZstd()
    corpus = Corpus()
    out = {"libzstd": "1.5.7", "source": "/root/reference/zstd/zstd.c via oracle/Makefile", "level": 3, "cases": []}
    blob = bytearray()
    for name, data in inputs(corpus).items():
        entry = {"name": name, "size": len(data), "input_sha256": hashlib.sha256(data).hexdigest(), "frames": {}}
        for tag, flags in (("default", reflib.DEFAULT_FLAGS), ("checksum", reflib.DEFAULT_FLAGS | reflib.F_CHECKSUM),
                           ("nosize", reflib.F_DICTID)):
            frame = ref.compress(data, level=3, flags=flags)
            rec = {"size": len(frame), "sha256": hashlib.sha256(frame).hexdigest()}
            if len(frame) <= 64:
                rec["hex"] = frame.hex()
            if tag == "default" and (name in ("corpus0", "corpus1", "corpus2", "corpus3", "corpus4", "corpus5", "hello500", "quat70000")
                                     or name.endswith("_cut")):
                rec["blob_offset"] = len(blob)
                blob += frame
            entry["frames"][tag] = rec
        out["cases"].append(entry)
    # a level-19 multi-block frame and a dictionary frame for decoder coverage
    big = b"".join(corpus.frame_bytes(300 + i) for i in range(3))
    f19 = ref.compress(big, level=19)
    out["multiblock_level19"] = {"input": "corpus frames 300..302 concatenated", "size": len(big), "blob_offset": len(blob),
                                 "frame_size": len(f19), "input_sha256": hashlib.sha256(big).hexdigest()}
    blob += f19
    samples = [corpus.frame_bytes(40 + i)[: 2000 + 37 * i] for i in range(200)]
    d = ref.train_dictionary(16384, samples)
    dict_frames = [ref.compress(s, dict_data=d) for s in samples[:8]]
    out["dictionary"] = {"dict_blob_offset": len(blob), "dict_size": len(d), "samples": "corpus frames 40..47 cut to 2000+37*i",
                         "frames": []}
    blob += d
    for s, fr in zip(samples[:8], dict_frames):
        out["dictionary"]["frames"].append({"blob_offset": len(blob), "size": len(fr), "input_sha256": hashlib.sha256(s).hexdigest(),
                                            "input_size": len(s)})
        blob += fr
    # frames of several blocks (sources above 128 KiB): SURVEY config 1 (1 MiB of random bytes -> 1 048 609 bytes) and compressible ones
    out["multiblock_compress"] = multiblock_cases(ref, corpus)
    # other strategies of the same path: the `fast` parser (levels 1, 2 and negative levels)
    out["levels"] = level_cases(ref, inputs(corpus))
    # dictionary COMPRESSION (attached-dictionary mode, sources <= 16 KiB): trained and raw-content dictionaries
    out["dictionary_compress"] = dict_compress_cases(ref, corpus, d)
    with open(os.path.join(HERE, "golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    with open(os.path.join(HERE, "frames.bin"), "wb") as fh:
        fh.write(blob)
    print("cases", len(out["cases"]), "blob bytes", len(blob))


if __name__ == "__main__":
    main()
