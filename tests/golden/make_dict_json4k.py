"""Regenerates tests/golden/dict_json4k.bin: the trained dictionary of BASELINE.json configs[3] ("shared ZstdCompressionDict,
train_dictionary on 10 k JSON samples") at the size SURVEY.md 8(d)4 / BASELINE.md section 3 name -- train_dictionary(112640, samples),
the reference's default dictionary size (c-ext/compressiondict.c:56-61) -- and dict_json4k_16k.bin, the 16 KiB dictionary of rounds 1-2
that the emulator tests keep using (a 110 KiB dictionary's 384 KiB of tagged tables take minutes per build under emulation).
Test / bench infrastructure.

Run in the authoring container, where the reference build (oracle/_ref/libzstd_ref.so, compiled from /root/reference/zstd/zstd.c by
oracle/Makefile) exists:   python tests/golden/make_dict_json4k.py
What train_dictionary(dict_size, samples) does with its defaults (c-ext/compressiondict.c:13-146): d = 8, steps = 4, level = 3 into
ZDICT_optimizeTrainFromBuffer_fastCover -- which is exactly ZDICT_trainFromBuffer (zstd.c, zdict section), called here through ctypes.
Samples: JSON-like documents 1 000 000 .. 1 009 999 of tests/corpus.py (4 KiB each); the bench compresses documents 0 .. 262 143.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests import reflib                      # noqa: E402
from tests.corpus import Corpus               # noqa: E402

DICT_SIZES = {"dict_json4k": 112640, "dict_json4k_16k": 16384}
N_SAMPLES = 10000
SAMPLE_START = 1000000
DOC = 4096


def main():
    ref = reflib.RefZstd()
    docs = Corpus(frame_size=DOC).json_docs(SAMPLE_START, N_SAMPLES).numpy()
    samples = [docs[i].tobytes() for i in range(N_SAMPLES)]
    for name, size in DICT_SIZES.items():
        make(ref, samples, name, size)


def make(ref, samples, name, DICT_SIZE):
    d = ref.train_dictionary(DICT_SIZE, samples)
    with open(os.path.join(HERE, name + ".bin"), "wb") as f:
        f.write(d)
    # a few pinned vectors: frames of documents 0..7 with this dictionary at level 3 (what bench.py --config dict must reproduce)
    probe = Corpus(frame_size=DOC).json_docs(0, 8).numpy()
    frames = [ref.compress(probe[i].tobytes(), level=3, dict_data=d) for i in range(8)]
    meta = {"dict_size": len(d), "dict_sha256": hashlib.sha256(d).hexdigest(), "samples": "json_docs %d..%d, %d bytes each" % (SAMPLE_START, SAMPLE_START + N_SAMPLES - 1, DOC),
            "trainer": "ZDICT_trainFromBuffer (libzstd 1.5.7, reference build) == train_dictionary(%d, samples) defaults" % DICT_SIZE,
            "probe_frames_sha256": [hashlib.sha256(f).hexdigest() for f in frames], "probe_frame_sizes": [len(f) for f in frames],
            "probe_docs_sha256": [hashlib.sha256(probe[i].tobytes()).hexdigest() for i in range(8)]}
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(meta)


if __name__ == "__main__":
    main()
