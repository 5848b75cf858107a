"""Pins libzstd 1.5.7's answers (oracle/_ref, built from /root/reference/zstd/zstd.c) to the hand-made frames of tests/craft.py:
python tests/golden/make_edge_frames.py  ->  tests/golden/edge_frames.json (frame bytes in hex, accepted or not, sha256 of the output)."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import craft, reflib


def main():
    ref = reflib.RefZstd()
    rows = []
    for name, f, n, ok in craft.edge_frames() + craft.skippable_frames() + craft.encoding_variants():
        try: out = ref.decompress(f, n)
        except RuntimeError: out = None
        assert (out is not None) == ok, name
        rows.append(dict(name=name, frame=f.hex() if len(f) < 4096 else None, frame_sha256=hashlib.sha256(f).hexdigest(), size=n, accepted=ok,
                         out_sha256=hashlib.sha256(out).hexdigest() if ok else None))
    json.dump(dict(libzstd=int(ref.lib.ZSTD_versionNumber()), frames=rows), open(os.path.join(HERE, "edge_frames.json"), "w"), indent=1)
    print(len(rows), "frames")


if __name__ == "__main__":
    main()
