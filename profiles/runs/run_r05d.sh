#!/bin/bash
# round 5, GPU session D: K3 is bound by the line fetches that leave the L2 (r05b / r05c): does keeping the streamed-once data (sequences, decoded
# literals) out of the L2's way help? -DZP_K3_NT=1 reads them with non-temporal loads (r03b measured 28.4 against 29.05 ms on round 3's kernel and
# left it off); =3 also the far-match sources; with five / four waves per SIMD (fewer frames in flight: more L2 per frame). One set of frames,
# one process per build (tests/tools/decode_variants_ab.py), two rounds.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05d && O=gpurun_out/r05d
export TMPDIR=/tmp
timeout 900 python tests/tools/decode_variants_ab.py --steps 5 --rounds 2 product nt1 nt3 k3w5 nt1w5 k3w4 2>&1 | grep -v amdgpu.ids | tee $O/k3_nt_loads.txt
