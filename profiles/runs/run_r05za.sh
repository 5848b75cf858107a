#!/bin/bash
# round 5, GPU session ZA (diagnostic): E2's sequence stream (59 % of its 30 ms) split into its four parts per round of 64 sequences -- the lanes' table
# constants, the three state chains, packing + ORing the bits into the LDS buffer, flushing whole bytes (-DZE_PROF_STREAM, ZHIP_PROF=1)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05za && O=gpurun_out/r05za
export TMPDIR=/tmp
ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_e2prof.so ZHIP_PROF=1 ZHIP_E1F_PICK=0 timeout 300 python bench.py --config compress --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep zhip-prof | tail -13 | tee $O/e2_stream_parts.txt
