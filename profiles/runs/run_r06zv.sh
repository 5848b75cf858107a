#!/bin/bash
# round 6, GPU session ZV: the compress step by level -- the fast strategy (levels 1, 2, negative) takes the lane-serial match kernel, which no bench line times
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zv && O=gpurun_out/r06zv
export TMPDIR=/tmp
timeout 1200 python tests/tools/compress_levels_rate.py 16384 2>&1 | tail -1 | tee $O/compress_levels.txt
