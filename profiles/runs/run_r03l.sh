#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03l && O=gpurun_out/r03l
export TMPDIR=/tmp
ZHIP_PROF=1 timeout 300 python bench.py --config dict --steps 2 --warmup 1 --no-cpu-baseline > $O/d_prof.json 2> $O/d_prof.err
grep -h "zhip-prof" $O/d_prof.err | grep -v "0.00%" | tail -24
