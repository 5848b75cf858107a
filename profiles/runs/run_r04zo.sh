#!/bin/bash
# round 4, GPU session ZO: parity stress on the round's last commit -- single-block sources through the default kernels and through the flat kernel only
# (tests/stress_gpu_compress.py), sources of 1-9 blocks through the flat several-block search (tests/stress_gpu_blocks.py)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zo && O=gpurun_out/r04zo
export TMPDIR=/tmp
{ for s in 11 12 13; do timeout 300 python tests/stress_gpu_compress.py $s 2>&1 | grep -v amdgpu.ids | tail -3; done
  for s in 21 22 23 24 25 26; do timeout 300 python tests/stress_gpu_blocks.py $s 2>&1 | tail -1; done; } | tee $O/stress_gpu.txt
