#!/bin/bash
# round 3, session ZF: a longer GPU campaign of tests/stress_gpu_blocks.py (96 sources per seed)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03zf && O=gpurun_out/r03zf
export TMPDIR=/tmp
for s in $(seq 9 44); do timeout 300 python tests/stress_gpu_blocks.py $s 96 2>&1 | tail -1; done | tee $O/stress_gpu_blocks.txt | tail -5
grep -c "compress mismatches 0 decompress mismatches 0 one-shot mismatches 0" $O/stress_gpu_blocks.txt
