#!/bin/bash
# round 3, session ZJ: tests/stress_gpu_blocks.py with its dictionary part (several-block frames against trained / raw-content dictionaries, both directions)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03zj && O=gpurun_out/r03zj
export TMPDIR=/tmp
for s in 101 102 103 104 105 106 107 108; do timeout 300 python tests/stress_gpu_blocks.py $s 32 2>&1 | tail -1; done | tee $O/stress_gpu_blocks_dict.txt
