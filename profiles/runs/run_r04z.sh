#!/bin/bash
# round 4, GPU session Z: E1f against the allocation of its tables, many trials in one process (tests/tools/e1f_alloc_trials.py)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04z && O=gpurun_out/r04z
export TMPDIR=/tmp
timeout 600 python tests/tools/e1f_alloc_trials.py 3 0,2,16,64,512 2>&1 | tee $O/e1f_alloc_trials.txt | tail -20
