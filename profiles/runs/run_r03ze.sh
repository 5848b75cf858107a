#!/bin/bash
# round 3, session ZE: GPU stress of the several-block modes through the Python API (tests/stress_gpu_blocks.py), then the reference's own hot-path tests on the final build
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03ze && O=gpurun_out/r03ze
export TMPDIR=/tmp
for s in 1 2 3 4 5 6 7 8; do timeout 300 python tests/stress_gpu_blocks.py $s 2>&1 | tail -1; done | tee $O/stress_gpu_blocks.txt
sh tests/run_reference_hotpath_tests.sh > $O/reference_tests_tail.txt 2>&1; tail -2 $O/reference_tests_tail.txt
