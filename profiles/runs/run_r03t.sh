#!/bin/bash
# round 3, session T: GPU suite after the zero-output fix of the test
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03t && O=gpurun_out/r03t
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -5 $O/pytest_gpu.txt
