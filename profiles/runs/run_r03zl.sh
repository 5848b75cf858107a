#!/bin/bash
# round 3, session ZL: GPU suite with the encoding-variant frames
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03zl && O=gpurun_out/r03zl
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -6 $O/pytest_gpu.txt
