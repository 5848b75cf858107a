#!/bin/bash
# round 6, GPU session ZJ: the tables' launch numbers across calls of one context (new GPU test), the compress-side suite again on the build that tracks the zeroed extent and
# invalidates after un-numbered use
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zj && O=gpurun_out/r06zj
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_compress.py tests/test_gpu_boundary.py tests/test_cext_backend.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_compress.txt
timeout 900 python tests/stress_gpu_compress.py 2>&1 | tail -1 | tee $O/stress_gpu_compress.txt
timeout 900 python tests/stress_gpu_blocks.py 2>&1 | tail -1 | tee -a $O/stress_gpu_compress.txt
