#!/bin/bash
# round 6, GPU session ZQ: K0 forced on for a small, mixed batch (ZHIP_K0_MIN=0: the suite's batches are below the 6 144 frames from which K0 runs) -- new GPU test -- and the decode tests again
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zq && O=gpurun_out/r06zq
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_decompress.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_decompress.txt
ZHIP_K0_MIN=0 timeout 1500 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py tests/test_cext_backend.py -x -q -m gpu 2>&1 | tail -3 | sed 's/^/ZHIP_K0_MIN=0 (K0 on every batch): /' | tee -a $O/pytest_decompress.txt
