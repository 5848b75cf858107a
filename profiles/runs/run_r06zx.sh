#!/bin/bash
# round 6, GPU session ZX: the new GPU test of fast-strategy batches above 32 768 sources, and the compress-side tests again
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zx && O=gpurun_out/r06zx
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_compress.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_compress.txt
