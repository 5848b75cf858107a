#!/bin/bash
# round 3, GPU session P: batches of multi-block frames with a full grid of the generic kernels (size hint)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03p && O=gpurun_out/r03p
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 600 python tests/multiblock_rate.py 2048 1024 > $O/mb_1m.txt 2>&1; tail -1 $O/mb_1m.txt
timeout 600 python tests/multiblock_rate.py 8192 256 > $O/mb_256k.txt 2>&1; tail -1 $O/mb_256k.txt
timeout 300 python tests/small_batch_latency.py > $O/lat.txt 2>&1; tail -1 $O/lat.txt
