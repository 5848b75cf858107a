#!/bin/bash
# round 3, session Y: after doubling the block slots per frame -- several-block rates (chunks of fewer frames now), one-shot large frames, GPU suite
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03y && O=gpurun_out/r03y
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
timeout 400 python tests/multiblock_rate.py 2048 1024 > $O/multiblock_2048x1MiB.txt 2>&1; tail -1 $O/multiblock_2048x1MiB.txt
timeout 600 python tests/multiblock_rate.py 16384 256 > $O/multiblock_16384x256KiB.txt 2>&1; tail -1 $O/multiblock_16384x256KiB.txt
timeout 400 python tests/small_batch_latency.py > $O/small_batch_latency.txt 2>&1; tail -1 $O/small_batch_latency.txt
