#!/bin/bash
# round 3, session ZB: the LDS-source match kernel with the whole wave searching a source (ze_dfast_wave) against one lane (ZHIP_E1_WAVE=0): small-batch latencies, GPU suite
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03zb && O=gpurun_out/r03zb
export TMPDIR=/tmp
timeout 400 python tests/small_batch_latency.py > $O/small_batch_latency_wave.txt 2>&1; tail -1 $O/small_batch_latency_wave.txt
ZHIP_E1_WAVE=0 timeout 400 python tests/small_batch_latency.py > $O/small_batch_latency_lane.txt 2>&1; tail -1 $O/small_batch_latency_lane.txt
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
