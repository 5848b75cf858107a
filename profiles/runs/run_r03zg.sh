#!/bin/bash
# round 3, session ZG: a mixed batch (16 384 x 128 KiB + 16 x 4 MiB frames) through the host-buffer API: several-block mode with the pool sized from all sizes, against ZHIP_BLOCKS=0; GPU suite
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03zg && O=gpurun_out/r03zg
export TMPDIR=/tmp
timeout 600 python tests/mixed_batch_rate.py > $O/mixed.txt 2>&1; tail -1 $O/mixed.txt
ZHIP_BLOCKS=0 timeout 600 python tests/mixed_batch_rate.py > $O/mixed_generic.txt 2>&1; tail -1 $O/mixed_generic.txt
timeout 600 python tests/mixed_batch_rate.py 16384 0 > $O/uniform.txt 2>&1; tail -1 $O/uniform.txt
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
