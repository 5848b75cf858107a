#!/bin/bash
# round 3, session W: GPU suite with the compress several-block test; 8 192 / 16 384 x 256 KiB rates at the default threshold
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03w && O=gpurun_out/r03w
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -5 $O/pytest_gpu.txt
for cfg in "8192 256" "16384 256" "8192 1024"; do
  set -- $cfg
  timeout 900 python tests/multiblock_rate.py $1 $2 > $O/mb_${1}x${2}.txt 2>&1; tail -1 $O/mb_${1}x${2}.txt | cut -c1-120; python - <<P
import json
try:
    d=json.loads(open('$O/mb_${1}x${2}.txt').read().strip().splitlines()[-1]); print('$1 x $2 KiB: compress', d['compress_GBps'], 'GB/s', d['compress_ms'], 'ms; decompress', d['decompress_GBps'], d['decompress_ms'])
except Exception as e: print('$1 x $2 failed', e)
P
done
