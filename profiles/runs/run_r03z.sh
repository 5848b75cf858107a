#!/bin/bash
# round 3, session Z: sources per wave of the several-block flat search (64 / 32 / 16 / 8) against the generic kernel
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03z && O=gpurun_out/r03z
export TMPDIR=/tmp
for cfg in "8192 256" "4096 512" "8192 1024" "2048 1024"; do
  set -- $cfg
  for lanes in 64 32 16 8 generic; do
    if [ $lanes = generic ]; then export ZHIP_MBC_MIN=100000000; unset ZHIP_MBC_LANES; else export ZHIP_MBC_MIN=0 ZHIP_MBC_LANES=$lanes; fi
    timeout 600 python tests/multiblock_rate.py $1 $2 > $O/mb_${1}x${2}_$lanes.txt 2>&1
    python - <<P
import json
try:
    d=json.loads(open('$O/mb_${1}x${2}_$lanes.txt').read().strip().splitlines()[-1]); print('$1 x $2 KiB lanes $lanes: compress', d['compress_GBps'], 'GB/s', d['compress_ms'], 'ms')
except Exception as e: print('$1 x $2 $lanes failed', e)
P
  done
done
