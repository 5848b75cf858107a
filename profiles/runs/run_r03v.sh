#!/bin/bash
# round 3, session V: where the flat search for sources of several blocks starts to pay (batch size sweep, ZHIP_MBC_MIN=0 forces it on, default 4096)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03v && O=gpurun_out/r03v
export TMPDIR=/tmp
for cfg in "4096 512" "4096 256" "16384 256" "3072 384"; do
  set -- $cfg
  for mode in flat generic; do
    if [ $mode = flat ]; then export ZHIP_MBC_MIN=0; else export ZHIP_MBC_MIN=100000000; fi
    timeout 600 python tests/multiblock_rate.py $1 $2 > $O/mb_${1}x${2}_$mode.txt 2>&1
    python - <<P
import json
try:
    d=json.loads(open('$O/mb_${1}x${2}_$mode.txt').read().strip().splitlines()[-1]); print('$1 x $2 KiB $mode: compress', d['compress_GBps'], 'GB/s', d['compress_ms'], 'ms; decompress', d['decompress_GBps'])
except Exception as e: print('$1 x $2 $mode failed', e)
P
  done
done
