#!/bin/bash
# round 6, GPU session ZZI: the several-block flat search against the generic kernel at SMALL batches once more (the threshold dates from r03u / r03v: 2 048 x 1 MiB flat 2.5 s / generic 1.04 s),
# on the round's last kernels: ZHIP_MBC_MIN=0 forces the flat search
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzi && O=gpurun_out/r06zzi
export TMPDIR=/tmp
for sh in ${SHAPES:-2048:1024 4096:512 4096:256 1024:1024}; do shape="${sh/:/ }"
  for m in default 0; do
    if [ $m = default ]; then unset ZHIP_MBC_MIN; else export ZHIP_MBC_MIN=0; fi
    timeout 600 python tests/multiblock_rate.py $shape 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$shape', 'MBC_MIN=$m', 'compress', d['compress_GBps'], 'GB/s', d['compress_ms'], 'ms  decompress', d['decompress_GBps'])" | tee -a $O/mbc_small_batches.txt
  done
done
