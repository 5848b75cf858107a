#!/bin/bash
# round 4, GPU session ZJ: large batches of sources of two blocks through the flat several-block search, now with four probes per trip for chunks of up to 32 768 sources
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zj && O=gpurun_out/r04zj
export TMPDIR=/tmp
for n in 16384 32768; do
  timeout 600 python tests/multiblock_rate.py $n 256 > $O/mb_${n}x256.log 2>&1; python - <<P
import json
l = json.loads(open('$O/mb_${n}x256.log').read().strip().splitlines()[-1])
print('$n x 256 KiB: compress %.2f GB/s (%.1f ms), decompress %.1f GB/s' % (l['compress_GBps'], l['compress_ms'], l['decompress_GBps']))
P
  ZHIP_FLAT4_MAX=0 timeout 600 python tests/multiblock_rate.py $n 256 > $O/mb_${n}x256_two.log 2>&1; python - <<P
import json
l = json.loads(open('$O/mb_${n}x256_two.log').read().strip().splitlines()[-1])
print('$n x 256 KiB, two probes: compress %.2f GB/s (%.1f ms)' % (l['compress_GBps'], l['compress_ms']))
P
done | tee $O/summary.txt
