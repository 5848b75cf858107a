#!/bin/bash
# round 5, GPU session Q: configs[3]'s decompress direction (262 144 x 4 KiB documents with the shared dictionary: four chunks of 65 536 on two slot streams,
# 8.8 ms) against the chunk size and the number of slot streams (ZHIP_DCHUNK / ZHIP_NSLOT)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05q && O=gpurun_out/r05q
export TMPDIR=/tmp
B="python bench.py --config dict --steps 5 --warmup 2 --no-cpu-baseline"
for cfg in "65536 2" "65536 3" "32768 3" "32768 2" "131072 2" "16384 3"; do
  set -- $cfg
  ZHIP_DCHUNK=$1 ZHIP_NSLOT=$2 timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); x=d['decompress']; print('DCHUNK $1 NSLOT $2  decompress %.1f GB/s %.2f ms  compress %.1f GB/s' % (x['value'], x['ms_per_step'], d['value']), {k.replace('zhip_decode_','').replace('_kernel',''): v['avg_ms'] for k, v in x['kernels'].items()})"
done | tee $O/dict_decompress_chunks.txt
