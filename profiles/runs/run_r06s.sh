#!/bin/bash
# round 6, GPU session S: where the side stream (K1b beside K2) belongs. ZHIP_X_SIDE = 0 never; 1 every batch but those of small frames (r06q's product); 2 only batches of
# ONE chunk; 3 like 1, and batches of several chunks run them one after the other on one slot stream. The round-trip config's 131 072 frames (two chunks) and the host-buffer
# calls (chunks of 32 768 behind a first one of 2 048) under each.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06s && O=gpurun_out/r06s
export TMPDIR=/tmp
for i in 1 2; do for X in 0 1 2 3; do
  ZHIP_X_SIDE=$X timeout 900 python bench.py --config roundtrip --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); x=d['decompress']; print('xside=$X roundtrip 131072', x['value'], x['ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in x['kernels'].items()})" | tee -a $O/side_rule_ab.txt
done; done
for i in 1 2 3; do for X in 0 1; do
  ZHIP_X_SIDE=$X timeout 900 python tests/host_api_rate.py 65536 2>/dev/null | tail -1 | sed "s/^/xside=$X host_api 65536 /" | cut -c1-400 | tee -a $O/side_rule_ab.txt
done; done
for i in 1 2; do for X in 0 1; do
  ZHIP_X_SIDE=$X timeout 900 python tests/host_api_rate.py 8192 2>/dev/null | tail -1 | sed "s/^/xside=$X host_api 8192 /" | cut -c1-400 | tee -a $O/side_rule_ab.txt
done; done
