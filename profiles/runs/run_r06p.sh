#!/bin/bash
# round 6, GPU session P: K1b and K2 need nothing of each other (both follow K1 and the bins). One of them behind the other's launch on a side stream, so that its workgroups
# land where the first one's waves have drained -- the first kernel's tail (4.27 group rounds per K2 wave at 65 536 frames). ZHIP_X_SIDE=0 (one stream, the product so far),
# 1 (K2 first, K1b on the side stream), 2 (K1b first, K2 on the side stream): parity of the decode tests under each, then three bench runs each, headline + several-block + dictionary
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06p && O=gpurun_out/r06p
export TMPDIR=/tmp
for X in 1 2; do
  ZHIP_X_SIDE=$X timeout 900 python -m pytest tests/test_gpu_decompress.py -x -q -m gpu 2>&1 | tail -2 | sed "s/^/xside=$X /" | tee -a $O/pytest_decompress.txt
done
for i in 1 2 3; do for X in 0 1 2; do
  ZHIP_X_SIDE=$X timeout 600 python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('xside=$X', d['value'], d['ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in k.items()})" | tee -a $O/side_stream_ab.txt
done; done
for X in 0 1 2; do
  ZHIP_X_SIDE=$X timeout 600 python bench.py --config blocks --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('xside=$X blocks', d.get('value'), d.get('ms_per_step'), {k: (v.get('value'), v.get('ms_per_step')) for k, v in d.items() if isinstance(v, dict) and 'value' in v})" | tee -a $O/side_stream_ab.txt
  ZHIP_X_SIDE=$X timeout 600 python bench.py --config dict --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('xside=$X dict', d.get('value'), d.get('ms_per_step'), d['decompress']['value'], d['decompress']['ms_per_step'])" | tee -a $O/side_stream_ab.txt
done
