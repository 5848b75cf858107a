#!/bin/bash
# round 6, GPU session ZI: the flat search hands a lane over to four probes per trip once fewer than T lanes of its wave are still searching (ZHIP_X_SWITCH=T; 0 = never):
# a launch's sources start together and end apart, the search is transfer-bound while most lanes are busy and chain-bound at the end. Parity, then T swept at 65 536 and 131 072 sources.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zi && O=gpurun_out/r06zi
export TMPDIR=/tmp
ZHIP_X_SWITCH=32 timeout 1500 python -m pytest tests/test_gpu_compress.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2 | tee $O/pytest_compress_switch32.txt
for i in 1 2; do for T in 0 16 32 48 56; do
  ZHIP_X_SWITCH=$T timeout 600 python bench.py --config compress --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('T=$T 65536', d['value'], d['ms_per_step'], {n.replace('zhip_encode_','').replace('_kernel',''): round(v['avg_ms'],2) for n,v in d['kernels'].items()}, d['regime']['table_pick'])" | tee -a $O/switch_sweep.txt
done; done
for T in 0 32 48; do
  ZHIP_X_SWITCH=$T timeout 900 python bench.py --config roundtrip --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('T=$T 131072', d['compress']['value'], d['compress']['ms_per_step'], {n.replace('zhip_encode_','').replace('_kernel',''): round(v['avg_ms'],2) for n,v in d['compress']['kernels'].items()})" | tee -a $O/switch_sweep.txt
done
