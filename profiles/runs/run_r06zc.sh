#!/bin/bash
# round 6, GPU session ZC: the table pick with up to six candidates where probes are cheap and the first three are alike (dictionary batches): eight fresh processes, what each kept
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zc && O=gpurun_out/r06zc
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_compress.py -x -q -m gpu -k "dict or pick" 2>&1 | tail -2 | tee $O/pytest_pick.txt
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python bench.py --config dict --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dict', d.get('value'), d.get('ms_per_step'), round(d['kernels']['zhip_encode_match_flat_kernel']['avg_ms'],2), d.get('table_pick'))" | tee -a $O/dict_pick_six.txt
done
