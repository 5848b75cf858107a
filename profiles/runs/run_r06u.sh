#!/bin/bash
# round 6, GPU session U: the dictionary batch's match kernel by documents per launch -- would two or four launches cost what one of 262 144 does (the entropy kernel of
# launch k could then run beside the match kernel of launch k + 1)?
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06u && O=gpurun_out/r06u
export TMPDIR=/tmp
for i in 1 2; do for D in 262144 131072 65536 32768; do
  timeout 600 python bench.py --config dict --docs $D --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('docs=$D', d.get('value'), d.get('ms_per_step'), {n.replace('zhip_encode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in d['kernels'].items()}, d.get('table_pick'))" | tee -a $O/dict_docs_per_launch.txt
done; done
