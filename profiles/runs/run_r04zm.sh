#!/bin/bash
# round 4, GPU session ZM: 131 072 sources per launch, the generalised function's NP = 2 instance against the hand-written two-probe function of the commit
# before (built as libzstd_hip_handwritten2.so), alternating in fresh processes on one box
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zm && O=gpurun_out/r04zm
export TMPDIR=/tmp
L=$PWD/python-zstandard_amd/csrc
R="python bench.py --config roundtrip --steps 2 --warmup 1 --no-cpu-baseline"
for i in 1 2; do for v in handwritten2 np2; do
  if [ $v = np2 ]; then timeout 300 $R > $O/rt_${i}_$v.json 2>> $O/err.txt; else ZHIP_LIB=$L/libzstd_hip_handwritten2.so timeout 300 $R > $O/rt_${i}_$v.json 2>> $O/err.txt; fi
  python - <<P
import json
l = json.loads(open('$O/rt_${i}_$v.json').read().strip().splitlines()[-1]); c = l['compress']
print('run $i $v: round trip', l['value'], 'compress', c['value'], {k.replace('zhip_encode_','').replace('_kernel',''): v['avg_ms'] for k, v in c['kernels'].items()})
P
done; done | tee $O/pairs.txt
