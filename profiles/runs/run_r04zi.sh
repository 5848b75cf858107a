#!/bin/bash
# round 4, GPU session ZI: three against two probes per trip at 65 536 sources per launch in FRESH processes, alternating (r04zg compared them inside one process)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zi && O=gpurun_out/r04zi
export TMPDIR=/tmp
C="python bench.py --config compress --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-host-api"
for i in 1 2 3 4; do for f in 0 1; do ZHIP_FLAT3=$f timeout 300 $C > $O/b_${i}_flat3_$f.json 2>> $O/err.txt; python - <<P
import json
l = json.loads(open('$O/b_${i}_flat3_$f.json').read().strip().splitlines()[-1])
print('run $i ZHIP_FLAT3=$f', l['value'], 'GB/s', {k.replace('zhip_encode_','').replace('_kernel',''): v['avg_ms'] for k, v in l['kernels'].items()})
P
done; done | tee $O/fresh_process_pairs.txt
