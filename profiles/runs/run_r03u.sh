#!/bin/bash
# round 3, session U: sources of several blocks in the flat match kernel (compress) -- GPU suite, rates, the default bench line (regressions?)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03u && O=gpurun_out/r03u
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -5 $O/pytest_gpu.txt
timeout 400 python tests/multiblock_rate.py 2048 1024 > $O/mb_2048x1024.txt 2>&1; tail -1 $O/mb_2048x1024.txt
ZHIP_BLOCKS=0 timeout 400 python tests/multiblock_rate.py 2048 1024 > $O/mb_2048x1024_generic.txt 2>&1; tail -1 $O/mb_2048x1024_generic.txt
timeout 400 python tests/multiblock_rate.py 8192 256 > $O/mb_8192x256.txt 2>&1; tail -1 $O/mb_8192x256.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench.time; tail -3 $O/bench.time
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03u/bench_default.json').read().strip().splitlines()[-1])
print('decompress', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
c=d['compress']; print('compress', c['value'], c['ms_per_step'], {k:v['avg_ms'] for k,v in c['kernels'].items()})
print('dict', d['dict']['value'], d['dict']['decompress']['value'], 'roundtrip', d['roundtrip']['value'])
P
