#!/bin/bash
# round 3, session S (skippable frames, saturated offsets): frames of several blocks through the phase-split decode kernels -- GPU suite, rates against the generic kernel, headline regression
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03s && O=gpurun_out/r03s
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -5 $O/pytest_gpu.txt
timeout 400 python tests/multiblock_rate.py 2048 1024 > $O/mb_2048x1024.txt 2>&1; tail -1 $O/mb_2048x1024.txt
ZHIP_BLOCKS=0 timeout 400 python tests/multiblock_rate.py 2048 1024 > $O/mb_2048x1024_generic.txt 2>&1; tail -1 $O/mb_2048x1024_generic.txt
timeout 400 python tests/multiblock_rate.py 8192 256 > $O/mb_8192x256.txt 2>&1; tail -1 $O/mb_8192x256.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > $O/bench_noextra.json 2> $O/bench_noextra.err ) 2> $O/bench.time
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03s/bench_noextra.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
P
