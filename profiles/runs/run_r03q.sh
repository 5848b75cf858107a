#!/bin/bash
# round 3, session Q: GPU suite after the hostile-frame / 18-bit match length fixes, default bench line (perf unchanged?)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03q && O=gpurun_out/r03q
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -5 $O/pytest_gpu.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra > $O/bench_noextra.json 2> $O/bench_noextra.err ) 2> $O/bench.time
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03q/bench_noextra.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
P
