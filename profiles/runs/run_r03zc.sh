#!/bin/bash
# round 3, session ZC: the final build's several-block rates and latencies (after the slot sizing and sources-per-wave changes), GPU suite, smoke, default bench line
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03zc && O=gpurun_out/r03zc
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
for cfg in "2048 1024" "16384 256" "8192 256"; do set -- $cfg; timeout 600 python tests/multiblock_rate.py $1 $2 > $O/multiblock_${1}x${2}KiB.txt 2>&1; tail -1 $O/multiblock_${1}x${2}KiB.txt | cut -c1-700; done
timeout 400 python tests/small_batch_latency.py > $O/small_batch_latency.txt 2>&1; tail -1 $O/small_batch_latency.txt
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03zc/bench_default.json').read().strip().splitlines()[-1])
print('decompress', d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
c=d['compress']; print('compress', c['value'], c['ms_per_step']); print('dict', d['dict']['value'], d['dict']['decompress']['value'], 'roundtrip', d['roundtrip']['value'])
P
