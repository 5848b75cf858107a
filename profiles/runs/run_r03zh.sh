#!/bin/bash
# round 3, session ZH: the default bench line with the 'blocks' sub-object (frames of several blocks)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03zh && O=gpurun_out/r03zh
export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time; tail -3 $O/bench_default.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r03zh/bench_default.json').read().strip().splitlines()[-1])
print('decompress', d['value'], d['ms_per_step']); print('compress', d['compress']['value']); print('dict', d['dict'].get('value'), 'roundtrip', d['roundtrip'].get('value'))
print('blocks', json.dumps(d.get('blocks'))[:900])
P
