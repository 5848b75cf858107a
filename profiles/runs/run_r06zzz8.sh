#!/bin/bash
# round 6, tenth closing session (the pick's candidates bounded by the memory free when it begins) (+ KX / EX, the fast strategy's changes, block counts in the host decompress call): build check, smoke, the whole GPU suite, the driver-style default line -- on the round's last commit
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzz8 && O=gpurun_out/r06zzz8
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 | cut -c1-300 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python -c "
import json,sys; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('summary'))); print(json.dumps(d['roofline'].get('memory_requests'))); print(json.dumps(d['compress']['roofline'].get('memory_requests')))" | tee $O/bench_default_summary.txt
tail -c 2000 $O/bench_default.json > $O/bench_default_tail_2000.txt
