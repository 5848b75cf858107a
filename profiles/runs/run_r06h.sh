#!/bin/bash
# round 6, GPU session H: configs[3] with K1's lane pass claiming its bins per wave (kernel trace), the GPU suite on the knob-pruned build, the driver-style default line
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06h && O=gpurun_out/r06h
export TMPDIR=/tmp
P=$O/kt; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python bench.py --config dict --no-cpu-baseline --steps 3 --warmup 1 > $P/bench.json 2> $P/err.log
python - $P <<'PY' | tee $O/dict_kernel_trace.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f))):
        if "zhip" in r.get("Name", ""): print(r.get("Name", "")[:44], r.get("Calls"), "avg_ns", r.get("AverageNs"), "min", r.get("MinNs"), "max", r.get("MaxNs"))
PY
tail -1 $P/bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dict compress', d['value'], 'decompress', d['decompress']['value'], d['decompress']['ms_per_step'])" | tee -a $O/dict_kernel_trace.txt
find $P -name "*.csv" -delete; find $P -name "*.db" -delete
timeout 600 python bench.py --config dict --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dict compress', d['value'], 'decompress', d['decompress']['value'], d['decompress']['ms_per_step'], d['decompress']['kernels'])" | tee $O/dict.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python -c "
import json,sys; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('summary'))); print(d['kernels'])" | tee $O/bench_default_summary.txt
