#!/bin/bash
# round 6, GPU session ZZZ10: the kernel trace of the DEFAULT command itself (python bench.py --steps 20 --warmup 5, every leg), beside the per-direction traces of tests/run_profiles.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06zzz10; O=gpurun_out/r06zzz10; P=/tmp/prof_default; rm -rf $P
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python bench.py --steps 20 --warmup 5 > $O/bench_default_under_rocprof.json 2> $O/err.log; echo "rc $?"
f=$(find $P -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $O/default_command_kernel_stats.csv
python - <<'PY' | tee gpurun_out/r06zzz10/summary.txt
import json
d = json.loads(open('gpurun_out/r06zzz10/bench_default_under_rocprof.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('summary')))
print(json.dumps({k: v for k, v in d.get('kernels', {}).items()})[:900])
PY
head -14 $O/default_command_kernel_stats.csv | cut -c1-160
