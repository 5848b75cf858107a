#!/bin/bash
# round 5, GPU session O: bench.py exactly as the driver runs it, on the round's last build (pick threshold 12 %, traffic.json of r05)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05o && O=gpurun_out/r05o
export TMPDIR=/tmp
S=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? wall $(( $(date +%s) - S )) s" | tee $O/bench_default.rc
