#!/bin/bash
# round 5, GPU session U: the round's last build -- the GPU suite, smoke(), and bench.py exactly as the driver runs it
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05u && O=gpurun_out/r05u
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.txt
S=$(date +%s); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? wall $(( $(date +%s) - S )) s" | tee $O/bench_default.rc
