#!/bin/bash
# round 3, session ZI: the final build's record line (default bench with dict / roundtrip / blocks sub-objects), GPU suite, smoke
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03zi && O=gpurun_out/r03zi
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python tests/tools/status_from_bench.py $O/bench_default.json | cut -c1-330
