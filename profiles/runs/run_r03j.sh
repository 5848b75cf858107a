#!/bin/bash
# round 3, GPU session J (the round's record): GPU suite, smoke, the default bench line (configs[1-4]), rocprofv3 passes at the default chunk,
# host-API rates and small-batch latencies, the reference's own hot-path tests
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03j && O=gpurun_out/r03j
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
timeout 300 python tests/small_batch_latency.py > $O/small_batch_latency.txt 2>&1; tail -1 $O/small_batch_latency.txt
timeout 400 python tests/host_api_rate.py > $O/host_api_rate.txt 2>&1; tail -3 $O/host_api_rate.txt
timeout 400 python tests/multiblock_rate.py 2048 1024 > $O/multiblock_rate.txt 2>&1; tail -1 $O/multiblock_rate.txt
sh tests/run_reference_hotpath_tests.sh > $O/reference_tests_tail.txt 2>&1; tail -2 $O/reference_tests_tail.txt; cp gpurun_out/reference_hotpath_tests.log $O/ 2>/dev/null
TAG=r03 sh tests/run_profiles.sh > $O/run_profiles.log 2>&1; tail -25 $O/run_profiles.log
