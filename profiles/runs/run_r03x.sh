#!/bin/bash
# round 3, GPU session X (the round's record after the several-block modes and the parity fixes): GPU suite, smoke, the default bench line
# (configs[1-4]), rocprofv3 passes at the default chunk, host-API rates, small-batch and one-shot latencies, several-block rates, the reference's own hot-path tests
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03x && O=gpurun_out/r03x
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
timeout 400 python tests/small_batch_latency.py > $O/small_batch_latency.txt 2>&1; tail -1 $O/small_batch_latency.txt
ZHIP_BLOCKS=0 timeout 400 python tests/small_batch_latency.py > $O/small_batch_latency_generic_kernel.txt 2>&1; tail -1 $O/small_batch_latency_generic_kernel.txt
timeout 400 python tests/host_api_rate.py > $O/host_api_rate.txt 2>&1; tail -3 $O/host_api_rate.txt
timeout 400 python tests/multiblock_rate.py 2048 1024 > $O/multiblock_2048x1MiB.txt 2>&1; tail -1 $O/multiblock_2048x1MiB.txt
timeout 600 python tests/multiblock_rate.py 16384 256 > $O/multiblock_16384x256KiB.txt 2>&1; tail -1 $O/multiblock_16384x256KiB.txt
sh tests/run_reference_hotpath_tests.sh > $O/reference_tests_tail.txt 2>&1; tail -2 $O/reference_tests_tail.txt; cp gpurun_out/reference_hotpath_tests.log $O/ 2>/dev/null
TAG=r03 sh tests/run_profiles.sh > $O/run_profiles.log 2>&1; tail -28 $O/run_profiles.log
