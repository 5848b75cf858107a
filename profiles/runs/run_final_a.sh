#!/bin/sh
# round-end style validation on the GPU box: the -m gpu suite, smoke(), the default bench line and the compress-direction line
cd /root/repo
mkdir -p gpurun_out

( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/final_pytest.log 2>&1
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/final_smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
timeout 400 python bench.py --direction compress --steps 3 --warmup 1 > gpurun_out/final_bench_compress.json 2> gpurun_out/final_bench_compress.err
cat gpurun_out/final_pytest.log gpurun_out/final_smoke.log; cat gpurun_out/final_bench.json gpurun_out/final_bench_compress.json; tail -n 4 gpurun_out/final_bench*.err
