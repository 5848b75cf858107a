#!/bin/bash
# round 5, GPU session N: the round's closing evidence on the final build -- (1) the GPU suite, (2) parity stress campaigns (single-block sources through the
# default kernels and the flat kernel only; sources of 1-9 blocks through the flat several-block search), (3) the entropy kernel at three / four / five
# waves per SIMD, (4) the rocprofv3 passes of tests/run_profiles.sh (kernel traces, FETCH / WRITE, SQ counters -> profiles/r05_*, traffic.json), (5) bench.py
# exactly as the driver runs it
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05n && O=gpurun_out/r05n
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
{ for s in 31 32; do timeout 300 python tests/stress_gpu_compress.py $s 2>&1 | grep -v amdgpu.ids | tail -3; done
  for s in 41 42 43; do timeout 300 python tests/stress_gpu_blocks.py $s 2>&1 | tail -1; done; } | tee $O/stress_gpu.txt
for v in product e2w3 e2w5; do
  L=python-zstandard_amd/csrc/libzstd_hip.so; [ $v != product ] && L=python-zstandard_amd/csrc/libzstd_hip_$v.so
  ZHIP_LIB=$PWD/$L timeout 300 python bench.py --config compress --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], {k.replace('zhip_encode_',''): v['avg_ms'] for k, v in d['kernels'].items()})"
done | tee $O/e2_waves.txt
TAG=r05 sh tests/run_profiles.sh > $O/run_profiles.log 2>&1; tail -5 $O/run_profiles.log
mkdir -p $O/summary && cp gpurun_out/summary/* $O/summary/ 2>/dev/null
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
