#!/bin/bash
# round 3, GPU session B: the changed GPU tests, the default bench line with its new sub-objects, K3 streaming-load variants,
# the structured two-stream pipeline (front of chunk k + 1 beside K3 of chunk k) with co-resident K2 shapes
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03b && O=gpurun_out/r03b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_boundary.py::test_dictionary_batch_config4_shape tests/test_gpu_decompress.py::test_truncated_and_corrupt_frames_do_not_crash -x -q > $O/pytest_changed.txt 2>&1; tail -3 $O/pytest_changed.txt
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time; tail -2 $O/bench_default.err
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra"
L=$PWD/python-zstandard_amd/csrc
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b.err; }
run k3nt1 ZHIP_LIB=$L/libzstd_hip_k3nt1.so
run k3nt3 ZHIP_LIB=$L/libzstd_hip_k3nt3.so
for ch in 8192 16384; do
  run split_base_c$ch ZHIP_SPLIT=1 ZHIP_DCHUNK=$ch ZHIP_NSLOT=3
  for v in coq12 coq12p coq10p coq13p; do
    run split_${v}_c$ch ZHIP_LIB=$L/libzstd_hip_$v.so ZHIP_SPLIT=1 ZHIP_DCHUNK=$ch ZHIP_NSLOT=3
    run split_${v}_c${ch}_k12 ZHIP_LIB=$L/libzstd_hip_$v.so ZHIP_SPLIT=1 ZHIP_DCHUNK=$ch ZHIP_NSLOT=3 ZHIP_K3_PER_CU=12
  done
done
run split_coq12p_c4096 ZHIP_LIB=$L/libzstd_hip_coq12p.so ZHIP_SPLIT=1 ZHIP_DCHUNK=4096 ZHIP_NSLOT=3
run split_coq12p_c32768 ZHIP_LIB=$L/libzstd_hip_coq12p.so ZHIP_SPLIT=1 ZHIP_DCHUNK=32768 ZHIP_NSLOT=2
tail -3 $O/b.err
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
