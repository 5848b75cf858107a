#!/bin/bash
# round 4, GPU session E: K3 round-4 form with the branch-free sequential loop for in-batch matches (18 instructions per match instead of 37), 5-8 waves per SIMD
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04f && O=gpurun_out/r04f
export TMPDIR=/tmp
L=$PWD/python-zstandard_amd/csrc
( time ZHIP_LIB=$L/libzstd_hip_k3r4.so timeout 1200 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu_k3r4.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -5 $O/pytest_gpu_k3r4.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run base X=1
run k3r4 ZHIP_LIB=$L/libzstd_hip_k3r4.so

run k3r4w7 ZHIP_LIB=$L/libzstd_hip_k3r4w7.so
run k3r4o16 ZHIP_LIB=$L/libzstd_hip_k3r4o16.so

ZHIP_PROF=1 ZHIP_LIB=$L/libzstd_hip_k3r4.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra > $O/b_prof.json 2> $O/b_prof.err; grep -h "zhip-prof" $O/b_prof.err | sed -n 6,11p
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
