#!/bin/bash
# round 4, GPU session Q: K3's LDS hand-offs as wave-level fences instead of __syncthreads() (which also waits for the wave's global stores), exact need-masks vs the cell map
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04q && O=gpurun_out/r04q
export TMPDIR=/tmp
L=$PWD/python-zstandard_amd/csrc
( time ZHIP_LIB=$L/libzstd_hip_lsync.so timeout 1200 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu_lsync.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu_lsync.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra --no-host-api"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run base X=1
run lsync ZHIP_LIB=$L/libzstd_hip_lsync.so
run exact ZHIP_LIB=$L/libzstd_hip_exact.so
run lsync7 ZHIP_LIB=$L/libzstd_hip_lsync7.so
run lsync_b ZHIP_LIB=$L/libzstd_hip_lsync.so
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
