#!/bin/bash
# round 4, GPU session A: the rewritten K3 (sequential in-batch matches, own-lane items up to 32 bytes) against round 3's, 5-8 waves per SIMD, own-lane limit 16
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04a && O=gpurun_out/r04a
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -5 $O/pytest_gpu.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra"
L=$PWD/python-zstandard_amd/csrc
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run v2 X=1
run v1 ZHIP_LIB=$L/libzstd_hip_k3v1.so
run own16 ZHIP_LIB=$L/libzstd_hip_own16.so
run w5 ZHIP_LIB=$L/libzstd_hip_w5.so
run w7 ZHIP_LIB=$L/libzstd_hip_w7.so
run w8 ZHIP_LIB=$L/libzstd_hip_w8.so
ZHIP_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra > $O/b_prof.json 2> $O/b_prof.err; grep -h "zhip-prof" $O/b_prof.err | tail -12
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
tail -3 $O/*.err | tail -30
