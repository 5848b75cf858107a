#!/bin/bash
# round 3, GPU session E: K3 after the instruction diet (aligned flush with a carried tail, long items copied by the wave), and at 5 / 6 waves per SIMD
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03e && O=gpurun_out/r03e
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py -x -q > $O/pytest_dec.txt 2>&1; tail -2 $O/pytest_dec.txt
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra"
L=$PWD/python-zstandard_amd/csrc
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run diet X=1
run diet_w5 ZHIP_LIB=$L/libzstd_hip_k3w5.so
run diet_w6 ZHIP_LIB=$L/libzstd_hip_k3w6.so
run diet_prof ZHIP_PROF=1
grep -h "zhip-prof" $O/b_diet_prof.err | sed -n 6,11p
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
