#!/bin/bash
# round 3, GPU session G: K3 register diet (literal runs above 16 bytes to the units: 113 -> 105 VGPRs) and five waves per SIMD
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03g && O=gpurun_out/r03g
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra"
L=$PWD/python-zstandard_amd/csrc
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }

run lf16 ZHIP_LIB=$L/libzstd_hip_lf16.so
run lf16w5 ZHIP_LIB=$L/libzstd_hip_lf16w5.so
run lf16w6 ZHIP_LIB=$L/libzstd_hip_lf16w6.so
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
