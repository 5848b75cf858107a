#!/bin/bash
# round 3, GPU session I: K1b frames per wave (8 x 6 waves / 10 x 4 / 12 x 4 / 16 x 3 per CU)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03i && O=gpurun_out/r03i
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra"
L=$PWD/python-zstandard_amd/csrc
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
for v in huf10 huf12 huf16; do run $v ZHIP_LIB=$L/libzstd_hip_$v.so; done
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
