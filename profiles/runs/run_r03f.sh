#!/bin/bash
# round 3, GPU session F: K3 with the carried flush tail + sorted frame order, K2 with foldable quad DPP adds
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03f && O=gpurun_out/r03f
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run default X=1
run default_prof ZHIP_PROF=1
run idx ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_k3idx.so
run idx_prof ZHIP_PROF=1 ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_k3idx.so
grep -h 'zhip-prof' $O/b_idx_prof.err | sed -n 6,11p
grep -h "zhip-prof" $O/b_default_prof.err | sed -n 6,11p
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
