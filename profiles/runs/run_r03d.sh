#!/bin/bash
# round 3, GPU session D: K3 with the LDS history window against the windowless form (time, phase timers), window sizes
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03d && O=gpurun_out/r03d
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decompress.py -x -q > $O/pytest_dec.txt 2>&1; tail -2 $O/pytest_dec.txt
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra"
L=$PWD/python-zstandard_amd/csrc
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run win X=1
run nowin ZHIP_LIB=$L/libzstd_hip_nowin.so
run win6k ZHIP_LIB=$L/libzstd_hip_win6k.so
run win12k ZHIP_LIB=$L/libzstd_hip_win12k.so
run win_prof ZHIP_PROF=1
run nowin_prof ZHIP_PROF=1 ZHIP_LIB=$L/libzstd_hip_nowin.so
grep -h "zhip-prof" $O/b_win_prof.err | tail -14; echo ----; grep -h "zhip-prof" $O/b_nowin_prof.err | tail -14
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
