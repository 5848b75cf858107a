#!/bin/bash
# round 3, GPU session N: K3 without the phase-timer accumulators in its registers (77 VGPRs): 5 / 6 / 7 / 8 waves per SIMD; dictionary K3 at 4
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03n && O=gpurun_out/r03n
export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra"
L=$PWD/python-zstandard_amd/csrc
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run w5 X=1
run w6 ZHIP_LIB=$L/libzstd_hip_w6.so
run w7 ZHIP_LIB=$L/libzstd_hip_w7.so
run w8 ZHIP_LIB=$L/libzstd_hip_w8.so
ZHIP_LIB=$L/libzstd_hip_w6.so timeout 300 python bench.py --config dict --steps 5 --warmup 1 --no-cpu-baseline > $O/d_w6.json 2> $O/d.err
ZHIP_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra > $O/b_prof.json 2> $O/b_prof.err; grep -h "zhip-prof" $O/b_prof.err | sed -n 6,11p
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
python -c "
import json
l=json.loads(open('$O/d_w6.json').read().strip().splitlines()[-1]); d=l['decompress']; print('dict', l['value'], d['value'], d['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in d['kernels'].items()})"
