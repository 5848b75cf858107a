#!/bin/bash
# round 4, GPU session S: K1 with a register-cached forward bit reader (distribution parsing) and fewer refills in the weights loop
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04s && O=gpurun_out/r04s
export TMPDIR=/tmp
L=$PWD/python-zstandard_amd/csrc
( time timeout 1200 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra --no-host-api"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run base X=1
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
run base2 X=1
for f in $O/b_base*.json; do echo "$(basename $f): $(python -c "
import json,sys
l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
")"; done
ZHIP_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra --no-host-api > $O/b_prof.json 2> $O/b_prof.err; grep -h "zhip-prof" $O/b_prof.err | sed -n 1,5p
