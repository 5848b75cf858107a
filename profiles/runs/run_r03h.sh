#!/bin/bash
# round 3, GPU session H: the new default K3 (96 VGPRs, five waves per SIMD): GPU suite, no-far-gather diagnostic, dictionary K3 at 4 waves
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03h && O=gpurun_out/r03h
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra"
L=$PWD/python-zstandard_amd/csrc
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run default X=1
run nofar ZHIP_BENCH_NO_VERIFY=1 ZHIP_LIB=$L/libzstd_hip_nofar.so
timeout 300 python bench.py --config dict --steps 5 --warmup 1 --no-cpu-baseline > $O/d_default.json 2> $O/d.err
ZHIP_LIB=$L/libzstd_hip_k3d4.so timeout 300 python bench.py --config dict --steps 5 --warmup 1 --no-cpu-baseline > $O/d_k3d4.json 2>> $O/d.err
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
for f in $O/d_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); d=l['decompress']; print(l['value'], d['value'], d['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
