#!/bin/bash
# round 4, GPU session C: K1b in its two-level form (256-cell direct table + computed long codes: 768 bytes of tables per frame, 16 frames per wave, nine waves per CU);
# the library built with -Os / -O2 (K3 is issue-bound: does less unrolling help?)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04c && O=gpurun_out/r04c
export TMPDIR=/tmp
L=$PWD/python-zstandard_amd/csrc
( time ZHIP_LIB=$L/libzstd_hip_r4b.so timeout 1200 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu_r4b.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -5 $O/pytest_gpu_r4b.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run base X=1
run r4b ZHIP_LIB=$L/libzstd_hip_r4b.so
run os ZHIP_LIB=$L/libzstd_hip_os.so
run o2 ZHIP_LIB=$L/libzstd_hip_o2.so
run r4bos ZHIP_LIB=$L/libzstd_hip_r4bos.so
ZHIP_LIB=$L/libzstd_hip_r4b.so timeout 300 python bench.py --config dict --steps 5 --warmup 1 --no-cpu-baseline > $O/d_r4b.json 2> $O/d_r4b.err
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
python -c "
import json
l=json.loads(open('$O/d_r4b.json').read().strip().splitlines()[-1]); d=l['decompress']; print('dict', l['value'], d['value'], d['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in d['kernels'].items()})"
for f in $O/*.err; do echo "== $f"; tail -n 2 $f; done 2>/dev/null | tail -20
