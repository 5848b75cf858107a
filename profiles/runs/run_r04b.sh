#!/bin/bash
# round 4, GPU session B: K3 with round-4 staging + dependency rounds, the lean K1b (2-byte cells, absolute bit cursor, 16 symbols per trip), against round 3's kernels
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04b && O=gpurun_out/r04b
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -5 $O/pytest_gpu.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra"
L=$PWD/python-zstandard_amd/csrc
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; }
run new X=1
run r03 ZHIP_LIB=$L/libzstd_hip_r03.so
run k3v1 ZHIP_LIB=$L/libzstd_hip_k3v1.so
run k1bv1 ZHIP_LIB=$L/libzstd_hip_k1bv1.so
run seqnear ZHIP_LIB=$L/libzstd_hip_seqnear.so
run huf16 ZHIP_LIB=$L/libzstd_hip_huf16.so
ZHIP_PROF=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra > $O/b_prof.json 2> $O/b_prof.err; grep -h "zhip-prof" $O/b_prof.err | sed -n 6,11p
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
for f in $O/*.err; do echo "== $f"; tail -n 3 $f; done 2>/dev/null | tail -30
