#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03m && O=gpurun_out/r03m
export TMPDIR=/tmp
L=$PWD/python-zstandard_amd/csrc
ZHIP_BENCH_NO_VERIFY=1 ZHIP_LIB=$L/libzstd_hip_k1na.so timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra > $O/b_k1na.json 2> $O/b.err
python -c "
import json
l=json.loads(open('$O/b_k1na.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})"
tail -3 $O/b.err
