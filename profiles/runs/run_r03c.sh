#!/bin/bash
# round 3, GPU session C: the whole GPU suite on the new library (flat dictionary search, sharded compaction export), configs[3] after the
# flattening, and the K3 diagnostic without far-match gathers (how much of K3 is the random-gather rate of the memory system)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03c && O=gpurun_out/r03c
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -4 $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.time
timeout 600 python bench.py --config dict --steps 5 --warmup 1 > $O/bench_dict.json 2> $O/bench_dict.err; tail -2 $O/bench_dict.err
ZHIP_NO_FLAT=1 timeout 600 python bench.py --config dict --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dict_noflat.json 2>> $O/bench_dict.err
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra"
ZHIP_BENCH_NO_VERIFY=1 ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_nofar.so timeout 300 $B > $O/b_nofar.json 2> $O/b.err
for f in $O/b_*.json $O/bench_dict*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()}, l.get('decompress',{}).get('value'))
except Exception as e: print('ERR', e)
")"; done
