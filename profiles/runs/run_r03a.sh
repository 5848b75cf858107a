#!/bin/bash
# round 3, GPU session A: memory-system facts (random-gather rate by region size, lane-per-frame executor mock, same-wave RAW),
# the decode pipeline at fewer K3 waves per CU (Infinity-Cache locality), K2 + K3 co-resident variants
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03a && O=gpurun_out/r03a
export TMPDIR=/tmp
timeout 300 tests/ubench/membench 65536 > $O/membench_65536.txt 2>&1; echo "membench rc $?" >> $O/membench_65536.txt
timeout 200 tests/ubench/membench 131072 nog > $O/membench_131072.txt 2>&1
timeout 200 tests/ubench/membench 262144 nog > $O/membench_262144.txt 2>&1
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --compress-frames 0"
timeout 300 $B > $O/bench_base.json 2> $O/bench_base.err
for k in 4 8 12 24; do ZHIP_K3_PER_CU=$k timeout 300 $B > $O/bench_k3percu$k.json 2>> $O/bench_base.err; done
for v in coq12 coq10; do
  for ch in 8192 16384 32768; do
    ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_$v.so ZHIP_DCHUNK=$ch ZHIP_NSLOT=3 timeout 300 $B > $O/bench_${v}_chunk$ch.json 2>> $O/bench_base.err
  done
done
for ch in 8192 16384 32768; do ZHIP_DCHUNK=$ch ZHIP_NSLOT=3 timeout 300 $B > $O/bench_base_chunk$ch.json 2>> $O/bench_base.err; done
tail -5 $O/bench_base.err
for f in $O/bench_*.json; do echo "$f: $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
