#!/bin/bash
# round 6, GPU session ZL: the host-buffer calls INSIDE bench.py's process (where the default line measures them: 49 -> 43 -> 38-39 GB/s over the round's lines, while the same calls
# alone stayed at 47-49) with the library as it is, without K0, and without K0 and the side stream
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zl && O=gpurun_out/r06zl
export TMPDIR=/tmp
for i in 1 2; do for V in noside product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', d['value'], d['ms_per_step'], json.dumps(d.get('host_api')))" | cut -c1-420 | tee -a $O/host_api_in_bench_ab2.txt
done; done
