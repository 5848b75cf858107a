#!/bin/bash
# round 6, GPU session K: bench.py's host_api object in the FULL-size process (65 536 frames resident), round 5's library against this round's
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06k && O=gpurun_out/r06k
export TMPDIR=/tmp
for V in r05 r06 r05 r06; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V = r05 ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_r05.so
  ZHIP_LIB=$L timeout 600 python bench.py --compress-frames 0 --no-extra --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', d['value'], d.get('host_api'))" | tee -a $O/host_api_in_full_bench.txt
done
