#!/bin/bash
# round 6, GPU session J: bench.py's host_api object at 8 192 frames, round 5's library against this round's, inside bench.py's own process shape
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06j && O=gpurun_out/r06j
export TMPDIR=/tmp
for i in 1 2; do for V in r05 r06; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V = r05 ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_r05.so
  ZHIP_LIB=$L timeout 600 python bench.py --frames 8192 --compress-frames 0 --no-extra --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', d['value'], d.get('host_api'))" | tee -a $O/host_api_in_bench.txt
done; done
