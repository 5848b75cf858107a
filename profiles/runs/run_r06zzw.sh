#!/bin/bash
# round 6, GPU session ZZW: the placement pick over EIGHT candidates (short probes) against three (-DZHIP_PICK_CANDIDATES=3): compress-side GPU tests on the product, then the compress bench in five
# alternating pairs of processes -- which candidate was kept, what the real launches take, the step
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzw && O=gpurun_out/r06zzw
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_compress.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300 | tee $O/pytest_compress.txt
for i in 1 2 3 4 5; do for V in pick3 product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python bench.py --config compress --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-host-api 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d.get('regime') or (d.get('compress') or {}).get('regime'); print('$V', d.get('value'), 'GB/s', d.get('ms_per_step'), 'ms', json.dumps({k: r[k] for k in ('match_kernel_ms_per_65536_frames', 'class')}), json.dumps({k: r['table_pick'][k] for k in ('candidates_ms', 'kept')}), 'verified', d.get('verified'))" | tee -a $O/pick8_ab.txt
done; done
