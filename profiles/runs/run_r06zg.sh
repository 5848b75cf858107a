#!/bin/bash
# round 6, GPU session ZG: launch numbers in the cells of the dictionary-less flat search too (position 18 | tag 8 | launch number 6: no 25 GiB memset per launch) against the
# build that zeroes (-DZHIP_TABLE_EPOCHS=0): the compress-side GPU tests and the full-size gate, then compress / round trip / host-buffer calls alternating
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zg && O=gpurun_out/r06zg
export TMPDIR=/tmp
timeout 2000 python -m pytest tests/test_gpu_compress.py tests/test_gpu_boundary.py tests/test_gpu_fullsize.py tests/test_cext_backend.py tests/test_gpu_multidevice.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_compress.txt
timeout 900 python tests/stress_gpu_compress.py 2>&1 | tail -1 | tee $O/stress_gpu_compress.txt
for i in 1 2 3; do for V in noepoch product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python bench.py --config compress --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V 65536', d['value'], d['ms_per_step'], {n.replace('zhip_encode_','').replace('_kernel',''): round(v['avg_ms'],2) for n,v in d['kernels'].items()}, d['regime']['table_pick'])" | tee -a $O/table_epochs_ab.txt
done; done
for V in noepoch product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 900 python bench.py --config roundtrip --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V 131072', d['compress']['value'], d['compress']['ms_per_step'], {n.replace('zhip_encode_','').replace('_kernel',''): round(v['avg_ms'],2) for n,v in d['compress']['kernels'].items()})" | tee -a $O/table_epochs_ab.txt
  ZHIP_LIB=$L timeout 900 python tests/host_api_rate.py 65536 2>/dev/null | tail -1 | sed "s/^/$V /" | cut -c1-230 | tee -a $O/table_epochs_ab.txt
done
