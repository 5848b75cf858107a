#!/bin/bash
# round 6, GPU session ZZZ5: where the first call's 5 s go -- hipMalloc / hipMemset / hipFree of table-sized allocations by themselves, and the first call with three candidates against eight
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzz5 && O=gpurun_out/r06zzz5
export TMPDIR=/tmp
timeout 300 python tests/tools/malloc_time.py 2>/dev/null | tee $O/malloc_time.txt
for V in pick3 product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python tests/tools/first_call_cost.py 65536 2>/dev/null | tail -1 | sed "s/^/$V: /" | tee -a $O/first_call_cost.txt
done
