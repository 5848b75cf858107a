#!/bin/bash
# round 6, GPU session Z2: the decode step by batch size with / without K0 (-DZHIP_K0=0): K0 is a lane-serial walk whose latency a small batch cannot hide
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06z2 && O=gpurun_out/r06z2
export TMPDIR=/tmp
for i in 1 2; do for V in nok0 product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python tests/tools/decode_batch_sizes.py 1 64 512 2048 4096 8192 16384 32768 2>/dev/null | tail -1 | sed "s/^/$V /" | tee -a $O/k0_by_batch_size.txt
done; done
