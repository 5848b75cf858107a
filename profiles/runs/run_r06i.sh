#!/bin/bash
# round 6, GPU session I: the host-buffer calls at 8 192 x 128 KiB, round 5's library against this round's (the driver-style line showed decompress 28.6 -> 20.5 GB/s
# there while 65 536 frames went 44.8 -> 48.2): same process shape, two runs each, and this round's with the working set NOT kept (ZHIP_KEEP_GB=0)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06i && O=gpurun_out/r06i
export TMPDIR=/tmp
for i in 1 2; do
  ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_r05.so timeout 600 python tests/host_api_rate.py 8192 2>/dev/null | tail -1 | sed 's/^/r05     /' | tee -a $O/host_api_8192.txt
  timeout 600 python tests/host_api_rate.py 8192 2>/dev/null | tail -1 | sed 's/^/r06     /' | tee -a $O/host_api_8192.txt
done
ZHIP_KEEP_GB=0 timeout 600 python tests/host_api_rate.py 8192 2>/dev/null | tail -1 | sed 's/^/r06keep0 /' | tee -a $O/host_api_8192.txt
