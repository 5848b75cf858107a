#!/bin/bash
# round 6, GPU session ZB: the host-buffer calls (multi_*_to_buffer through Python, PCIe inclusive) on the round's last build (the library of the session's start, commit 98e1054, was meant to alternate with it and did not build from the archive: only the product ran)
# (commit 98e1054: no side stream, no K0), alternating in one session -- the default line's host_api figure moved 49 -> 43 -> 38 GB/s across three boxes
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zb && O=gpurun_out/r06zb
export TMPDIR=/tmp
for i in 1 2 3; do for V in r06start product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 900 python tests/host_api_rate.py 65536 2>/dev/null | tail -1 | sed "s/^/$V 65536 /" | cut -c1-260 | tee -a $O/host_api_ab.txt
done; done
for i in 1 2; do for V in r06start product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 900 python tests/host_api_rate.py 8192 2>/dev/null | tail -1 | sed "s/^/$V 8192 /" | cut -c1-260 | tee -a $O/host_api_ab.txt
done; done
