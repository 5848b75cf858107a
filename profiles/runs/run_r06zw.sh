#!/bin/bash
# round 6, GPU session ZW: fast-strategy batches (levels 1, 2, negative) above 32 768 sources -- every source in flight at once (65 536 per chunk, sixteen per wave) against two
# chunks of 32 768 at eight per wave (-DZHIP_FAST_WIDE=0)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zw && O=gpurun_out/r06zw
export TMPDIR=/tmp
for V in nofastwide product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 1500 python tests/tools/compress_levels_rate.py 65536 2>&1 | tail -1 | sed "s/^/$V /" | tee -a $O/compress_levels_65536.txt
done
