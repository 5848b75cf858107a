#!/bin/bash
# round 6, GPU session ZZZ3: do the DECODE step's allocations have placement kinds like the match kernel's tables? The same 65 536 frames decoded in fresh processes with a dummy allocation of
# 0 ... 150 GiB held in front of the run's own (source, destination, the context's arenas land on other physical pages each time): the step and its kernels
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzz3 && O=gpurun_out/r06zzz3
export TMPDIR=/tmp
timeout 900 python tests/tools/decode_variants_ab.py --frames 65536 --steps 10 --rounds 1 product 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400 | sed 's/^/hold 0 GiB (parent run): /' | tee -a $O/decode_placement.txt
export ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip.so
for h in 0 5 13 29 47 71 103 150; do
  ZHIP_AB_HOLD_GIB=$h timeout 600 python tests/tools/decode_variants_ab.py --frames 65536 --steps 10 --child product 2>/dev/null | grep "^step" | cut -c1-400 | sed "s/^/hold $h GiB: /" | tee -a $O/decode_placement.txt
done
