#!/bin/bash
# round 5, GPU session X: does K1 (2.2 ms, 11 waves per CU by its 14.4 KiB of LDS, 138 VGPRs) scale with its waves? ZHIP_K1_PER_CU = 6 / 8 / 10 / 11
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05x && O=gpurun_out/r05x
export TMPDIR=/tmp
for w in 11 10 8 6; do echo "ZHIP_K1_PER_CU=$w"; ZHIP_K1_PER_CU=$w timeout 300 python tests/tools/decode_variants_ab.py --steps 5 --rounds 1 product 2>&1 | grep -v amdgpu.ids; done | tee $O/k1_waves_per_cu.txt
