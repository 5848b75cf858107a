#!/bin/bash
# round 5, GPU session E: the product K3 with fewer waves per CU (ZHIP_K3_PER_CU: the floor kernel is FASTER at 16 per CU than at 28, r05c --
# fewer frames in flight leave more of the L2 to each), and with non-temporal sequence / literal loads at those occupancies
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05e && O=gpurun_out/r05e
export TMPDIR=/tmp
for w in 24 20 16 12; do
  echo "ZHIP_K3_PER_CU=$w"
  ZHIP_K3_PER_CU=$w timeout 600 python tests/tools/decode_variants_ab.py --steps 5 --rounds 1 product nt1 floor 2>&1 | grep -v amdgpu.ids
done | tee $O/k3_waves_per_cu.txt
