#!/bin/bash
# round 6, GPU session ZZM: the chunk in halves lost as built (r06zzl: K1b's second launch holds the CUs' LDS, K3's first half only lands where it drains); K1b's second launch
# on a half / a quarter of its workgroups, so that K3's waves find room beside it
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzm && O=gpurun_out/r06zzm
export TMPDIR=/tmp
timeout 1500 python tests/tools/decode_variants_ab.py --frames 65536 --steps 10 --rounds 2 nohalves halvesb2 halvesb4 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-500 | tee $O/halves_bdiv_ab_65536.txt
