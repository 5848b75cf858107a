#!/bin/bash
# round 5, GPU session Y: the headline decode step against chunk size / slot streams once more (round 2 found one chunk of 65 536 best; the kernels' balance has changed since)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05y && O=gpurun_out/r05y
export TMPDIR=/tmp
for cfg in "65536 2" "32768 2" "32768 3" "16384 3" "21846 3"; do set -- $cfg; echo "ZHIP_DCHUNK=$1 ZHIP_NSLOT=$2"; ZHIP_DCHUNK=$1 ZHIP_NSLOT=$2 timeout 300 python tests/tools/decode_variants_ab.py --steps 5 --rounds 1 product 2>&1 | grep -v amdgpu.ids; done | tee $O/decode_chunks.txt
