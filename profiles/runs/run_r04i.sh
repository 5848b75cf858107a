#!/bin/bash
# round 4, GPU session I: host-API compress at 65 536 x 128 KiB against the number of packing threads (chunk slots on: 4 x 16 384)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04i && O=gpurun_out/r04i
export TMPDIR=/tmp
for t in 8 16 32 64; do ZHIP_PACK_THREADS=$t timeout 600 python tests/host_api_rate.py 65536 > $O/host_api_65536_pack$t.log 2>&1; echo "pack $t: $(tail -1 $O/host_api_65536_pack$t.log | cut -c1-200)"; done
ZHIP_PACK_THREADS=32 ZHIP_ESLOTS=1 timeout 600 python tests/host_api_rate.py 65536 > $O/host_api_65536_pack32_slots1.log 2>&1; echo "pack 32 slots 1: $(tail -1 $O/host_api_65536_pack32_slots1.log | cut -c1-200)"
ZHIP_PACK_THREADS=32 timeout 600 python tests/host_api_rate.py 8192 > $O/host_api_8192_pack32.log 2>&1; echo "8192 pack 32: $(tail -1 $O/host_api_8192_pack32.log | cut -c1-200)"
ZHIP_PACK_THREADS=32 ZHIP_ESLOT_ITEMS=2048 timeout 600 python tests/host_api_rate.py 8192 > $O/host_api_8192_pack32_i2048.log 2>&1; echo "8192 pack 32 items 2048: $(tail -1 $O/host_api_8192_pack32_i2048.log | cut -c1-200)"
