#!/bin/bash
# round 4, GPU session ZE: small batches of sources of several blocks (2 048 x 1 MiB: the generic kernel's nested-loop search, 2.1 GB/s) through the
# flat several-block search instead -- a lane per source, 1 / 2 / 4 / 8 / 32 sources per wave, two and four probes per trip; 8 192 x 256 KiB the same
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04ze && O=gpurun_out/r04ze
export TMPDIR=/tmp
run() { name=$1; shift; n=$1; shift; k=$1; shift; env "$@" timeout 300 python tests/multiblock_rate.py $n $k > $O/mb_$name.log 2>&1; echo "$name: $(tail -1 $O/mb_$name.log | cut -c1-330)"; }
run 2048_generic 2048 1024 X=1
run 2048_l1 2048 1024 ZHIP_MBC_MIN=1 ZHIP_MBC_LANES=1
run 2048_l2 2048 1024 ZHIP_MBC_MIN=1 ZHIP_MBC_LANES=2
run 2048_l4 2048 1024 ZHIP_MBC_MIN=1 ZHIP_MBC_LANES=4
run 2048_l8 2048 1024 ZHIP_MBC_MIN=1 ZHIP_MBC_LANES=8
run 2048_l32 2048 1024 ZHIP_MBC_MIN=1 ZHIP_MBC_LANES=32
run 2048_l2_two 2048 1024 ZHIP_MBC_MIN=1 ZHIP_MBC_LANES=2 ZHIP_FLAT4_MAX=0
run 8192_default 8192 256 X=1
run 8192_l8 8192 256 ZHIP_MBC_MIN=1 ZHIP_MBC_LANES=8
run 8192_l32_two 8192 256 ZHIP_MBC_MIN=1 ZHIP_FLAT4_MAX=0
