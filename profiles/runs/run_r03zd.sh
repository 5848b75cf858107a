#!/bin/bash
# round 3, session ZD: a several-block batch of four chunks (arena slots reused) in both directions, every frame verified
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03zd && O=gpurun_out/r03zd
export TMPDIR=/tmp
timeout 900 python tests/multiblock_rate.py 32768 256 > $O/multiblock_32768x256KiB.txt 2>&1; tail -2 $O/multiblock_32768x256KiB.txt | cut -c1-900
