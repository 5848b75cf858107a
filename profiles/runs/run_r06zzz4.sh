#!/bin/bash
# round 6, GPU session ZZZ4: what the first large compress call of a context costs with the eight-candidate pick (tests/tools/first_call_cost.py), 65 536 and 32 768 sources of 128 KiB
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzz4 && O=gpurun_out/r06zzz4
export TMPDIR=/tmp
for n in 65536 32768; do timeout 600 python tests/tools/first_call_cost.py $n 2>/dev/null | tail -1 | tee -a $O/first_call_cost.txt; done
