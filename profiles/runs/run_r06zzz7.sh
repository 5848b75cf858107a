#!/bin/bash
# round 6, GPU session ZZZ7: the pick's candidates kept inside the memory that was free when it began: the first call again (65 536 and 131 072 sources), candidate by candidate
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzz7 && O=gpurun_out/r06zzz7
export TMPDIR=/tmp
for n in 65536 131072 32768; do ZHIP_PROF=1 timeout 600 python tests/tools/first_call_cost.py $n 2>&1 | grep "pick candidate\|sources" | cut -c1-400 | tee -a $O/pick_candidates_cost.txt; done
