#!/bin/bash
# round 6, GPU session ZZZ6: the first call's cost, candidate by candidate (ZHIP_PROF=1 prints reserve / zero + probe times of the pick's candidates)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzz6 && O=gpurun_out/r06zzz6
export TMPDIR=/tmp
ZHIP_PROF=1 timeout 600 python tests/tools/first_call_cost.py 65536 2>&1 | grep "pick candidate\|sources" | cut -c1-400 | tee $O/pick_candidates_cost.txt
