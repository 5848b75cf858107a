#!/bin/bash
# round 4, GPU session ZG: 65 536 sources per launch with two / three / four probes per trip, four rounds in one process (placement regimes: compare within a round)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zg && O=gpurun_out/r04zg
export TMPDIR=/tmp
timeout 600 python tests/tools/e1f_alloc_trials.py 4 ZHIP_FLAT3=0,ZHIP_FLAT3=1,ZHIP_FLAT4_MAX=65536 2>&1 | grep -v amdgpu.ids | tee $O/trials_probes_65536.txt | tail -14
