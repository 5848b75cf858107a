#!/bin/bash
# round 5, GPU session Z: K1's / K3's / E2's phase timers (ZHIP_PROF=1) on the round's last build
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05z && O=gpurun_out/r05z
export TMPDIR=/tmp
ZHIP_PROF=1 timeout 300 python bench.py --no-extra --no-host-api --no-cpu-baseline --steps 2 --warmup 1 --compress-frames 65536 2>&1 >/dev/null | grep zhip-prof | tail -60 | tee $O/phase_timers.txt
