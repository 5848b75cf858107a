#!/bin/bash
# round 6, GPU session M: the host-buffer calls at 65 536 x 128 KiB on one device slot and on two slots of the one GPU (ZHIP_DEVICES=0,0: two host threads, two contexts,
# the batch cut in halves) -- what the fan-out costs where it cannot help
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06m && O=gpurun_out/r06m
export TMPDIR=/tmp
timeout 900 python tests/host_api_rate.py 65536 2>/dev/null | tail -1 | cut -c1-260 | sed 's/^/one slot   /' | tee -a $O/host_api_slots.txt
ZHIP_DEVICES=0,0 timeout 900 python tests/host_api_rate.py 65536 2>/dev/null | tail -1 | cut -c1-260 | sed 's/^/two slots  /' | tee -a $O/host_api_slots.txt
ZHIP_DEVICES=0,0 timeout 900 python tests/host_api_rate.py 8192 2>/dev/null | tail -1 | cut -c1-260 | sed 's/^/two slots  /' | tee -a $O/host_api_slots.txt
