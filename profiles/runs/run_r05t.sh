#!/bin/bash
# round 5, GPU session T (diagnostic): is the match kernel's placement regime a matter of WHERE INSIDE an allocation the tables start (then an offset could
# be chosen without a second allocation) or of the allocation's pages? The same second allocation timed at several offsets, three processes
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05t && O=gpurun_out/r05t
export TMPDIR=/tmp
for k in 1 2 3; do ZHIP_DIAG_PICK_OFFSETS=1,2,64,256,1024,2048,4096 timeout 300 python bench.py --config compress --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep zhip-diag; done | tee $O/table_offsets.txt
