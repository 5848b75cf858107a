#!/bin/bash
# round 5, GPU session ZB (diagnostic): the host-buffer API against the flags of its pinned STAGING buffers (the upload's copies run at 44.7 GB/s where plain pinned
# copies reach 57): default, write-combined (0x4), non-coherent (0x40000000), both
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05zb && O=gpurun_out/r05zb
export TMPDIR=/tmp
for f in default 0x4 default 0x4 default 0x4; do
  if [ $f = default ]; then timeout 300 python tests/tools/host_api_stage_flags.py 2>&1 | grep ZHIP_DIAG; else ZHIP_DIAG_STAGE_FLAGS=$f timeout 300 python tests/tools/host_api_stage_flags.py 2>&1 | grep -E "ZHIP_DIAG|rror" | tail -2; fi
done | tee $O/stage_flags_ab.txt
