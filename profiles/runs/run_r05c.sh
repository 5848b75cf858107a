#!/bin/bash
# round 5, GPU session C: K3's memory floor against occupancy and against an LDS window over the recent output (tests/ubench/k3_floor_diag.hpp,
# -DZP_FLOOR_WIN): would serving the far matches whose source lies a few KiB behind the batch from LDS lower the floor by more than the
# occupancy its buffer costs? All builds in one process on one set of frames (tests/tools/decode_variants_ab.py), two rounds.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05c && O=gpurun_out/r05c
export TMPDIR=/tmp
timeout 600 python tests/tools/decode_variants_ab.py --steps 5 --rounds 1 product floor floorw4 floorw3 floorw2 floorw6free floorw4free floorw3free floorw2free 2>&1 | grep -v amdgpu.ids | tee $O/k3_floor_window2.txt
