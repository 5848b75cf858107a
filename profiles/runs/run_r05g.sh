#!/bin/bash
# round 5, GPU session G: what K2's and K1b's STORES cost their chains (vmcnt is in order on gfx9: the ring refill's wait also sits out older stores;
# compacting K2's destination took it from 10.0 to 8.5 ms, r05f). Diagnostic builds without the stores (wrong bytes downstream) and builds
# with non-temporal stores, one set of frames, one process per build; then: can E1f's fast placement be picked (tests/tools/e1f_pick_best.py)?
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05g && O=gpurun_out/r05g
export TMPDIR=/tmp
timeout 900 python tests/tools/decode_variants_ab.py --steps 10 --rounds 2 product zqnost zqnt hufnost hufnt 2>&1 | grep -v amdgpu.ids | tee $O/k2_k1b_stores.txt
timeout 600 python tests/tools/e1f_pick_best.py 4 2>&1 | grep -v amdgpu.ids | tee $O/e1f_pick_best.txt
