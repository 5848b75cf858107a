#!/bin/bash
# (this session's tools -- tests/tools/e1f_window_sweep.py, the ZHIP_E1F_WIN knob, the windowed kernels -- are in git tag r06-e1f-window: the form was measured and removed)
# round 6, GPU session C: (1) where a trip of the flat match kernel spends its cycles (-DZE_PROF_FLAT: s_memtime deltas per phase, lane 0 of every wave), the form
# of rounds 1-5 against the LDS-window form, at 8 192 and 65 536 sources per launch; (2) the multi-device split of the host-buffer API on two / three
# device slots of the one GPU (ZHIP_DEVICES=0,0): byte-identical collections, first failing item
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06c && O=gpurun_out/r06c
export TMPDIR=/tmp ZHIP_E1F_PICK=0 ZHIP_E1LDS_MAX=0
for F in 8192 65536; do for W in 0 1; do
  echo "== $F sources, ZHIP_E1F_WIN=$W" | tee -a $O/flat_phases.txt
  ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_flatprof.so ZHIP_PROF=1 ZHIP_E1F_WIN=$W timeout 300 python bench.py --config compress --frames $F --no-cpu-baseline --steps 1 --warmup 1 2>&1 >/dev/null | grep "zhip-prof" | grep -A8 "flat search" | tail -8 | tee -a $O/flat_phases.txt
done; done
timeout 1200 python -m pytest tests/test_gpu_multidevice.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_multidevice.txt
