#!/bin/bash
# round 6, GPU session ZZR: would a probe over the sources' FIRST BYTES rank the match kernel's table allocations like a whole launch does? (the pick times whole launches: 0.4-0.47 s a candidate, so it
# looks at three.) Diagnostic build -DZHIP_PICK_STUDY=1: eight candidate allocations per process, each timed whole and capped at 32 / 16 / 8 KiB per source; four compress-only processes and two
# in bench.py's default order (decode first)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzr && O=gpurun_out/r06zzr
export TMPDIR=/tmp
export ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_pickstudy.so
for i in 1 2 3 4; do echo "process $i (compress only)" | tee -a $O/pick_study.txt; timeout 600 python bench.py --config compress --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-host-api 2>&1 >/dev/null | grep pick-study | tee -a $O/pick_study.txt; done
for i in 5 6; do echo "process $i (decode leg first)" | tee -a $O/pick_study.txt; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-host-api 2>&1 >/dev/null | grep pick-study | tee -a $O/pick_study.txt; done
