#!/bin/bash
# round 6, GPU session ZZX: how often is a table allocation which kind? 24 candidates per process (probe launches; whole launches for the first eight), four processes
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzx && O=gpurun_out/r06zzx
export TMPDIR=/tmp
export ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_pickstudy24.so
for i in 1 2 3 4; do echo "process $i" | tee -a $O/kinds.txt; timeout 600 python bench.py --config compress --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-host-api 2>&1 >/dev/null | grep pick-study | sed -e 's/; waves.*(min/ (min/' | cut -c1-200 | tee -a $O/kinds.txt; done
