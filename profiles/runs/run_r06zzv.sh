#!/bin/bash
# round 6, GPU session ZZV: is a slow table allocation slow EVERYWHERE? The diagnostic build's probe launches leave every wave's duration (a wave = 64 consecutive sources = 24 MiB of the allocation);
# printed as the mean per sixteenth of the allocation, eight candidates per process, three processes
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzv && O=gpurun_out/r06zzv
export TMPDIR=/tmp
export ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_pickstudy.so
for i in 1 2 3; do echo "process $i" | tee -a $O/wave_clocks.txt; timeout 600 python bench.py --config compress --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-host-api 2>&1 >/dev/null | grep pick-study | tee -a $O/wave_clocks.txt; done
