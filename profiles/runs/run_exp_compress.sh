#!/bin/sh
# GPU experiment: entropy-kernel phase timers at the BASELINE size
cd /root/repo
mkdir -p gpurun_out
ZHIP_PROF=1 timeout 300 python bench.py --direction compress --frames 65536 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/exp5_c64k.json 2> gpurun_out/exp5_c64k.err
grep zhip-prof gpurun_out/exp5_c64k.err; cat gpurun_out/exp5_c64k.json | cut -c100-160,560-900
