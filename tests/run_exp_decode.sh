#!/bin/sh
# GPU experiment: decode direction at the BASELINE size, then the phase timers
cd /root/repo
mkdir -p gpurun_out
timeout 300 python bench.py --compress-frames 0 --no-cpu-baseline > gpurun_out/exp_d64k.json 2> gpurun_out/exp_d64k.err
cat gpurun_out/exp_d64k.json | cut -c80-140,560-1100; tail -n 2 gpurun_out/exp_d64k.err
ZHIP_PROF=1 timeout 300 python bench.py --compress-frames 0 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/exp_dprof.json 2> gpurun_out/exp_dprof.err
grep zhip-prof gpurun_out/exp_dprof.err | grep -A6 "K3:" | tail -7
