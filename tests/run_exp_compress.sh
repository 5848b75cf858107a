#!/bin/sh
# GPU experiment: compress direction at the BASELINE size (no watchdog: it would quantise the timing)
cd /root/repo
mkdir -p gpurun_out
timeout 300 python bench.py --direction compress --frames 65536 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/exp4_c64k.json 2> gpurun_out/exp4_c64k.err
cat gpurun_out/exp4_c64k.json | cut -c100-160,560-1100; tail -n 3 gpurun_out/exp4_c64k.err
