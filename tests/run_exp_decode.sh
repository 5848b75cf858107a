#!/bin/sh
# GPU experiment: decode direction at the BASELINE size
cd /root/repo
mkdir -p gpurun_out
timeout 300 python bench.py --compress-frames 0 --no-cpu-baseline > gpurun_out/exp_d64k.json 2> gpurun_out/exp_d64k.err
cat gpurun_out/exp_d64k.json | cut -c80-140,560-1000; tail -n 2 gpurun_out/exp_d64k.err
