#!/bin/sh
# GPU experiment: compress direction, entropy-kernel occupancy variants at the BASELINE size + a light-load point
cd /root/repo
mkdir -p gpurun_out
export ZHIP_WATCHDOG=1
for v in "" _e2w4; do
  ZHIP_LIB=/root/repo/python-zstandard_amd/csrc/libzstd_hip$v.so timeout 300 python bench.py --direction compress --frames 65536 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp3_c64k$v.json 2> gpurun_out/exp3_c64k$v.err
done
timeout 300 python bench.py --direction compress --frames 16384 --steps 2 --warmup 1 > gpurun_out/exp3_c16k.json 2> gpurun_out/exp3_c16k.err
cat gpurun_out/exp3_c*.json | cut -c100-160,560-1500; tail -n 3 gpurun_out/exp3_c*.err
