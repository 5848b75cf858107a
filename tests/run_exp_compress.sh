#!/bin/sh
# GPU experiment: compress direction with the flat match kernel, lane-count variants at the BASELINE size
cd /root/repo
mkdir -p gpurun_out
export ZHIP_WATCHDOG=1
for v in "" _fl32 _fl16; do
  ZHIP_LIB=/root/repo/python-zstandard_amd/csrc/libzstd_hip$v.so timeout 300 python bench.py --direction compress --frames ${EXP_FRAMES:-65536} --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp2_c64k$v.json 2> gpurun_out/exp2_c64k$v.err
done
cat gpurun_out/exp2_c64k*.json | cut -c1-120,560-900; tail -n 3 gpurun_out/exp2_c*.err
