#!/bin/sh
# GPU experiment: compress direction with the flat match kernel, lane-count variants
cd /root/repo
mkdir -p gpurun_out
export ZHIP_WATCHDOG=1
( timeout 600 python -m pytest tests/test_gpu_compress.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/exp_pytest.log 2>&1
for v in "" _fl32 _fl16; do
  ZHIP_LIB=/root/repo/python-zstandard_amd/csrc/libzstd_hip$v.so timeout 400 python bench.py --direction compress --frames 16384 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp_c16k$v.json 2> gpurun_out/exp_c16k$v.err
done
timeout 500 python bench.py --direction compress --frames 65536 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/exp_c64k.json 2> gpurun_out/exp_c64k.err
tail -3 gpurun_out/exp_pytest.log; cat gpurun_out/exp_c16k*.json gpurun_out/exp_c64k.json | cut -c1-1500; tail -3 gpurun_out/exp_c*.err
