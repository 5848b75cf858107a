#!/bin/bash
# round 6, GPU session L: the reference's own hot-path tests through the shim -- on one device and with the batch calls cut over two device slots (ZHIP_DEVICES=0,0) --
# and the GPU stress drivers (compress, several blocks) on this round's build
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06l && O=gpurun_out/r06l
export TMPDIR=/tmp
sh tests/run_reference_hotpath_tests.sh > /dev/null 2>&1; cp gpurun_out/reference_hotpath_tests.log $O/reference_hotpath_tests.log; tail -3 $O/reference_hotpath_tests.log
ZHIP_DEVICES=0,0 sh tests/run_reference_hotpath_tests.sh > /dev/null 2>&1; cp gpurun_out/reference_hotpath_tests.log $O/reference_hotpath_tests_two_device_slots.log; tail -3 $O/reference_hotpath_tests_two_device_slots.log
for seed in 61 62; do timeout 900 python tests/stress_gpu_compress.py $seed 2>&1 | tail -2; done | tee $O/stress_gpu_compress.txt
timeout 900 python tests/stress_gpu_blocks.py 63 2>&1 | tail -3 | tee $O/stress_gpu_blocks.txt
ZHIP_DEVICES=0,0 timeout 900 python tests/stress_gpu_compress.py 64 2>&1 | tail -2 | tee $O/stress_gpu_compress_two_device_slots.txt
