#!/bin/bash
# round 6, GPU session ZZQ: more parity stress on the final sources -- decode (large mixed batches, five seeds), compress and several-block drivers on one device slot and on two (ZHIP_DEVICES=0,0: the in-call fan-out)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzq && O=gpurun_out/r06zzq
export TMPDIR=/tmp
for s in 6401 6402 6403 6404 6405; do timeout 600 python tests/stress_gpu_decode.py $s 12000 2>&1 | tail -1 | cut -c1-900 | tee -a $O/stress.txt; done
for s in 6411 6412; do timeout 600 python tests/stress_gpu_compress.py $s 2>&1 | tail -1 | cut -c1-400 | tee -a $O/stress.txt; timeout 600 python tests/stress_gpu_blocks.py $s 2>&1 | tail -1 | cut -c1-400 | tee -a $O/stress.txt; done
export ZHIP_DEVICES=0,0 ZHIP_DEVICE_MIN_BYTES=0
for s in 6421; do timeout 600 python tests/stress_gpu_compress.py $s 2>&1 | tail -1 | cut -c1-400 | sed 's/^/two slots: /' | tee -a $O/stress.txt; timeout 600 python tests/stress_gpu_blocks.py $s 2>&1 | tail -1 | cut -c1-400 | sed 's/^/two slots: /' | tee -a $O/stress.txt; done
