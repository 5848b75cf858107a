#!/bin/bash
# round 6, GPU session Y: K0 in the several-block mode too (a record per block of a frame's first blocks), E2's chains eight steps per trip -- the whole GPU suite on that build,
# frames of several blocks with / without K0 (-DZHIP_K0=0), the GPU stress drivers
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06y && O=gpurun_out/r06y
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
for i in 1 2 3; do for V in nok0 product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python tests/multiblock_rate.py 2048 1024 2>/dev/null | tail -2 | tr '\n' ' ' | sed "s/^/$V 2048x1MiB /" | cut -c1-700 | tee -a $O/blocks_k0_ab.txt; echo | tee -a $O/blocks_k0_ab.txt
done; done
for V in nok0 product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python tests/multiblock_rate.py 8192 256 2>/dev/null | tail -2 | tr '\n' ' ' | sed "s/^/$V 8192x256KiB /" | cut -c1-700 | tee -a $O/blocks_k0_ab.txt; echo | tee -a $O/blocks_k0_ab.txt
done
timeout 900 python tests/stress_gpu_blocks.py 2>&1 | tail -3 | tee $O/stress_gpu_blocks.txt
timeout 900 python tests/stress_gpu_compress.py 2>&1 | tail -3 | tee $O/stress_gpu_compress.txt
