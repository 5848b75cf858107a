#!/bin/bash
# round 6, GPU session ZZF: parity stress on the round's last build -- the new large-mixed-batch decode driver (tests/stress_gpu_decode.py: K0 / KX / the side stream
# over whole, damaged, checksummed and short frames in one launch), then the compress and several-block drivers with fresh seeds
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzf && O=gpurun_out/r06zzf
export TMPDIR=/tmp
for s in 6201 6202 6203; do timeout 900 python tests/stress_gpu_decode.py $s 2>&1 | tail -2 | cut -c1-1200 | tee -a $O/stress_gpu_decode.txt; done
timeout 900 python tests/stress_gpu_decode.py 6204 20000 2>&1 | tail -2 | cut -c1-1200 | tee -a $O/stress_gpu_decode.txt
timeout 900 python tests/stress_gpu_compress.py 6205 2>&1 | tail -2 | cut -c1-600 | tee $O/stress_gpu_compress.txt
timeout 900 python tests/stress_gpu_blocks.py 6206 2>&1 | tail -2 | cut -c1-600 | tee $O/stress_gpu_blocks.txt
