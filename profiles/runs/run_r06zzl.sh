#!/bin/bash
# round 6, GPU session ZZL: the chunk in halves -- K1b and K3 in two launches each, K3's first half beside K1b's second (zhip_decompress_batch_device) -- against the build
# without (-DZHIP_HALVES=0): decode-side GPU tests on the product, then three alternating rounds at 65 536 frames, one at 131 072 and one at 32 768 / 16 384
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzl && O=gpurun_out/r06zzl
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-300 | tee $O/pytest_decode.txt
timeout 900 python tests/stress_gpu_decode.py 6208 20000 2>&1 | tail -1 | cut -c1-900 | tee $O/stress_gpu_decode.txt
timeout 1500 python tests/tools/decode_variants_ab.py --frames 65536 --steps 10 --rounds 3 nohalves product 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-500 | tee $O/halves_ab_65536.txt
timeout 900 python tests/tools/decode_variants_ab.py --frames 131072 --steps 5 --rounds 1 nohalves product 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-500 | tee $O/halves_ab_131072.txt
timeout 900 python tests/tools/decode_variants_ab.py --frames 32768 --steps 10 --rounds 1 nohalves product 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-500 | tee $O/halves_ab_32768.txt
timeout 900 python tests/tools/decode_variants_ab.py --frames 16384 --steps 10 --rounds 1 nohalves product 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-500 | tee $O/halves_ab_16384.txt
