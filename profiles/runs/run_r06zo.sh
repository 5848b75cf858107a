#!/bin/bash
# round 6, GPU session ZO: K1b at 12 frames per workgroup with its ring trimmed to the lanes in use (40 896 bytes: fits what ONE leaving K2 wave frees) as the product, against 8
# (-DZP_HUF_FRAMES=8, the shape of rounds 2-5): decode tests, then the decode line alternating, the several-block frames, one 128 KiB frame
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zo && O=gpurun_out/r06zo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2 | tee $O/pytest_decode.txt
D="python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline"
for i in 1 2 3; do for V in huf8 product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 $D --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$V', d['value'], d['ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in k.items()})" | tee -a $O/k1b_12_frames_ab.txt
done; done
for V in huf8 product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python bench.py --config blocks --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V blocks', d.get('value'), d.get('ms_per_step'))" | tee -a $O/k1b_12_frames_ab.txt
  ZHIP_LIB=$L timeout 600 python bench.py --config dict --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V dict decompress', d['decompress']['value'], d['decompress']['ms_per_step'])" | tee -a $O/k1b_12_frames_ab.txt
  ZHIP_LIB=$L timeout 300 python tests/tools/decode_batch_sizes.py 1 64 2048 8192 32768 2>/dev/null | tail -1 | sed "s/^/$V by batch size /" | tee -a $O/k1b_12_frames_ab.txt
done
