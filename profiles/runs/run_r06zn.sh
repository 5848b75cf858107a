#!/bin/bash
# round 6, GPU session ZN: K1b at 10 / 12 frames per workgroup (four workgroups per CU, one wave per SIMD: 40 / 48 frames per CU in flight; 34.8 / 40.96 KiB of LDS each -- the
# second does not fit the 40.93 KiB one leaving K2 wave frees) against the product's 8 (six per CU, 26.6 KiB), K1b beside K2 as in the product
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zn && O=gpurun_out/r06zn
export TMPDIR=/tmp
D="python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline"
for i in 1 2; do for V in product huf10 huf12; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 $D --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$V', d['value'], d['ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in k.items()})" | tee -a $O/k1b_frames_per_wg.txt
done; done
