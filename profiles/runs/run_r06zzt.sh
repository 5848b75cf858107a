#!/bin/bash
# round 6, GPU session ZZT: with probes at ~17 ms a candidate the host-buffer API's contexts (chunks of 32 768 sources, tables kept between calls) can afford the placement pick too: the pick from 16 384 sources
# per launch on (-DZHIP_PICK_MIN=16384) against the product (49 152), multi_compress_to_buffer of 65 536 x 128 KiB through Python, five processes each, alternating
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzt && O=gpurun_out/r06zzt
export TMPDIR=/tmp
for i in 1 2 3 4 5; do for V in product pick16k; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python tests/host_api_rate.py 65536 2>/dev/null | tail -1 | python -c "
import sys, json; l = sys.stdin.read(); d = json.loads(l[l.index('{'):]); print('$V', 'compress', d['compress_GBps'], 'decompress', d['decompress_GBps'])" | tee -a $O/host_api_pick.txt
done; done
