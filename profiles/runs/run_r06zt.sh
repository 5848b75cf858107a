#!/bin/bash
# round 6, GPU session ZT: KX with a plain store for its "a frame carries a checksum" word (65 536 atomics on one address cost K1 0.5 ms), EX -- checksum trailers of write_checksum
# frames by a lane per frame after the entropy kernel -- against the build that hashes inside that kernel (-DZHIP_TRAILER_LATER=0); the GPU tests of both directions
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zt && O=gpurun_out/r06zt
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout 900 python tests/tools/decode_checksum_cost.py 65536 2>/dev/null | tail -1 | tee $O/checksum_cost.txt
for V in notrailer product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 900 python tests/tools/compress_checksum_cost.py 32768 2>/dev/null | tail -1 | sed "s/^/$V compress /" | tee -a $O/checksum_cost.txt
done
