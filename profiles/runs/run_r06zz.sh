#!/bin/bash
# round 6, GPU session ZZ: frames cut by libzstd's block splitter (level 19: several blocks in 128 KiB) through the host-buffer call -- block counts from the headers, the chunk to the
# several-block mode -- the new GPU test, the decode tests, and the rate of level-3 / level-19 frames through the Python API (r06zy: 2 048 level-19 frames 44 ms device-resident in the generic kernel)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zz && O=gpurun_out/r06zz
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_decompress.py tests/test_cext_backend.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_decode.txt
timeout 900 python tests/tools/split_block_frames_rate.py 2048 2>/dev/null | tail -1 | tee $O/split_block_frames.txt
timeout 900 python tests/tools/split_block_frames_rate.py 8192 2>/dev/null | tail -1 | tee -a $O/split_block_frames.txt
