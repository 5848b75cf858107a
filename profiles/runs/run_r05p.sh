#!/bin/bash
# round 5, GPU session P: the table pick's GPU test (49 152 small sources through a device context, every 97th frame against libzstd), and the host-buffer
# decompress call against the size of its pipeline's first chunk (ZHIP_HCHUNK_D0), beside what the link gives for plain pinned copies
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05p && O=gpurun_out/r05p
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_compress.py -x -q -m gpu -k "table_placement or mixed_small" 2>&1 | tail -4 | tee $O/pytest_pick.txt
timeout 600 python tests/tools/host_decompress_first_chunk.py 2>&1 | grep -v amdgpu.ids | tee $O/host_decompress_first_chunk.txt
