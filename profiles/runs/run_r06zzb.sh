#!/bin/bash
# round 6, GPU session ZZB: fast-strategy sources leave the lane-serial match kernel as sequences only (the entropy kernel gathers the literals wave-parallel; the lane copied them
# byte by byte), tables zeroed 16 bytes a store: compress-side tests, then the levels' rates at 16 384 and 65 536 sources (before: r06zv / r06zw)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzb && O=gpurun_out/r06zzb
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_compress.py tests/test_gpu_boundary.py tests/test_cext_backend.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_compress.txt
timeout 900 python tests/stress_gpu_compress.py 2>&1 | tail -1 | tee -a $O/pytest_compress.txt
timeout 1500 python tests/tools/compress_levels_rate.py 16384 2>&1 | tail -1 | tee $O/compress_levels.txt
timeout 1500 python tests/tools/compress_levels_rate.py 65536 2>&1 | tail -1 | tee -a $O/compress_levels.txt
