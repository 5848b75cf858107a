#!/bin/bash
# round 5, GPU session A: (1) the compress-side GPU tests on the slim encode arena (sequence area + 512 bytes per source when every row is double-fast),
# (2) the match kernel's rate against the frames in flight, 65 536 ... 262 144 sources in ONE launch
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05a && O=gpurun_out/r05a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_compress.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_compress.txt
timeout 900 python tests/tools/e1f_frames_in_flight.py 65536,131072,196608,262144 2>&1 | grep -v amdgpu.ids | tee $O/e1f_frames_in_flight.txt
