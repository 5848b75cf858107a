#!/bin/bash
# round 5, GPU session V: the new GPU test of an understated size hint (the arena's budget overrun: frames come back from the generic kernel)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05v && O=gpurun_out/r05v
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decompress.py -x -q -m gpu -k "understated or dictionary_frames" 2>&1 | tail -6 | tee $O/pytest.txt
