#!/bin/bash
# round 5, GPU session K: ONE compact arena for literals and sequences in both modes of the decode pipeline (160 KiB per frame / item on average) --
# the whole GPU suite, the decode step + scratch, batches of unusual entropy shapes (does the common budget hold?), frames of several blocks
# (2 048 x 1 MiB, 8 192 x 256 KiB), the small-batch latencies
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05k && O=gpurun_out/r05k
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 600 python tests/tools/decode_variants_ab.py --steps 10 --rounds 2 product 2>&1 | grep -v amdgpu.ids | tee $O/decode_one_arena.txt
timeout 600 python tests/tools/decode_entropy_shapes.py 2>&1 | grep -v amdgpu.ids | tee $O/decode_entropy_shapes.txt
timeout 600 python tests/multiblock_rate.py 2048 1024 2>/dev/null | tail -1 | tee $O/multiblock_2048x1MiB.json
timeout 600 python tests/multiblock_rate.py 8192 256 2>/dev/null | tail -1 | tee $O/multiblock_8192x256KiB.json
timeout 600 python tests/small_batch_latency.py 2>/dev/null | tail -1 | tee $O/small_batch_latency.json
