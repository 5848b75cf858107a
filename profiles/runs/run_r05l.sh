#!/bin/bash
# round 5, GPU session L: the one arena with literal rooms growing down and sequence rooms growing up (each kind packed) -- GPU suite, decode step + scratch,
# frames of several blocks
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05l && O=gpurun_out/r05l
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 600 python tests/tools/decode_variants_ab.py --steps 10 --rounds 2 product 2>&1 | grep -v amdgpu.ids | tee $O/decode_one_arena_two_ends.txt
timeout 600 python tests/multiblock_rate.py 2048 1024 2>/dev/null | tail -1 | tee $O/multiblock_2048x1MiB.json
timeout 600 python tests/tools/decode_entropy_shapes.py 2>&1 | grep -v amdgpu.ids | tee $O/decode_entropy_shapes.txt
