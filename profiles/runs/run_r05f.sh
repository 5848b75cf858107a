#!/bin/bash
# round 5, GPU session F: the compact decode arenas (K1 / K2 claim literal / sequence room from per-chunk budgets: ~12 GiB of scratch per 65 536-frame
# chunk instead of 31) -- the whole GPU suite, then the decode step's kernel times and the context's scratch beside round 4's fixed-slot build
# (git tag r04-experiments is not built here: the numbers of r05b / r05d, same box class, are the comparison), then the host-API crossover table
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05f && O=gpurun_out/r05f
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 600 python tests/tools/decode_variants_ab.py --steps 10 --rounds 2 product 2>&1 | grep -v amdgpu.ids | tee $O/decode_compact_arenas.txt
timeout 900 python tests/crossover.py 2>&1 | grep -v amdgpu.ids | tee $O/crossover.txt
