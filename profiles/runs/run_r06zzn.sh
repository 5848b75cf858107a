#!/bin/bash
# round 6, GPU session ZZN: the host-API crossover table (tests/crossover.py: one multi_*_to_buffer call against libzstd on the host's threads, 3 item sizes x 2 directions x 7 batch sizes) on the round's last build -- INTEGRATION.md section 6's table dates from r05f
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzn && O=gpurun_out/r06zzn
export TMPDIR=/tmp
timeout 1500 python tests/crossover.py 2>&1 | grep -v amdgpu.ids | tee $O/crossover.jsonl | cut -c1-200 | tail -8
