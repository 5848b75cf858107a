#!/bin/bash
# round 6, GPU session ZZZ2: smoke and the whole GPU suite on the last commit's library (the wave-clock hook behind zd_wall_clock)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzz2 && O=gpurun_out/r06zzz2
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | cut -c1-300 | tee $O/pytest_gpu.txt
