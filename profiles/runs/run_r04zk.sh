#!/bin/bash
# round 4, GPU session ZK: the GPU suite and smoke() once more on the round's last commit (the several-block search's probe rule changed after r04zh)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zk && O=gpurun_out/r04zk
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
