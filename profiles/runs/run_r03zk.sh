#!/bin/bash
# round 3, session ZK: rocprofv3 kernel trace of the several-block batch (2 048 x 1 MiB, both directions) on the final build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r03zk; P=/tmp/prof_blocks; rm -rf $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python tests/multiblock_rate.py 2048 1024 > gpurun_out/r03zk/multiblock_under_rocprof.json 2> gpurun_out/r03zk/err.log; echo "rc $?"
f=$(find $P -name "*kernel_stats.csv" | head -1); grep -h "zhip_\|^\"Name\"" "$f" > gpurun_out/r03zk/r03_blocks_2048x1MiB_kernel_stats.csv; cat gpurun_out/r03zk/r03_blocks_2048x1MiB_kernel_stats.csv | cut -c1-110
tail -1 gpurun_out/r03zk/multiblock_under_rocprof.json | cut -c1-200
