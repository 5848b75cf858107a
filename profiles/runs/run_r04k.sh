#!/bin/bash
# round 4, GPU session K: the chunk slots with more hardware queues (r04j: HIP's default 4 queues made slot 3's kernels wait behind slot 0's)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04k && O=gpurun_out/r04k
export TMPDIR=/tmp
for q in 8 16; do GPU_MAX_HW_QUEUES=$q timeout 600 python tests/host_api_rate.py 65536 > $O/host_api_65536_q$q.log 2>&1; echo "queues $q: $(tail -1 $O/host_api_65536_q$q.log | cut -c1-200)"; done
GPU_MAX_HW_QUEUES=8 ZHIP_ESLOT_ITEMS=8192 timeout 600 python tests/host_api_rate.py 65536 > $O/host_api_65536_q8_i8192.log 2>&1; echo "queues 8 items 8192: $(tail -1 $O/host_api_65536_q8_i8192.log | cut -c1-200)"
GPU_MAX_HW_QUEUES=8 ZHIP_ESLOT_ITEMS=2048 timeout 600 python tests/host_api_rate.py 8192 > $O/host_api_8192_q8_i2048.log 2>&1; echo "8192 queues 8 items 2048: $(tail -1 $O/host_api_8192_q8_i2048.log | cut -c1-200)"
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra --no-host-api > $O/b_q8.json 2>$O/b_q8.err; python -c "
import json; l=json.loads(open('$O/b_q8.json').read().strip().splitlines()[-1]); print('decode with 8 queues', l['value'], l['ms_per_step'])"
