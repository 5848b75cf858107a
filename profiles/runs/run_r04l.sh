#!/bin/bash
# round 4, GPU session L: three compress chunk slots on consecutive streams (4 hardware queues), dictionary slots sized from the size hint (configs[3] in one launch)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04l && O=gpurun_out/r04l
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_compress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -4 $O/pytest_gpu.txt
timeout 600 python tests/host_api_rate.py 65536 > $O/host_api_65536.log 2>&1; echo "slots 3: $(tail -1 $O/host_api_65536.log | cut -c1-200)"
ZHIP_ESLOTS=2 timeout 600 python tests/host_api_rate.py 65536 > $O/host_api_65536_s2.log 2>&1; echo "slots 2: $(tail -1 $O/host_api_65536_s2.log | cut -c1-200)"
ZHIP_ESLOTS=1 timeout 600 python tests/host_api_rate.py 65536 > $O/host_api_65536_s1.log 2>&1; echo "slots 1: $(tail -1 $O/host_api_65536_s1.log | cut -c1-200)"
timeout 600 python bench.py --config dict --steps 5 --warmup 1 > $O/b_dict.json 2> $O/b_dict.err; python -c "
import json
l=json.loads(open('$O/b_dict.json').read().strip().splitlines()[-1]); d=l['decompress']; print('dict', l['value'], l['ms_per_step'], l['kernels'], 'decompress', d['value'], d['ms_per_step'])"
tail -3 $O/b_dict.err
