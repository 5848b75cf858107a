#!/bin/bash
# round 5, GPU session R: the decode arena's budget follows the caller's size hint (4 KiB documents: 17 KiB of room each instead of 160) and small frames get a
# third chunk slot -- the dictionary / boundary GPU tests, configs[3] twice, the decode step's scratch at both sizes
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05r && O=gpurun_out/r05r
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
for k in 1 2; do timeout 300 python bench.py --config dict --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); x=d['decompress']; print('dict: decompress %.1f GB/s %.2f ms  compress %.1f GB/s %.2f ms' % (x['value'], x['ms_per_step'], d['value'], d['ms_per_step']), {k.replace('zhip_decode_','').replace('_kernel',''): v['avg_ms'] for k, v in x['kernels'].items()})"; done | tee $O/dict.txt
timeout 600 python tests/tools/decode_variants_ab.py --steps 10 --rounds 1 product 2>&1 | grep -v amdgpu.ids | tee $O/decode.txt
