#!/bin/bash
# round 5, GPU session S: the table placement pick on dictionary batches (configs[3]: 262 144 x 4 KiB, 12 GiB of tables) -- four fresh processes, every frame
# compared with libzstd's inside each run
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05s && O=gpurun_out/r05s
export TMPDIR=/tmp
for k in 1 2 3 4; do timeout 300 python bench.py --config dict --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); x=d['decompress']; print('dict: compress %.1f GB/s %.2f ms  pick %s  decompress %.1f GB/s %.2f ms' % (d['value'], d['ms_per_step'], d['table_pick'], x['value'], x['ms_per_step']), {k.replace('zhip_encode_','').replace('_kernel',''): v['avg_ms'] for k, v in d['kernels'].items()})"; done | tee $O/dict_pick.txt
ZHIP_E1F_PICK=0 timeout 300 python bench.py --config dict --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no pick: compress %.1f GB/s %.2f ms' % (d['value'], d['ms_per_step']))" | tee -a $O/dict_pick.txt
