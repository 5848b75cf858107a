#!/bin/bash
# round 4, GPU session O: K2 with straight-line repeat-offset selects (316 -> 296 instructions per four steps; "repeat offset 1 minus one = 0" travels as offset 0, K3 checks offset - 1)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04o && O=gpurun_out/r04o
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra --no-host-api"
for i in 1 2; do timeout 300 $B > $O/b_new$i.json 2>> $O/b_new.err; done
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
")"; done
timeout 300 python bench.py --config dict --steps 5 --warmup 1 --no-cpu-baseline > $O/b_dict.json 2> $O/b_dict.err; python -c "
import json
l=json.loads(open('$O/b_dict.json').read().strip().splitlines()[-1]); d=l['decompress']; print('dict', l['value'], l['ms_per_step'], 'decompress', d['value'], d['ms_per_step'], d['kernels'])"
timeout 300 python bench.py --config blocks --steps 5 --warmup 1 --no-cpu-baseline > $O/b_blocks.json 2> $O/b_blocks.err; python -c "
import json
l=json.loads(open('$O/b_blocks.json').read().strip().splitlines()[-1]); print('blocks', l['value'], l['ms_per_step'], l['kernels'])"
