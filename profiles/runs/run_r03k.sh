#!/bin/bash
# round 3, GPU session K: dictionary frames whose tables are all "repeat" read the dictionary's tables where they lie (no per-frame table slots)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03k && O=gpurun_out/r03k
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_decompress.py tests/test_gpu_compress.py -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 300 python bench.py --config dict --steps 5 --warmup 1 --no-cpu-baseline > $O/d_shared.json 2> $O/d.err
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra > $O/b_default.json 2>> $O/d.err
for f in $O/d_*.json; do echo "$(basename $f): $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); d=l['decompress']; print(l['value'], d['value'], d['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('ERR', e)
")"; done
python -c "
import json
l=json.loads(open('$O/b_default.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})"
