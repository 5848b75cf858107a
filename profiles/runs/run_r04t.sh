#!/bin/bash
# round 4, GPU session T: the double-fast search in its link form (ZHIP_E1LINKS=1: records by the plain lane-per-frame pre-pass, then the
# per-lane state machine that follows them) -- parity through the GPU compress tests, then the search kernel's time at 64 / 32 / 16 frames per wave
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04t && O=gpurun_out/r04t
export TMPDIR=/tmp
( time ZHIP_E1LINKS=1 ZHIP_E1LDS_MAX=0 timeout 900 python -m pytest tests/test_gpu_compress.py -m gpu -x -q > $O/pytest_gpu_links.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu_links.txt
B="python bench.py --config compress --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-host-api"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.json 2>> $O/b_$name.err; python - <<P
import json
try:
    l = json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1])
    print('$name', l['value'], l['ms_per_step'], l.get('bit_exact_vs_libzstd'), {k.replace('zhip_encode_','').replace('_kernel',''): v['avg_ms'] for k, v in l['kernels'].items()})
except Exception as e: print('$name', 'ERR', e, open('$O/b_$name.err').read()[-600:])
P
}
run base X=1
run links64 ZHIP_E1LINKS=1 ZHIP_E1LINK_LANES=64
run links32 ZHIP_E1LINKS=1 ZHIP_E1LINK_LANES=32
run links16 ZHIP_E1LINKS=1 ZHIP_E1LINK_LANES=16
