#!/bin/bash
# round 4, GPU session W: E1f's two regimes, second experiment -- is it RECYCLED VRAM? (a) bench.py's default order with the decode context kept
# alive while the compress context allocates (tables from VRAM the decode arenas never used); (b) the compress-only run after allocating and
# releasing VRAM the size of the decode arenas; (c) both plain, as controls on this box.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04w && O=gpurun_out/r04w
export TMPDIR=/tmp
show() { python - <<P
import json
try:
    l = json.loads(open('$O/b_$1.json').read().strip().splitlines()[-1])
    c = l.get('compress', l)
    print('$1', 'compress', c['value'], c['ms_per_step'], {k.replace('zhip_encode_','').replace('_kernel',''): v['avg_ms'] for k, v in c['kernels'].items()})
except Exception as e: print('$1', 'ERR', e, open('$O/b_$1.err').read()[-600:])
P
}
D="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-host-api"
C="python bench.py --config compress --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-host-api"
ZHIP_BENCH_KEEP_DECODE_CTX=1 timeout 300 $D > $O/b_default_keep.json 2> $O/b_default_keep.err; show default_keep
ZHIP_BENCH_PREFRAG=9,24,3,1 timeout 300 $C > $O/b_alone_prefrag.json 2> $O/b_alone_prefrag.err; show alone_prefrag
timeout 300 $C > $O/b_alone.json 2> $O/b_alone.err; show alone
timeout 300 $D > $O/b_default.json 2> $O/b_default.err; show default
