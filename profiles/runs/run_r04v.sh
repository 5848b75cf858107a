#!/bin/bash
# round 4, GPU session V: E1f's two regimes. Hypothesis: the 25 GiB of hash tables are random-access memory, and what
# they cost depends on how the driver backs them -- fresh VRAM (a process that compresses first) against VRAM recycled from the decode
# direction's arenas (bench.py's default order). Same box, same process order, tables asked for as physically contiguous memory or not.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04v && O=gpurun_out/r04v
export TMPDIR=/tmp
show() { python - <<P
import json
try:
    l = json.loads(open('$O/b_$1.json').read().strip().splitlines()[-1])
    c = l.get('compress', l)
    print('$1', 'decode' if 'compress' in l else '', l['value'] if 'compress' in l else '', 'compress', c['value'], c['ms_per_step'], {k.replace('zhip_encode_','').replace('_kernel',''): v['avg_ms'] for k, v in c['kernels'].items()})
except Exception as e: print('$1', 'ERR', e, open('$O/b_$1.err').read()[-600:])
P
}
D="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-host-api"
C="python bench.py --config compress --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-host-api"
timeout 300 $D > $O/b_default.json 2> $O/b_default.err; show default
ZHIP_TABLES_CONTIG=1 timeout 300 $D > $O/b_default_contig.json 2> $O/b_default_contig.err; show default_contig
timeout 300 $C > $O/b_alone.json 2> $O/b_alone.err; show alone
ZHIP_TABLES_CONTIG=1 timeout 300 $C > $O/b_alone_contig.json 2> $O/b_alone_contig.err; show alone_contig
