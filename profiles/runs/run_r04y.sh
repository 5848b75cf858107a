#!/bin/bash
# round 4, GPU session Y: the flat match kernel's tables in chunk-mapped memory (ZHIP_TABLES_VMM = MiB per physical chunk; r04x: the table
# traffic alone runs 29 % faster there) -- compress-only and bench.py's default order, and the dictionary batch whose tables are the same buffer
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04y && O=gpurun_out/r04y
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
X="python bench.py --config dict --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; cmd=$1; shift; env "$@" timeout 300 $cmd > $O/b_$name.json 2> $O/b_$name.err; show $name; }
run alone_0 "$C" ZHIP_TABLES_VMM=0
run alone_64 "$C" ZHIP_TABLES_VMM=64
run alone_2 "$C" ZHIP_TABLES_VMM=2
run alone_16 "$C" ZHIP_TABLES_VMM=16
run default_64 "$D" ZHIP_TABLES_VMM=64
run dict_0 "$X" ZHIP_TABLES_VMM=0
run dict_64 "$X" ZHIP_TABLES_VMM=64
