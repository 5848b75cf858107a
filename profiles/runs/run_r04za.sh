#!/bin/bash
# round 4, GPU session ZA: the flat match kernel with 131 072 frames in flight (two waves per SIMD) against two launches of 65 536 -- the round-trip
# configuration (131 072 buffers per GPU), each setting twice
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04za && O=gpurun_out/r04za
export TMPDIR=/tmp
R="python bench.py --config roundtrip --steps 2 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout 300 $R > $O/b_$name.json 2> $O/b_$name.err; python - <<P
import json
try:
    l = json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); c = l['compress']
    print('$name', 'roundtrip', l['value'], 'compress', c['value'], c['ms_per_step'], {k.replace('zhip_encode_','').replace('_kernel',''): (v['avg_ms'], v['launches']) for k, v in c['kernels'].items()}, l.get('round_trip_exact'))
except Exception as e: print('$name', 'ERR', e, open('$O/b_$name.err').read()[-800:])
P
}
run c65536_a X=1
run c131072_a ZHIP_ECHUNK_MAX=131072
run c65536_b X=1
run c131072_b ZHIP_ECHUNK_MAX=131072
