#!/bin/bash
# round 4, GPU session ZB: (1) the flat match kernel's idle candidate loads pointed at one shared address (-DZE_FLAT_IDLE) against the lane's own probe
# position, compress-only, each twice in alternation (E1f has placement regimes: pairs, not single runs); (2) the round trip at its default settings
# (batches above 65 536 frames now take 131 072 per launch)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zb && O=gpurun_out/r04zb
export TMPDIR=/tmp
L=$PWD/python-zstandard_amd/csrc
C="python bench.py --config compress --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-host-api"
run() { name=$1; shift; cmd=$1; shift; env "$@" timeout 300 $cmd > $O/b_$name.json 2> $O/b_$name.err; python - <<P
import json
try:
    l = json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); c = l.get('compress', l) if '$name'.startswith('rt') else l
    print('$name', l['value'], 'compress', c['value'], c['ms_per_step'], {k.replace('zhip_encode_','').replace('_kernel',''): (v['avg_ms'], v['launches']) for k, v in c['kernels'].items()})
except Exception as e: print('$name', 'ERR', e, open('$O/b_$name.err').read()[-800:])
P
}
run base_a "$C" X=1
run idle_a "$C" ZHIP_LIB=$L/libzstd_hip_idle.so
run base_b "$C" X=1
run idle_b "$C" ZHIP_LIB=$L/libzstd_hip_idle.so
run rt_default "python bench.py --config roundtrip --steps 2 --warmup 1 --no-cpu-baseline" X=1
python tests/tools/e1f_alloc_trials.py 2 0,0 2>&1 | tee $O/trials_base.txt | tail -4
ZHIP_LIB=$L/libzstd_hip_idle.so python tests/tools/e1f_alloc_trials.py 2 0,0 2>&1 | tee $O/trials_idle.txt | tail -4
