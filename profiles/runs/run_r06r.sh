#!/bin/bash
# round 6, GPU session R: K3's tail. K3 takes frames in index order (the corpus' own mix spreads the far-match gathers, r03f) and ends on whatever frames come last; with
# ZHIP_X_K3LIGHT=T it walks the chunk twice -- frames of at least T sequences first, lighter ones after -- so the tail is made of short frames. Then the default line once
# (how the side stream sits with the two-chunk round trip, the dictionary batch, the several-block frames and the host-buffer calls).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06r && O=gpurun_out/r06r
export TMPDIR=/tmp
D="python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline"
for i in 1 2; do for T in 0 2000 5000 8000 11000; do
  ZHIP_X_K3LIGHT=$T timeout 600 $D --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('light_below=$T', d['value'], d['ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in k.items()})" | tee -a $O/k3_light_last_ab.txt
done; done
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python -c "
import json,sys; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('summary')))" | tee $O/bench_default_summary.txt
