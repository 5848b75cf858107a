#!/bin/bash
# round 4, GPU session ZL: the two-probe search as the NP = 2 instance of the generalised function (the hand-written two-probe function removed) -- the GPU
# suite, and the launches that still take two probes: 131 072 sources per launch (round trip), the LDS-source kernel (small batches), 32 768 x 256 KiB
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zl && O=gpurun_out/r04zl
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 300 python bench.py --config roundtrip --steps 2 --warmup 1 --no-cpu-baseline > $O/rt.json 2> $O/rt.err; python - <<P
import json
l = json.loads(open('$O/rt.json').read().strip().splitlines()[-1]); c = l['compress']
print('round trip', l['value'], 'compress', c['value'], {k.replace('zhip_encode_','').replace('_kernel',''): v['avg_ms'] for k, v in c['kernels'].items()})
P
timeout 600 python tests/small_batch_latency.py > $O/small_batch_latency.txt 2>&1; tail -1 $O/small_batch_latency.txt | cut -c1-330
timeout 600 python tests/multiblock_rate.py 32768 256 > $O/mb_32768x256.log 2>&1; python - <<P
import json
l = json.loads(open('$O/mb_32768x256.log').read().strip().splitlines()[-1])
print('32768 x 256 KiB: compress %.2f GB/s (%.1f ms)' % (l['compress_GBps'], l['compress_ms']))
P
