#!/bin/bash
# round 6, GPU session F: K3 with the flushed output kept in front of the batch as an LDS history (far-match sources inside it staged LDS -> LDS) against the same
# build without it (-DZP_HIST_KEEP=0u -DZP_HIST_SLIDE=0u), three bench runs each; K1's lane-per-frame pass for dictionary batches as a kernel of its own
# (configs[3]); the GPU suite; the driver-style default line
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06f && O=gpurun_out/r06f
export TMPDIR=/tmp
for i in 1 2 3; do for V in nohist product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V = nohist ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_nohist.so
  ZHIP_LIB=$L timeout 600 python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$V', d['value'], d['ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in k.items()})" | tee -a $O/k3_history_ab.txt
done; done
timeout 600 python bench.py --config dict --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dict compress', d['value'], 'decompress', d['decompress']['value'], d['decompress']['ms_per_step'], d['decompress']['kernels'])" | tee $O/dict.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python -c "
import json,sys; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('summary'))); print(d['kernels'])" | tee $O/bench_default_summary.txt
