#!/bin/bash
# round 6, GPU session ZZG: the N > 1 launch path of the LAST bench.py with one rank (ZHIP_BENCH_FORCE_DIST=1 under torch.distributed.run: RCCL init, barriers, max over ranks,
# the round trip's all-gatherv) -- what the driver's 2 / 4 / 8-GPU runs execute, as far as a one-GPU box can show it; and `--gpus 2` asked of a one-GPU box (must fail loudly, not hang)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzg && O=gpurun_out/r06zzg
export TMPDIR=/tmp
ZHIP_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 2> $O/force_dist.err | tail -1 > $O/force_dist.json
echo "rc $?" | tee $O/force_dist_rc.txt
python -c "
import json; d=json.loads(open('$O/force_dist.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('summary'))); print({k: d.get(k) for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','verified')})" 2>&1 | tee $O/force_dist_summary.txt
tail -5 $O/force_dist.err | cut -c1-400 | tee $O/force_dist_err_tail.txt
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/gpus2_on_one_gpu.out 2> $O/gpus2_on_one_gpu.err; echo "gpus2 rc $?" | tee $O/gpus2_rc.txt; tail -3 $O/gpus2_on_one_gpu.err | cut -c1-400; tail -c 600 $O/gpus2_on_one_gpu.out
