#!/bin/bash
# round 6, GPU session ZZS: the placement pick with probes over the sources' first 8 KiB (and three more candidates when the first three are alike): compress-side GPU tests (the pick test compares
# all 49 152 frames of both calls; the full-size tests every frame of 65 536), then six processes of the compress bench -- which candidate was kept, what the real launches then take
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzs && O=gpurun_out/r06zzs
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_compress.py tests/test_gpu_fullsize.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300 | tee $O/pytest_compress.txt
for i in 1 2 3 4 5 6; do
  timeout 600 python bench.py --config compress --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-host-api 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d.get('regime') or (d.get('compress') or {}).get('regime'); print('process $i', d.get('value'), 'GB/s', d.get('ms_per_step'), 'ms  setup_s', d.get('setup_s'), json.dumps({k: r[k] for k in ('match_kernel_ms_per_65536_frames', 'class')}), json.dumps({k: r['table_pick'][k] for k in ('candidates_ms', 'kept')}), 'verified', d.get('verified'))" | tee -a $O/pick_outcomes.txt
done
