#!/bin/bash
# (this session's tools -- tests/tools/e1f_window_sweep.py, the ZHIP_E1F_WIN knob, the windowed kernels -- are in git tag r06-e1f-window: the form was measured and removed)
# round 6, GPU session A: the flat match kernel with the lanes' own bytes in LDS windows (ze_dfast_flat_w) -- parity first (compress tests, every frame against
# libzstd), then the A/B against rounds 1-5's form at 8 192 ... 131 072 sources per launch
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06a && O=gpurun_out/r06a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_boundary.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_compress.txt
timeout 900 python tests/tools/e1f_window_sweep.py 8192,32768,65536,131072 2>&1 | grep -v "^$" | tail -8 | tee $O/e1f_window_sweep.txt
timeout 600 python bench.py --config compress --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/bench_compress.json; python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d[\"value\"], d[\"ms_per_step\"], d[\"kernels\"], d.get(\"regime\"))" $O/bench_compress.json
