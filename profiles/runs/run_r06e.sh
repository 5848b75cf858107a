#!/bin/bash
# round 6, GPU session E: the whole GPU suite on the build with K1's lane-per-frame form for dictionary batches, the multi-device front of the host-buffer API and
# the kept working set; then the driver-style default bench line (configs[1] + compress + host_api + dict + roundtrip + blocks)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06e && O=gpurun_out/r06e
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench_default.err | tail -1 > $O/bench_default.json
python - $O/bench_default.json <<'PY' | tee $O/bench_default_summary.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(json.dumps(d.get("summary"), indent=0))
print("dict.decompress kernels", d.get("dict", {}).get("decompress", {}).get("kernels"))
print("host_api", d.get("host_api"))
PY
