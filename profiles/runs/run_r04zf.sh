#!/bin/bash
# round 4, GPU session ZF: the final build after the four-probe search -- the whole GPU suite, smoke(), the default bench line as the driver runs it, configs[4]'s process-group
# path with ONE rank over RCCL (ZHIP_BENCH_FORCE_DIST=1)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zf && O=gpurun_out/r04zf
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -4 $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python - <<PY
import json
l = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms", l["ms_per_step"], "verified", l.get("verified"))
r = l["roofline"]; print("roofline", r["kernel"][:60], r["achieved"], r["frac"], "traffic", r["traffic"], "dominant", r["dominant_kernel"]["kernel"], r["dominant_kernel"]["frac"], "e2e", r["end_to_end"]["frac"])
print("kernels", {k: v["avg_ms"] for k, v in l["kernels"].items()})
print("cpu", {k: l["cpu_baseline"].get(k) for k in ("value", "cores", "threads1", "cpu_model")})
print("compress", l["compress"]["value"], l["compress"]["ms_per_step"], l["compress"]["kernels"])
print("host_api", l.get("host_api"))
for k in ("dict", "roundtrip", "blocks"):
    s = l.get(k, {}); print(k, s.get("value"), s.get("ms_per_step"), s.get("error"), s.get("wall_s"), (s.get("compress") or {}).get("value"), (s.get("decompress") or {}).get("value"))
PY
ZHIP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 timeout 600 python bench.py --config roundtrip --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -c 400 $O/bench_force_dist.json; tail -2 $O/bench_force_dist.err
