#!/bin/bash
# round 4, GPU session N: the default bench line (pipeline roofline headline, host_api, threads1, cpu model, sub-objects) timed as the driver runs it;
# configs[4]'s process-group path with ONE rank over RCCL (ZHIP_BENCH_FORCE_DIST=1)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04n && O=gpurun_out/r04n
export TMPDIR=/tmp
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python - <<PY
import json
l = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms", l["ms_per_step"], "verified", l.get("verified"))
r = l["roofline"]; print("roofline", r["kernel"][:60], r["achieved"], r["frac"], "traffic", r["traffic"], "dominant", r["dominant_kernel"]["kernel"], r["dominant_kernel"]["frac"], "e2e", r["end_to_end"]["frac"])
print("cpu", {k: l["cpu_baseline"].get(k) for k in ("value", "cores", "threads1", "cpu_model")})
print("compress", l["compress"]["value"], l["compress"]["ms_per_step"], l["compress"]["roofline"]["frac"])
print("host_api", l.get("host_api"))
for k in ("dict", "roundtrip", "blocks"):
    s = l.get(k, {}); print(k, s.get("value"), s.get("ms_per_step"), s.get("error"), s.get("wall_s"))
PY
ZHIP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 timeout 600 python bench.py --config roundtrip --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -c 600 $O/bench_force_dist.json; tail -2 $O/bench_force_dist.err
