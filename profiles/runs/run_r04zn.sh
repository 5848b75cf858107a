#!/bin/bash
# round 4, GPU session ZN: the default bench line on the round's last commit
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zn && O=gpurun_out/r04zn
export TMPDIR=/tmp
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time | head -1
python - <<PY
import json
l = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", l["value"], "ms", l["ms_per_step"], "verified", l.get("verified"), "kernels", {k.replace('zhip_decode_','').replace('_kernel',''): v["avg_ms"] for k, v in l["kernels"].items()})
print("compress", l["compress"]["value"], l["compress"]["ms_per_step"], {k.replace('zhip_encode_','').replace('_kernel',''): v["avg_ms"] for k, v in l["compress"]["kernels"].items()})
print("host_api", l.get("host_api"))
for k in ("dict", "roundtrip", "blocks"):
    s = l.get(k, {}); print(k, s.get("value"), s.get("ms_per_step"), s.get("error"), (s.get("compress") or {}).get("value"), (s.get("decompress") or {}).get("value"))
PY
