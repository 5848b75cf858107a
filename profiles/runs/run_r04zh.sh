#!/bin/bash
# round 4, GPU session ZH: the round's last build -- the whole GPU suite, smoke(), the default bench line, the single-rank RCCL round trip;
# and one A/B: the 131 072-source launches of the round trip with three probes per trip instead of two
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zh && O=gpurun_out/r04zh
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
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
R="python bench.py --config roundtrip --steps 2 --warmup 1 --no-cpu-baseline"
for v in 65536 131072 65536 131072; do ZHIP_FLAT3_MAX=$v timeout 300 $R > $O/rt_flat3max_$v.json 2>> $O/rt.err; python - <<PY
import json
l = json.loads(open("$O/rt_flat3max_$v.json").read().strip().splitlines()[-1]); c = l["compress"]
print("roundtrip, three probes up to $v sources per launch:", l["value"], "compress", c["value"], {k.replace('zhip_encode_','').replace('_kernel',''): v["avg_ms"] for k, v in c["kernels"].items()})
PY
done
ZHIP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 timeout 600 python bench.py --config roundtrip --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -c 200 $O/bench_force_dist.json
