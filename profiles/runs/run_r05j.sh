#!/bin/bash
# round 5, GPU session J: bench.py exactly as the driver runs it (default line), timed by the shell
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05j && O=gpurun_out/r05j
export TMPDIR=/tmp
S=$(date +%s.%N); timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? wall $(echo "$(date +%s.%N) - $S" | bc) s" | tee $O/bench_default.rc
python - <<'PY' | tee $O/summary.txt
import json
d = json.loads([l for l in open("gpurun_out/r05j/bench_default.json") if l.startswith("{")][-1])
print("default line: value %.2f GB/s  ms_per_step %.3f  combined %s  compress %s  regime %s" % (d["value"], d["ms_per_step"], d.get("combined", {}).get("value"), d["compress"]["value"], json.dumps(d["compress"].get("regime", {}).get("table_pick"))))
print("  kernels", {k.replace("zhip_decode_", ""): v["avg_ms"] for k, v in d["kernels"].items()})
print("  roofline frac", d["roofline"]["frac"], "end_to_end", d["roofline"]["end_to_end"]["frac"])
print("  host_api", json.dumps(d.get("host_api")))
for k in ("dict", "roundtrip", "blocks"):
    s = d.get(k, {})
    print("  %s: value %s  %s %s" % (k, s.get("value"), {kk: s[kk].get("value") for kk in ("compress", "decompress") if isinstance(s.get(kk), dict)}, s.get("error", "")))
PY
