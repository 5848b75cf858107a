#!/bin/bash
# round 5, GPU session I: on the build with the compact decode arenas and the three-candidate table pick -- (1) the reference's own hot-path tests through
# the shim (last run in round 3), (2) bench.py exactly as the driver runs it (default line, timed), (3) the same headline through the
# process-group path with one rank (ZHIP_BENCH_FORCE_DIST=1 python bench.py --gpus 1), (4) compress in three fresh processes (which class the pick lands in)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05i && O=gpurun_out/r05i
export TMPDIR=/tmp
sh tests/run_reference_hotpath_tests.sh > /dev/null 2>&1; tail -4 gpurun_out/reference_hotpath_tests.log | tee $O/reference_hotpath_tests_tail.txt; cp gpurun_out/reference_hotpath_tests.log $O/reference_hotpath_tests.log
/usr/bin/time -v -o $O/bench_default.time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" | tee $O/bench_default.rc; grep -E "Elapsed|Maximum resident" $O/bench_default.time
ZHIP_BENCH_FORCE_DIST=1 timeout 400 python bench.py --gpus 1 --no-extra --no-host-api --steps 5 > $O/bench_force_dist_one_rank.json 2> $O/bench_force_dist_one_rank.err; echo "force-dist rc $?"
for k in 1 2 3; do timeout 300 python bench.py --config compress --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/compress_pick3_$k.json; done
python - <<'PY' | tee $O/summary.txt
import json, glob
d = json.load(open("gpurun_out/r05i/bench_default.json"))
print("default line: value %.2f GB/s  ms_per_step %.3f  combined %s  compress %s  regime %s" % (d["value"], d["ms_per_step"], d.get("combined", {}).get("value"), d["compress"]["value"], json.dumps(d["compress"].get("regime", {}).get("table_pick"))))
print("  kernels", {k.replace("zhip_decode_", ""): v["avg_ms"] for k, v in d["kernels"].items()})
print("  host_api", json.dumps(d.get("host_api")))
for k in ("dict", "roundtrip", "blocks"):
    s = d.get(k, {})
    print("  %s: value %s  %s" % (k, s.get("value"), {kk: s[kk].get("value") for kk in ("compress", "decompress") if isinstance(s.get(kk), dict)}))
f = json.load(open("gpurun_out/r05i/bench_force_dist_one_rank.json")); print("force-dist one rank: value %.2f ms_per_step %.3f" % (f["value"], f["ms_per_step"]))
for p in sorted(glob.glob("gpurun_out/r05i/compress_pick3_*.json")):
    c = json.load(open(p)); print(p.split("/")[-1], c["value"], json.dumps(c["regime"]["table_pick"]), c["regime"]["class"])
PY
