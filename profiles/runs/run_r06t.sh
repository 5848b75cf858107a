#!/bin/bash
# round 6, GPU session T: where a one-shot call of one 128 KiB source spends its 35 / 2.4 ms (r06s) -- wall clock, then the same process under a kernel + HIP API trace
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06t && O=gpurun_out/r06t
export TMPDIR=/tmp
timeout 300 python tests/tools/one_shot_trace.py 2>&1 | tail -1 | tee $O/one_shot.txt
P=$O/tr; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $P -- python tests/tools/one_shot_trace.py > $P/out.txt 2> $P/err.log
tail -1 $P/out.txt | sed 's/^/under rocprof: /' | tee -a $O/one_shot.txt
python - $P <<'PY' | tee -a $O/one_shot.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    print("== kernels"); [print(r.get("Name", "")[:60], r.get("Calls"), "avg_us", round(float(r.get("AverageNs")) / 1e3, 1), "max_us", round(float(r.get("MaxNs")) / 1e3, 1)) for r in list(csv.DictReader(open(f))) if "zhip" in r.get("Name", "")]
for f in glob.glob(sys.argv[1] + "/**/*hip_api_stats.csv", recursive=True):
    print("== HIP API"); [print(r.get("Name", "")[:40], r.get("Calls"), "total_ms", round(float(r.get("TotalDurationNs")) / 1e6, 2), "avg_us", round(float(r.get("AverageNs")) / 1e3, 1), "max_us", round(float(r.get("MaxNs")) / 1e3, 1)) for r in list(csv.DictReader(open(f)))[:16]]
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "zhip_encode" in r.get("Kernel_Name", "") or "zhip_compact" in r.get("Kernel_Name", "") or "zhip_scan" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-7:]
    t0 = int(last[0]["Start_Timestamp"]) if last else 0
    print("== the last compress call's kernels")
    for r in last: print(r["Kernel_Name"][:50], "start_us", round((int(r["Start_Timestamp"]) - t0) / 1e3, 1), "dur_us", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1), "grid", r.get("Grid_Size_X", r.get("Grid_Size")))
PY
rm -rf $P
