#!/bin/bash
# round 4, GPU session M: timeline of the host-API compress call with THREE slots (r04l: 2.4 GB/s)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r04m; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -- python tests/host_api_rate.py 65536 > $O/run.log 2>&1
python - > $O/timeline.txt <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:40], r.get("Stream_Id", "")))
for f in glob.glob("/tmp/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r.get("Direction", r.get("Name", "")), ""))
rows.sort()
# the LAST batch of encode kernels = the last timed compress call: find the last 8 match_flat kernels (2 calls x 4 chunks) -> take the last 4
flat = [i for i, r in enumerate(rows) if "match_flat" in r[2]]
if len(flat) >= 4:
    i0 = flat[-4]
    t0 = rows[i0][0]
    # back up to the first H2D copy of that call (copies within 1.5 s before)
    j = i0
    while j > 0 and rows[j - 1][0] > t0 - 600_000_000 and not ("encode_entropy" in rows[j - 1][2]): j -= 1
    base = rows[j][0]
    for s, e, name, st in rows[j:]:
        if e - s < 200_000 and not name.startswith("K zhip_encode"): continue
        print("%9.2f ms  +%8.2f ms  %s %s" % ((s - base) / 1e6, (e - s) / 1e6, name, st))
PY
head -80 $O/timeline.txt; tail -1 $O/run.log | cut -c1-200
