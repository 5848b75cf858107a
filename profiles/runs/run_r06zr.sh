#!/bin/bash
# round 6, GPU session ZR: what a content checksum costs the decode step (K3 verifies it with XXH64 on ONE lane) -- timers, then a kernel trace of the same tool
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zr && O=gpurun_out/r06zr
export TMPDIR=/tmp
P=$O/kt; mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python tests/tools/decode_checksum_cost.py 16384 > $P/out.txt 2> $P/err.log
tail -1 $P/out.txt | tee $O/decode_checksum_cost.txt
python - $P <<'PY' | tee -a $O/decode_checksum_cost.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "zhip_decode" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for lab, rs in (("plain, a step", rows[7 * 3: 7 * 4]), ("checksum, the last step", rows[-7:])):
        for r in rs: print(lab, r["Kernel_Name"][:30], "dur_us", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1))
PY
rm -rf $P
