#!/bin/bash
# round 6, GPU session Q: K1b beside K2 as the product (no knob: the side stream is the launch order of every batch but those of small frames) -- the whole GPU suite, the
# decode line three times, a kernel trace of the decode direction (how a trace shows the two kernels side by side), and three shape variants in the new order
# (K1b at 4 / 16 frames per workgroup: smaller workgroups fit the room ONE leaving K2 wave makes; chunks of 32 768 frames on two slot streams)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06q && O=gpurun_out/r06q
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
D="python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline"
for i in 1 2 3; do for V in product huf4 huf16 dchunk32k; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 $D --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$V', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in k.items()})" | tee -a $O/variants_ab.txt
done; done
P=$O/kt; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- $D --steps 3 --warmup 1 > $P/bench.json 2> $P/err.log
python - $P <<'PY' | tee $O/decode_kernel_trace.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]: print(r.get("Name", "")[:44], r.get("Calls"), "avg_ns", r.get("AverageNs"), "min", r.get("MinNs"), "max", r.get("MaxNs"))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "zhip_decode" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[-6]["Start_Timestamp"]) if len(rows) >= 6 else 0
    for r in rows[-6:]: print("last step:", r["Kernel_Name"][:40], "start_us", (int(r["Start_Timestamp"]) - t0) / 1e3, "end_us", (int(r["End_Timestamp"]) - t0) / 1e3)
PY
tail -1 $P/bench.json > $O/bench_under_rocprof_decode.json
find $P -name "*.csv" -delete; find $P -name "*.db" -delete; rm -rf $P
