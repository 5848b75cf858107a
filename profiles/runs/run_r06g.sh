#!/bin/bash
# round 6, GPU session G: (1) kernel trace of configs[3] (where K1's time goes now that a lane-per-frame pass runs first); (2) K3 with / without the LDS history
# (-DZP_HIST_KEEP=0u -DZP_HIST_SLIDE=0u), three bench runs each, and the L2's memory-side read requests of both (TCC_EA0_RDREQ)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06g && O=gpurun_out/r06g
export TMPDIR=/tmp
P=$O/kt; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python bench.py --config dict --no-cpu-baseline --steps 3 --warmup 1 > $P/bench.json 2> $P/err.log
python - $P <<'PY' | tee $O/dict_kernel_trace.txt
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]: print(r.get("Name", "")[:44], r.get("Calls"), "avg_ns", r.get("AverageNs"), "min", r.get("MinNs"), "max", r.get("MaxNs"))
PY
find $P -name "*.csv" -delete; find $P -name "*.db" -delete
for i in 1 2 3; do for V in nohist product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V = nohist ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_nohist.so
  ZHIP_LIB=$L timeout 600 python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$V', d['value'], d['ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in k.items()})" | tee -a $O/k3_history_ab.txt
done; done
for V in nohist product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V = nohist ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_nohist.so
  P=$O/tmp; rm -rf $P; mkdir -p $P
  ZHIP_LIB=$L timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $P -- python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline --steps 1 --warmup 0 > $P/bench.json 2> $P/err.log
  for f in $(find $P -name "*counter_collection.csv"); do python - "$f" $V <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    if "decode_exec" not in r.get("Kernel_Name", ""): continue
    k = r.get("Counter_Name", "?"); acc[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
for k in sorted(acc): print(sys.argv[2], "zhip_decode_exec_kernel", k, "mean_per_launch=%.6g" % (acc[k] / cnt[k]))
PY
  done | tee -a $O/k3_history_counters.txt
done
rm -rf $O/tmp
