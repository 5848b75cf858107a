#!/bin/bash
# round 6, GPU session O: the flat match kernels of large launches with per-lane LDS source windows in their second, minimal form (the search's structure untouched: the own
# bytes and the repeat-offset candidates' bytes come from windows where they hold them) -- parity, then compress runs alternating with the build without windows
# (-DZE_FLAT_NOWIN), placement picked in every process, and the L2's request counters of both
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06o && O=gpurun_out/r06o
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_compress.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_compress.txt
for i in 1 2 3; do for V in nowin product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V = nowin ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_nowin.so
  ZHIP_LIB=$L timeout 600 python bench.py --config compress --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V 65536', d['value'], d['ms_per_step'], round(d['kernels']['zhip_encode_match_flat_kernel']['avg_ms'],1), d['regime']['table_pick'])" | tee -a $O/window_ab.txt
done; done
for V in nowin product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V = nowin ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_nowin.so
  ZHIP_LIB=$L timeout 600 python bench.py --config roundtrip --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V 131072', d['compress']['value'], d['compress']['ms_per_step'], d['compress']['kernels'])" | tee -a $O/window_ab.txt
  P=$O/tmp; rm -rf $P; mkdir -p $P
  ZHIP_LIB=$L ZHIP_E1F_PICK=0 timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $P -- python bench.py --config compress --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $P/err.log
  for f in $(find $P -name "*counter_collection.csv"); do python - "$f" $V <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "match_flat" not in r.get("Kernel_Name", ""): continue
    k = r.get("Counter_Name", "?"); acc[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
for k in sorted(acc): print(sys.argv[2], "match_flat", k, "mean_per_launch=%.6g" % (acc[k] / cnt[k]))
PY
  done | tee -a $O/window_counters.txt
done
rm -rf $O/tmp
