#!/bin/bash
# (this session's tools -- tests/tools/e1f_window_sweep.py, the ZHIP_E1F_WIN knob, the windowed kernels -- are in git tag r06-e1f-window: the form was measured and removed)
# round 6, GPU session B: (1) what a trip of the flat match kernel costs in INSTRUCTIONS against waiting, rounds 1-5's form (ZHIP_E1F_WIN=0) against the
# LDS-window form with long matches counted on by their own lane: SQ counters at 8 192 sources per launch (one wave per 8 SIMDs: the chain-bound end) and at
# 65 536; (2) the A/B sweep of session A again on this build
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06b && O=gpurun_out/r06b
export TMPDIR=/tmp ZHIP_E1F_PICK=0 ZHIP_E1LDS_MAX=0
for F in 8192 65536; do for W in 0 1; do
  P=$O/sq_${F}_win$W; mkdir -p $P
  ZHIP_E1F_WIN=$W timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $P -- python bench.py --config compress --frames $F --no-cpu-baseline --steps 1 --warmup 0 > $P/bench.json 2> $P/err.log
  ZHIP_E1F_WIN=$W timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVES --output-format csv -d $P/b -- python bench.py --config compress --frames $F --no-cpu-baseline --steps 1 --warmup 0 > $P/bench2.json 2>> $P/err.log
  for f in $(find $P -name "*counter_collection.csv"); do python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    if "match_flat" not in r.get("Kernel_Name", ""): continue
    k = (r.get("Kernel_Name", "?")[:44], r.get("Counter_Name", "?"))
    acc[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
for k in sorted(acc): print(sys.argv[1].split("/")[2], k[0], k[1], "mean_per_launch=%.5g" % (acc[k] / cnt[k]), "dispatches=%d" % cnt[k])
PY
  done | tee -a $O/sq_summary.txt
  find $P -name "*.csv" -delete; find $P -name "*.db" -delete
done; done
timeout 900 python tests/tools/e1f_window_sweep.py 8192,32768,65536,131072 2>&1 | grep -v "^$" | tail -6 | tee $O/e1f_window_sweep.txt
du -sh gpurun_out/r06b
