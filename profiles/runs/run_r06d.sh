#!/bin/bash
# round 6, GPU session D: what the flat match kernel's random table traffic costs the memory system in BYTES -- the L2's memory-side request counters by request
# size (TCC_EA0_RDREQ / _32B, TCC_EA0_WRREQ / _64B), rounds 1-5's form and the LDS-window form at 65 536 sources, and K3 of the decode pipeline for comparison
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06d && O=gpurun_out/r06d
export TMPDIR=/tmp ZHIP_E1F_PICK=0 ZHIP_E1LDS_MAX=0
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z_]*REQ[A-Z0-9_]*\|TCC_HIT[A-Z_]*\|TCC_MISS[A-Z_]*\|TCP_TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/tcc_counters_available.txt
summ() { for f in $(find $1 -name "*counter_collection.csv"); do python - "$f" "$2" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    kn = r.get("Kernel_Name", "")
    if not ("match_flat" in kn or "decode_exec" in kn or "decode_seq" in kn or "decode_huf" in kn or "decode_lit" in kn or "entropy" in kn): continue
    k = (kn[:40], r.get("Counter_Name", "?"))
    acc[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
for k in sorted(acc): print(sys.argv[2], k[0], k[1], "mean_per_launch=%.6g" % (acc[k] / cnt[k]), "dispatches=%d" % cnt[k])
PY
done; }
for W in 0 1; do
  for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
    P=$O/tmp; rm -rf $P; mkdir -p $P
    ZHIP_E1F_WIN=$W timeout 300 rocprofv3 --pmc $SET --output-format csv -d $P -- python bench.py --config compress --frames 65536 --no-cpu-baseline --steps 1 --warmup 0 > $P/bench.json 2> $P/err.log
    summ $P compress_win$W | tee -a $O/tcc_summary.txt
  done
done
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  P=$O/tmp; rm -rf $P; mkdir -p $P
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $P -- python bench.py --config decompress --frames 65536 --compress-frames 0 --no-cpu-baseline --no-host-api --no-extra --steps 1 --warmup 0 > $P/bench.json 2> $P/err.log
  summ $P decode | tee -a $O/tcc_summary.txt
done
rm -rf $O/tmp
