#!/bin/bash
# round 5, GPU session M (diagnostic): does the match kernel care how far apart its 65 536 sequence streams lie? ZHIP_DIAG_EARENA_SEQS=n shrinks the
# per-source sequence slot from 43 712 to n entries (unsafe in general -- the search has no bound --, fine on the bench corpus whose frames stay
# below 12 000 sequences): 23 GiB -> 6.5 GiB of arena. Three processes each, pick on (the regime is then the fast one nearly always).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05m && O=gpurun_out/r05m
export TMPDIR=/tmp
B="python bench.py --config compress --steps 3 --warmup 2 --no-cpu-baseline"
for k in 1 2 3; do
  timeout 300 $B 2>/dev/null | tail -1 > $O/slots_43712_$k.json
  ZHIP_DIAG_EARENA_SEQS=12288 timeout 300 $B 2>/dev/null | tail -1 > $O/slots_12288_$k.json
done
python - <<'PY' | tee $O/e1f_sequence_slot_size.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r05m/slots_*.json")):
    try:
        d = json.load(open(f)); print("%-22s %6.3f GB/s  match %7.2f ms  entropy %6.2f ms  pick %s" % (f.split("/")[-1], d["value"], d["kernels"]["zhip_encode_match_flat_kernel"]["avg_ms"], d["kernels"]["zhip_encode_entropy_kernel"]["avg_ms"], d["regime"]["table_pick"]))
    except Exception as e:
        print(f, "failed", e)
PY
