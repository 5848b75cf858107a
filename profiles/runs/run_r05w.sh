#!/bin/bash
# round 5, GPU session W: two / three / four probes per trip at 65 536 sources per launch AGAIN, now that the table placement is picked (round 4 measured four probes at
# 496-500 ms in three of four rounds and 422 in one -- the placement lottery may have decided that, not the probes)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05w && O=gpurun_out/r05w
export TMPDIR=/tmp
B="python bench.py --config compress --steps 3 --warmup 2 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], {k.replace("zhip_encode_",""): v["avg_ms"] for k, v in d["kernels"].items()}, d["regime"]["table_pick"])'
for k in 1 2; do
  ZHIP_FLAT3=0 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P" two
  timeout 300 $B 2>/dev/null | tail -1 | python -c "$P" three
  ZHIP_FLAT4_MAX=65536 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P" four
done | tee $O/probes_at_65536_with_pick.txt
