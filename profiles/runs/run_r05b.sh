#!/bin/bash
# round 5, GPU session B: (1) the whole GPU suite on the pruned build (losing experiments of rounds 1-4 removed from csrc/, slim encode arena, idle
# loads at library-owned memory); (2) K3's memory floor (VERDICT r04 item 1's pre-flight): the decode headline with the product library and with
# libzstd_hip_floor.so (-DZP_K3_DIAG_FLOOR: K3's batch loop reduced to its global loads, scans and flush -- wrong bytes, unverified run), same box,
# same process order, K3's per-launch time from the library's HIP-event timers
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05b && O=gpurun_out/r05b
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
B="python bench.py --no-extra --no-host-api --no-cpu-baseline --compress-frames 0 --steps 10 --warmup 2"
timeout 300 $B 2>/dev/null | tail -1 > $O/bench_decode_product.json
ZHIP_BENCH_NO_VERIFY=1 ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_floor.so timeout 300 $B 2>/dev/null | tail -1 > $O/bench_decode_k3floor.json
timeout 300 $B 2>/dev/null | tail -1 > $O/bench_decode_product2.json
python - <<'PY' | tee $O/k3_floor.txt
import json
for n in ("product", "k3floor", "product2"):
    try:
        d = json.load(open("gpurun_out/r05b/bench_decode_%s.json" % n))
        print("%-9s ms_per_step %7.3f  kernels %s" % (n, d["ms_per_step"], {k.replace("zhip_decode_", ""): v["avg_ms"] for k, v in d["kernels"].items()}))
    except Exception as e:
        print(n, "failed:", e)
PY
