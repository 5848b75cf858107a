#!/bin/bash
# round 5, GPU session H: the flat match kernel's table placement PICK (zhip_compress_batch_device: the first large launch of a context times the
# kernel on two table allocations and keeps the faster) -- the compress-side GPU tests, then bench.py --config compress in four fresh processes
# (each reports regime.table_pick and the class it ended up in), once with ZHIP_E1F_PICK=0, then the round trip config at 131 072 frames per launch
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05h && O=gpurun_out/r05h
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_compress.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_compress.txt
B="python bench.py --config compress --steps 3 --warmup 2 --no-cpu-baseline"
for k in 1 2 3 4; do timeout 300 $B 2>/dev/null | tail -1 > $O/compress_pick_$k.json; done
ZHIP_E1F_PICK=0 timeout 300 $B 2>/dev/null | tail -1 > $O/compress_nopick.json
timeout 400 python bench.py --config roundtrip --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/roundtrip.json
python - <<'PY' | tee $O/table_pick.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r05h/*.json")):
    try:
        d = json.load(open(f))
        c = d.get("compress", d) if "roundtrip" in f else d
        print("%-22s value %7.3f GB/s  ms_per_step %8.2f  regime %s  kernels %s" % (f.split("/")[-1], d["value"], d["ms_per_step"], json.dumps(d.get("regime")), {k.replace("zhip_encode_", ""): v["avg_ms"] for k, v in c.get("kernels", {}).items()}))
    except Exception as e:
        print(f, "failed:", e)
PY
