# r02i: dictionary batches as ONE chunk (per-lane tables and arena slots of the attach cutoff's size), lane-serial match kernel with 8 / 16 / 32 / 64 frames per wave
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
    for sub in ("compress", "decompress"):
        if sub in d: print("   ", sub, d[sub]["value"], d[sub]["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): v["avg_ms"] for k, v in d[sub]["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
for v in base e1l16 e1l32 e1l64; do
  lib=$V/libzstd_hip_$v.so; [ $v = base ] && lib=$V/libzstd_hip.so
  ZHIP_LIB=$R/$lib timeout 300 python bench.py --config dict --no-cpu-baseline --steps 3 > $O/r02i_dict_$v.json 2> $O/r02i_dict_$v.err; show dict_$v $O/r02i_dict_$v.json
done
timeout 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_compress.py -m gpu -x -q > $O/r02i_pytest.log 2>&1; tail -4 $O/r02i_pytest.log
