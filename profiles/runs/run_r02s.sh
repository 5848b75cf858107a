# r02s: wave-parallel table builders in the entropy kernel -- GPU compress tests, compress / dict bench with kernel times
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_compress.py tests/test_gpu_boundary.py tests/test_cext_backend.py -m gpu -x -q > $O/r02s_pytest.log 2>&1; tail -3 $O/r02s_pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open(sys.argv[2].replace(".json", ".err")).read()[-600:])
PY
}
timeout 600 python bench.py --config compress --no-cpu-baseline --steps 3 > $O/r02s_compress.json 2> $O/r02s_compress.err; show compress $O/r02s_compress.json
timeout 300 python bench.py --config dict --no-cpu-baseline --steps 5 > $O/r02s_dict.json 2> $O/r02s_dict.err; show dict $O/r02s_dict.json
