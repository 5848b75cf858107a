# r02r: K2 stores sequences in aligned groups of four (two 16-byte stores) -- GPU decode tests, isolated kernel times, WRITE_SIZE pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_decompress.py -m gpu -x -q > $O/r02r_pytest.log 2>&1; tail -3 $O/r02r_pytest.log
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --config decompress --no-cpu-baseline --steps 5 > $O/r02r_$tag.json 2> $O/r02r_$tag.err
  python - $tag $O/r02r_$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_decode_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open(sys.argv[2].replace(".json", ".err")).read()[-800:])
PY
}
run iso ZHIP_NSLOT=1
run full ZHIP_X=1
