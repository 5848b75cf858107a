# r02n: K3's global-side piece loads unconditional where over-reading is safe, against the exact form (-DZP_K3_NO_GLD); full GPU suite
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02n_pytest.log 2>&1; tail -3 $O/r02n_pytest.log
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --config decompress --no-cpu-baseline --steps 5 > $O/r02n_$tag.json 2> $O/r02n_$tag.err
  python - $tag $O/r02n_$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_decode_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open(sys.argv[2].replace(".json", ".err")).read()[-800:])
PY
}
run base_iso ZHIP_NSLOT=1
run nogld_iso ZHIP_LIB=$R/$V/libzstd_hip_nogld.so ZHIP_NSLOT=1
run base ZHIP_X=1
run nogld ZHIP_LIB=$R/$V/libzstd_hip_nogld.so
