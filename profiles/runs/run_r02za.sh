# r02za: K3 (no dictionary) at three waves per SIMD against four; smoke()
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --config decompress --no-cpu-baseline --steps 5 > $O/r02za_$tag.json 2> $O/r02za_$tag.err
  python - $tag $O/r02za_$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_decode_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open(sys.argv[2].replace(".json", ".err")).read()[-800:])
PY
}
run w4_iso ZHIP_NSLOT=1
run w3_iso ZHIP_LIB=$R/$V/libzstd_hip_k3w3.so ZHIP_NSLOT=1 ZHIP_K3_PER_CU=12
run w4 ZHIP_X=1
run w3 ZHIP_LIB=$R/$V/libzstd_hip_k3w3.so ZHIP_K3_PER_CU=12
