# r02q: K2's scheduling fences -- all (product), only "cell request first", none
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --config decompress --no-cpu-baseline --steps 5 > $O/r02q_$tag.json 2> $O/r02q_$tag.err
  python - $tag $O/r02q_$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_decode_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open(sys.argv[2].replace(".json", ".err")).read()[-800:])
PY
}
run f2_iso ZHIP_NSLOT=1
run f1_iso ZHIP_LIB=$R/$V/libzstd_hip_zqf1.so ZHIP_NSLOT=1
run f0_iso ZHIP_LIB=$R/$V/libzstd_hip_zqf0.so ZHIP_NSLOT=1
