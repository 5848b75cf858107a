# r02zl: decode chunk size with the round's final kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --config decompress --no-cpu-baseline --steps 5 > $O/r02zl_$tag.json 2> $O/r02zl_$tag.err
  python - $tag $O/r02zl_$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_decode_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open(sys.argv[2].replace(".json", ".err")).read()[-500:])
PY
}
run c32768 ZHIP_X=1
run c16384 ZHIP_DCHUNK=16384
run c21846 ZHIP_DCHUNK=21846
run c65536 ZHIP_DCHUNK=65536
