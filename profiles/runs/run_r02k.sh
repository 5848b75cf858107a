# r02k: decode chunk size against K2's round quantum (256 CUs x 60 frames = 15 360 frames per round of the persistent seq kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
run() { tag=$1; frames=$2; shift 2
  env "$@" timeout 400 python bench.py --config decompress --no-cpu-baseline --steps 5 --frames $frames > $O/r02k_$tag.json 2> $O/r02k_$tag.err
  python - $tag $O/r02k_$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_decode_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run f61440_c32768 61440 ZHIP_X=1
run f61440_c30720 61440 ZHIP_DCHUNK=30720
run f61440_c15360 61440 ZHIP_DCHUNK=15360
run f61440_c61440 61440 ZHIP_DCHUNK=61440
run f65536_c30720 65536 ZHIP_DCHUNK=30720
run f65536_c46080 65536 ZHIP_DCHUNK=46080
