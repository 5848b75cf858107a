# r02m: K3's LDS-side piece copies -- unconditional reads / sink-slot stores against the branchy forms (isolated kernels: ZHIP_NSLOT=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
run() { tag=$1; shift
  env "$@" timeout 400 python bench.py --config decompress --no-cpu-baseline --steps 5 > $O/r02m_$tag.json 2> $O/r02m_$tag.err
  python - $tag $O/r02m_$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_decode_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open(sys.argv[2].replace(".json", ".err")).read()[-800:])
PY
}
for v in base nostlds noldlds nolds; do
  lib=$V/libzstd_hip_$v.so; [ $v = base ] && lib=$V/libzstd_hip.so
  run ${v}_iso ZHIP_LIB=$R/$lib ZHIP_NSLOT=1
done
run base ZHIP_X=1
run nostlds ZHIP_LIB=$R/$V/libzstd_hip_nostlds.so
