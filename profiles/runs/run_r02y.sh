# r02y: K3's dictionary instantiation with 4 (product) / 3 / 2 waves per SIMD -- dictionary bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
for v in base k3d3 k3d2; do
  lib=$V/libzstd_hip_$v.so; [ $v = base ] && lib=$V/libzstd_hip.so
  ZHIP_LIB=$R/$lib timeout 300 python bench.py --config dict --no-cpu-baseline --steps 5 > $O/r02y_$v.json 2> $O/r02y_$v.err
  python - $v $O/r02y_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); x = d["decompress"]; print(sys.argv[1], "dict decompress", x["value"], x["ms_per_step"], {k.replace("zhip_decode_", "").replace("_kernel", ""): v["avg_ms"] for k, v in x["kernels"].items()})
PY
done
