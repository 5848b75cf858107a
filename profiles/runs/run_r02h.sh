# r02h: entropy kernel without the match search compiled in (dict + compress lines), K2 split variants
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): v["avg_ms"] for k, v in d.get("kernels", {}).items()})
    for sub in ("compress", "decompress"):
        if sub in d: print("   ", sub, d[sub]["value"], d[sub]["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): v["avg_ms"] for k, v in d[sub]["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 400 python bench.py --config dict --no-cpu-baseline --steps 5 > $O/r02h_bench_dict.json 2> $O/r02h_bench_dict.err; echo "dict rc $?"; show dict $O/r02h_bench_dict.json
timeout 400 python bench.py --config compress --no-cpu-baseline --steps 3 > $O/r02h_bench_compress.json 2> $O/r02h_bench_compress.err; show compress $O/r02h_bench_compress.json
for v in base split split44; do
  lib=$V/libzstd_hip_$v.so; [ $v = base ] && lib=$V/libzstd_hip.so
  ZHIP_LIB=$R/$lib timeout 200 python bench.py --mix default --compress-frames 0 --no-cpu-baseline --steps 5 > $O/r02h_var_$v.json 2> $O/r02h_var_$v.err; show $v $O/r02h_var_$v.json; tail -2 $O/r02h_var_$v.err | grep -v amdgpu.ids
  ZHIP_NSLOT=1 ZHIP_LIB=$R/$lib timeout 200 python bench.py --mix default --compress-frames 0 --no-cpu-baseline --steps 3 > $O/r02h_iso_$v.json 2> $O/r02h_iso_$v.err; show iso_$v $O/r02h_iso_$v.json
done
