# r02zd: the table builders as out-of-line (noinline) routines -- entropy kernel time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
timeout 600 python bench.py --config compress --no-cpu-baseline --steps 3 > $O/r02zd_compress.json 2> $O/r02zd_compress.err
python - $O/r02zd_compress.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("compress", d["value"], d["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
PY
timeout 300 python bench.py --config dict --no-cpu-baseline --steps 5 > $O/r02zd_dict.json 2> $O/r02zd_dict.err
python - $O/r02zd_dict.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("dict", d["value"], d["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
PY
