# r02x: parallel work-order kernel -- full GPU suite, dictionary bench, decode bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02x_pytest.log 2>&1; tail -3 $O/r02x_pytest.log

python - $O/r02x_dict.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); x = d["decompress"]; print("dict decompress", x["value"], x["ms_per_step"], x["round_trip_exact"], {k: v["avg_ms"] for k, v in x["kernels"].items()})
PY
for m in iso full; do
  if [ $m = iso ]; then export ZHIP_NSLOT=1; else unset ZHIP_NSLOT; fi
  timeout 400 python bench.py --config decompress --no-cpu-baseline --steps 5 > $O/r02x_$m.json 2> $O/r02x_$m.err
  python - $m $O/r02x_$m.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_decode_", "").replace("_kernel", ""): (v["avg_ms"], v["launches"]) for k, v in d.get("kernels", {}).items()})
PY
done
