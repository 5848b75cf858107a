# r02v: dictionary frames through the decode pipeline -- full GPU suite, the reference's hot-path tests, dictionary bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02v_pytest.log 2>&1; tail -4 $O/r02v_pytest.log
sh tests/run_reference_hotpath_tests.sh > $O/r02v_ref.log 2>&1; tail -3 $O/r02v_ref.log
timeout 600 python bench.py --config dict --no-cpu-baseline --steps 5 > $O/r02v_dict.json 2> $O/r02v_dict.err
python - $O/r02v_dict.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("compress", d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()})
x = d["decompress"]; print("decompress", x["value"], x["ms_per_step"], x["round_trip_exact"], {k: v["avg_ms"] for k, v in x["kernels"].items()})
PY
