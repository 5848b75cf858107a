# r02j: table-copy mode + multi-block dictionary frames on the device -- GPU suite, the reference's own hot-path tests, dictionary bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/r02j_pytest.log 2>&1; tail -5 $O/r02j_pytest.log
sh tests/run_reference_hotpath_tests.sh > $O/r02j_ref.log 2>&1; tail -8 $O/r02j_ref.log
for k in 1 2 3; do timeout 600 python bench.py --config dict --no-cpu-baseline --steps 5 > $O/r02j_dict$k.json 2> $O/r02j_dict$k.err; python - $O/r02j_dict$k.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["kernels"].items()}, d["decompress"]["value"])
PY
done
