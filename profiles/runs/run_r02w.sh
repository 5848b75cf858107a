# r02w: where the dictionary decode's time goes -- isolated kernels (one chunk slot) and a kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
ZHIP_NSLOT=1 timeout 600 python bench.py --config dict --no-cpu-baseline --steps 5 > $O/r02w_dict_iso.json 2> $O/r02w_dict_iso.err
python - $O/r02w_dict_iso.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); x = d["decompress"]; print("iso decompress", x["value"], x["ms_per_step"], {k: (v["avg_ms"], v["launches"]) for k, v in x["kernels"].items()})
PY
P=$O/prof_r02w; rm -rf $P; mkdir -p $P
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python bench.py --config dict --no-cpu-baseline --steps 2 --warmup 1 > $P/bench.json 2> $P/err.log
f=$(find $P -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -d, -f1-6
rm -rf $P
