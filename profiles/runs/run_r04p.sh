#!/bin/bash
# round 4, GPU session P: K3 with need-masks from a 16-byte cell map (two LDS reads instead of two binary searches) and a straight-line flush
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04p && O=gpurun_out/r04p
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -3 $O/pytest_gpu.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --compress-frames 0 --no-extra --no-host-api"
for i in 1 2; do timeout 300 $B > $O/b_new$i.json 2>> $O/b_new.err; done
for f in $O/b_*.json; do echo "$(basename $f): $(python -c "
import json,sys
l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k.replace('zhip_decode_','').replace('_kernel',''):(v['avg_ms'],v['launches']) for k,v in l['kernels'].items()})
")"; done
cd /tmp; R=$GRAFT_REPO_ROOT; cd $R
ZHIP_X=1 timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d /tmp/prof_sq -- python bench.py --frames 65536 --warmup 1 --steps 1 --no-cpu-baseline --no-extra --compress-frames 0 --no-host-api > /tmp/prof_sq.json 2> /tmp/prof_sq.err
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("/tmp/prof_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"]); acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
for (kn, cn), (n, v) in sorted(acc.items()):
    if "exec_kernel" in kn or "seq_kernel" in kn: print("  %-28s %-22s launches %d  mean %.4g  per frame %.1f" % (kn, cn, n, v / n, v / n / 65536))
PY
