#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r03o && O=gpurun_out/r03o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $O/sq3 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --compress-frames 0 --no-extra > $O/b.json 2> $O/err.log
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('gpurun_out/r03o/sq3/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'zhip_' in r.get('Kernel_Name',''):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k, {c: round(sum(v)/len(v)) for c, v in sorted(acc[k].items())})
PY
