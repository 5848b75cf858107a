#!/bin/bash
# round 6, GPU session ZZP: configs[3]'s match kernel (262 144 x 4 KiB documents, shared dictionary) against the memory system's request ceiling: TCC request counters of one bench.py --config dict pass
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06zzp; O=gpurun_out/r06zzp; P=/tmp/prof_dict; rm -rf $P
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $P -- python bench.py --config dict --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_dict.json 2> $O/err.log; echo "rc $?"
python - <<'PY' | tee gpurun_out/r06zzp/dict_tcc_counters.txt
import csv, glob, collections
f = glob.glob('/tmp/prof_dict/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in f:
    for r in csv.DictReader(open(p)):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    if not k.startswith('zhip_'): continue
    print(k, {n: (len(v), round(sum(v) / len(v))) for n, v in c.items()})
PY
tail -c 1500 $O/bench_dict.json
