#!/bin/bash
# round 4, GPU session U: what the link form moves -- FETCH_SIZE / WRITE_SIZE / SQ counters of the link search and its plain pre-pass beside the
# table form's, one rocprofv3 --pmc pass each, one step of 65 536 frames (raw output stays in /tmp on the box; the per-kernel sums come back)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04u && O=$PWD/gpurun_out/r04u
export TMPDIR=/tmp
B="python bench.py --config compress --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-host-api"
P=/tmp/prof_r04u; rm -rf $P; mkdir -p $P
pass() { name=$1; shift; env ZHIP_E1LINKS=1 ZHIP_E1LINK_LANES=64 timeout 400 rocprofv3 "$@" --output-format csv -d $P/$name -- $B > $P/$name.json 2> $P/$name.err; echo "$name rc $?"; }
pass fetch --pmc FETCH_SIZE
pass write --pmc WRITE_SIZE
pass sq --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
python - <<P
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('$P/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path, newline='')):
        if 'zhip_' in r.get('Kernel_Name', ''): acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$O/links_counters.csv', 'w') as f:
    f.write('Kernel,Counter,Launches,MeanPerLaunch\n')
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]; f.write('%s,%s,%d,%.3f\n' % (k, c, len(v), sum(v) / len(v)))
print(open('$O/links_counters.csv').read())
P
