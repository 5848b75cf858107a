#!/bin/bash
# round 4, GPU session G: SQ instruction counters of K3's round-4 form (sequential in-batch matches, 18 instructions per match) next to round 3's
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r04g; mkdir -p $O
L=$PWD/python-zstandard_amd/csrc
B="python bench.py --frames 65536 --warmup 1 --steps 1 --no-cpu-baseline --no-extra --compress-frames 0"
prof() { name=$1; lib=$2; shift 2; d=/tmp/prof_$name; mkdir -p $d; ZHIP_LIB=$lib timeout 400 rocprofv3 "$@" --output-format csv -d $d -- $B > $d/bench.json 2> $d/err.log; echo "$name rc $?"; }
prof sq_k3r4 $L/libzstd_hip_k3r4.so --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES
prof sq3_k3r4 $L/libzstd_hip_k3r4.so --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
prof sq2_k3r4 $L/libzstd_hip_k3r4.so --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INST_CYCLES_VMEM
prof sq2_base $L/libzstd_hip.so --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INST_CYCLES_VMEM
python - > $O/summary.txt <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("/tmp/prof_*/")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"]); acc[k][0] += 1; acc[k][1] += float(row["Counter_Value"])
    print("==", os.path.basename(d.rstrip("/")))
    for (kn, cn), (n, v) in sorted(acc.items()):
        if "zhip_decode" in kn and "frames" not in kn and "bin" not in kn: print("  %-28s %-22s launches %d  mean %.4g" % (kn, cn, n, v / n))
PY
cat $O/summary.txt
