#!/bin/bash
# round 6, GPU session ZE: the flat dictionary search without its per-launch table zeroing -- launch numbers in the cells (ZhipEncodeArgs.tabEpoch) -- against the build that
# zeroes (-DZHIP_DICT_EPOCHS=0): the dictionary tests (every frame against libzstd), then configs[3] alternating, and the L2's request counters of both
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06ze && O=gpurun_out/r06ze
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_compress.py tests/test_gpu_boundary.py tests/test_cext_backend.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_compress.txt
for i in 1 2 3; do for V in noepoch product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python bench.py --config dict --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V dict', d['value'], d['ms_per_step'], {n.replace('zhip_encode_','').replace('_kernel',''): round(v['avg_ms'],2) for n,v in d['kernels'].items()}, d.get('table_pick'))" | tee -a $O/dict_epochs_ab.txt
done; done
for V in noepoch product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  P=$O/tmp; rm -rf $P; mkdir -p $P
  ZHIP_LIB=$L ZHIP_E1F_PICK=0 timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $P -- python bench.py --config dict --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $P/err.log
  for f in $(find $P -name "*counter_collection.csv"); do python - "$f" $V <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "match_flat" not in r.get("Kernel_Name", ""): continue
    acc[r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0)))
for k in sorted(acc): print(sys.argv[2], "match_flat", k, "per launch (last):", "%.6g" % acc[k][-1], "launches", len(acc[k]))
PY
  done | tee -a $O/dict_epochs_counters.txt
done
rm -rf $O/tmp
