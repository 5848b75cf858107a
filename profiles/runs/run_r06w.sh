#!/bin/bash
# round 6, GPU session W: K0 -- a lane per frame walks what K1's lane 0 used to (Huffman weights' description, the three sequence distributions), K1 takes weights and counts
# from its records. Parity (decompress + boundary + full-size tests), then the decode line alternating with the build without K0 (-DZHIP_K0=0), and a kernel trace.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06w && O=gpurun_out/r06w
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py tests/test_gpu_fullsize.py tests/test_gpu_multidevice.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_decode.txt
D="python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline"
for i in 1 2 3; do for V in nok0 product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 $D --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$V', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in k.items()})" | tee -a $O/k0_ab.txt
done; done
P=$O/kt; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- $D --steps 3 --warmup 1 > $P/bench.json 2> $P/err.log
python - $P <<'PY' | tee $O/decode_kernel_trace.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f))): 
        if "zhip" in r.get("Name", ""): print(r.get("Name", "")[:44], r.get("Calls"), "avg_ns", r.get("AverageNs"), "min", r.get("MinNs"), "max", r.get("MaxNs"))
PY
rm -rf $P
