# r02c: first GPU session of round 2 after the re-entry: microbenchmarks, the GPU suite, the new bench lines (default + dict), decode
# kernel-shape variants (A/B through ZHIP_LIB), phase timers.   gpurun --timeout 1500 -- 'sh tests/run_r02c.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
timeout 120 tests/ubench/ubench > $O/r02c_ubench.txt 2>&1; echo "ubench rc $?" >> $O/r02c_ubench.txt
cat $O/r02c_ubench.txt
ZHIP_WATCHDOG=1 timeout 300 python bench.py > $O/r02c_bench_full_65536.json 2> $O/r02c_bench_full.err; echo "bench rc $?"
cut -c1-2500 $O/r02c_bench_full_65536.json; tail -3 $O/r02c_bench_full.err
for v in base k2l30 k2l15 k2l7 huf8 huf4 longall c1 c2; do
  lib=$V/libzstd_hip_$v.so; [ $v = base ] && lib=$V/libzstd_hip.so
  ZHIP_LIB=$R/$lib timeout 200 python bench.py --compress-frames 0 --no-cpu-baseline --steps 5 > $O/r02c_var_$v.json 2> $O/r02c_var_$v.err
  echo "== $v rc $?"; python - <<PY
import json
try:
    d=json.load(open("$O/r02c_var_$v.json")); print("$v", d["value"], d["ms_per_step"], {k.replace("zhip_decode_","").replace("_kernel",""):v["avg_ms"] for k,v in d["kernels"].items()})
except Exception as e:
    print("$v failed", e); print(open("$O/r02c_var_$v.err").read()[-800:])
PY
done
for v in base c1 c2; do
  lib=$V/libzstd_hip_$v.so; [ $v = base ] && lib=$V/libzstd_hip.so
  ZHIP_NSLOT=1 ZHIP_LIB=$R/$lib timeout 200 python bench.py --compress-frames 0 --no-cpu-baseline --steps 3 > $O/r02c_iso_$v.json 2> $O/r02c_iso_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$O/r02c_iso_$v.json")); print("isolated $v", d["value"], d["ms_per_step"], {k.replace("zhip_decode_","").replace("_kernel",""):v["avg_ms"] for k,v in d["kernels"].items()})
except Exception as e:
    print("iso $v failed", e)
PY
done
ZHIP_PROF=1 timeout 200 python bench.py --frames 32768 --compress-frames 0 --no-cpu-baseline --steps 1 --warmup 1 > $O/r02c_prof.json 2> $O/r02c_prof.err; grep zhip-prof $O/r02c_prof.err | tail -24
timeout 400 python bench.py --config dict > $O/r02c_bench_dict.json 2> $O/r02c_bench_dict.err; echo "dict rc $?"; cut -c1-2500 $O/r02c_bench_dict.json; tail -5 $O/r02c_bench_dict.err
timeout 900 python -m pytest tests -m gpu -x -q --durations=12 > $O/r02c_pytest.log 2>&1; echo "pytest rc $?" >> $O/r02c_pytest.log
tail -25 $O/r02c_pytest.log
