# r02e: co-residency shapes (K3 with a 2 KiB assembly buffer beside a narrower K2 wave), K3 prefetch, the dictionary entropy-kernel fixes,
# the GPU suite, the reference's own hot-path tests, microbenchmarks.   gpurun --timeout 1800 -- 'sh tests/run_r02e.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): v["avg_ms"] for k, v in d["kernels"].items()})
    for sub in ("compress", "decompress"):
        if sub in d: print("   ", sub, d[sub]["value"], d[sub]["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): v["avg_ms"] for k, v in d[sub]["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for v in base pf asm2k co36 co40 co44 co48; do
  lib=$V/libzstd_hip_$v.so; [ $v = base ] && lib=$V/libzstd_hip.so
  ZHIP_LIB=$R/$lib timeout 200 python bench.py --compress-frames 0 --no-cpu-baseline --steps 5 > $O/r02e_var_$v.json 2> $O/r02e_var_$v.err; show $v $O/r02e_var_$v.json
done
for v in co36 co44; do
  for ns in 3; do
    ZHIP_NSLOT=$ns ZHIP_DCHUNK=21846 ZHIP_LIB=$R/$V/libzstd_hip_$v.so timeout 200 python bench.py --compress-frames 0 --no-cpu-baseline --steps 5 > $O/r02e_var_${v}_3slot.json 2> $O/r02e_var_${v}_3slot.err; show ${v}_3slot $O/r02e_var_${v}_3slot.json
  done
done
timeout 400 python bench.py --config dict > $O/r02e_bench_dict.json 2> $O/r02e_bench_dict.err; echo "dict rc $?"; show dict $O/r02e_bench_dict.json
ZHIP_PROF=1 timeout 300 python bench.py --config dict --docs 32768 --steps 1 --warmup 1 --no-cpu-baseline > $O/r02e_dictprof.json 2> $O/r02e_dictprof.err; grep zhip-prof $O/r02e_dictprof.err | head -10
timeout 1200 python -m pytest tests -m gpu -x -q --durations=12 > $O/r02e_pytest.log 2>&1; echo "pytest rc $?" >> $O/r02e_pytest.log
tail -22 $O/r02e_pytest.log
sh tests/run_reference_hotpath_tests.sh > $O/r02e_reference_tests.out 2>&1; tail -30 $O/r02e_reference_tests.out
timeout 100 tests/ubench/ubench > $O/r02e_ubench.txt 2>&1; echo "ubench rc $?" >> $O/r02e_ubench.txt
cat $O/r02e_ubench.txt
