# r02f: K2 at raised wave priority beside K3 (co-residency shapes), chunking knobs, reference hot-path tests, microbenchmarks
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): v["avg_ms"] for k, v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for v in basep co40p co44p co48p; do
  ZHIP_LIB=$R/$V/libzstd_hip_$v.so timeout 200 python bench.py --compress-frames 0 --no-cpu-baseline --steps 5 > $O/r02f_var_$v.json 2> $O/r02f_var_$v.err; show $v $O/r02f_var_$v.json
done
for cfg in "16384 2" "16384 3" "11264 3" "22528 2" "8192 3"; do
  set -- $cfg
  ZHIP_DCHUNK=$1 ZHIP_NSLOT=$2 ZHIP_LIB=$R/$V/libzstd_hip_co44p.so timeout 200 python bench.py --compress-frames 0 --no-cpu-baseline --steps 5 > $O/r02f_co44p_$1_$2.json 2> $O/r02f_co44p_$1_$2.err; show co44p_chunk$1_slots$2 $O/r02f_co44p_$1_$2.json
done
sh tests/run_reference_hotpath_tests.sh > $O/r02f_reference_tests.out 2>&1; tail -12 $O/r02f_reference_tests.out
timeout 100 tests/ubench/ubench > $O/r02f_ubench.txt 2>&1; echo "ubench rc $?" >> $O/r02f_ubench.txt
tail -22 $O/r02f_ubench.txt
