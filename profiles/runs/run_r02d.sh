# r02d: software-pipelined K2 loop (+ K3 all-long-matches form and 8-frame K1b waves as defaults), the pipelined host-buffer API, the
# entropy kernel's phase timers on the dictionary config, microbenchmarks.   gpurun --timeout 1800 -- 'sh tests/run_r02d.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
V=python-zstandard_amd/csrc
timeout 120 tests/ubench/ubench > $O/r02d_ubench.txt 2>&1; echo "ubench rc $?" >> $O/r02d_ubench.txt
cat $O/r02d_ubench.txt
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
ZHIP_WATCHDOG=1 timeout 200 python bench.py --compress-frames 0 --no-cpu-baseline --steps 5 > $O/r02d_dec.json 2> $O/r02d_dec.err; echo "rc $?"; show decode $O/r02d_dec.json; tail -2 $O/r02d_dec.err
ZHIP_NSLOT=1 timeout 200 python bench.py --compress-frames 0 --no-cpu-baseline --steps 3 > $O/r02d_iso.json 2> $O/r02d_iso.err; show isolated $O/r02d_iso.json
ZHIP_PROF=1 timeout 300 python bench.py --config dict --docs 32768 --steps 1 --warmup 1 --no-cpu-baseline > $O/r02d_dictprof.json 2> $O/r02d_dictprof.err; grep zhip-prof $O/r02d_dictprof.err | tail -30
ZHIP_LIB=$R/$V/libzstd_hip_tab3.so timeout 300 python bench.py --config dict --steps 3 --no-cpu-baseline > $O/r02d_dict_tab3.json 2> $O/r02d_dict_tab3.err; show dict_tab3 $O/r02d_dict_tab3.json
timeout 600 python tests/host_api_rate.py 16384 > $O/r02d_host_api_16384.log 2>&1; tail -3 $O/r02d_host_api_16384.log
timeout 900 python tests/host_api_rate.py 65536 > $O/r02d_host_api_65536.log 2>&1; tail -3 $O/r02d_host_api_65536.log
timeout 1200 python -m pytest tests -m gpu -x -q --durations=12 > $O/r02d_pytest.log 2>&1; echo "pytest rc $?" >> $O/r02d_pytest.log
tail -22 $O/r02d_pytest.log
ZHIP_WATCHDOG=1 timeout 300 python bench.py > $O/r02d_bench_full_65536.json 2> $O/r02d_bench_full.err; echo "bench rc $?"; show full $O/r02d_bench_full_65536.json
