# r02g: the round's bench lines on the "silesia" corpus mix (default / compress sub-report / dict / roundtrip), the host-API rate, GPU suite,
# the reference's own hot-path tests, then the rocprofv3 passes of tests/run_profiles.sh.   gpurun --timeout 2400 -- 'sh tests/run_r02g.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); print(sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("compression_ratio"), {k.replace("zhip_", "").replace("_kernel", ""): v["avg_ms"] for k, v in d.get("kernels", {}).items()})
    for sub in ("compress", "decompress"):
        if sub in d: print("   ", sub, d[sub]["value"], d[sub]["ms_per_step"], {k.replace("zhip_", "").replace("_kernel", ""): v["avg_ms"] for k, v in d[sub]["kernels"].items()})
    if "roofline" in d: print("    roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "end-to-end", d["roofline"]["end_to_end"]["frac"])
    if "cpu_baseline" in d: print("    cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > $O/r02g_pytest.log 2>&1; echo "pytest rc $?" >> $O/r02g_pytest.log; tail -14 $O/r02g_pytest.log
sh tests/run_reference_hotpath_tests.sh > $O/r02g_reference_tests.out 2>&1; tail -6 $O/r02g_reference_tests.out
ZHIP_WATCHDOG=1 timeout 400 python bench.py > $O/r02g_bench_full_65536.json 2> $O/r02g_bench_full.err; echo "bench rc $?"; show full $O/r02g_bench_full_65536.json; tail -2 $O/r02g_bench_full.err
timeout 300 python bench.py --mix default --compress-frames 0 --no-cpu-baseline > $O/r02g_bench_decode_r01mix.json 2> $O/r02g_bench_decode_r01mix.err; show decode_r01mix $O/r02g_bench_decode_r01mix.json
timeout 400 python bench.py --config dict > $O/r02g_bench_dict.json 2> $O/r02g_bench_dict.err; echo "dict rc $?"; show dict $O/r02g_bench_dict.json
timeout 600 python bench.py --config roundtrip --steps 3 --warmup 1 > $O/r02g_bench_roundtrip.json 2> $O/r02g_bench_roundtrip.err; echo "roundtrip rc $?"; show roundtrip $O/r02g_bench_roundtrip.json; tail -3 $O/r02g_bench_roundtrip.err
timeout 900 python tests/host_api_rate.py 65536 > $O/r02g_host_api_65536.log 2>&1; tail -2 $O/r02g_host_api_65536.log
TAG=r02 sh tests/run_profiles.sh > $O/r02g_profiles.log 2>&1; tail -25 $O/r02g_profiles.log
