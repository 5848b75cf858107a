# r02a: first measurement of the LDS-ring bit reader (K1b / K2): GPU suite, the default bench line, a decode-only kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a_pytest.log
tail -3 gpurun_out/r02a_pytest.log
ZHIP_WATCHDOG=1 timeout 600 python bench.py --compress-frames 0 > gpurun_out/r02a_bench_decode_65536.json 2> gpurun_out/r02a_bench_decode.err
cut -c1-1500 gpurun_out/r02a_bench_decode_65536.json; tail -3 gpurun_out/r02a_bench_decode.err
ZHIP_NSLOT=1 timeout 300 python bench.py --compress-frames 0 --no-cpu-baseline --steps 5 > gpurun_out/r02a_bench_decode_1slot.json 2>> gpurun_out/r02a_bench_decode.err
cut -c1-1200 gpurun_out/r02a_bench_decode_1slot.json
