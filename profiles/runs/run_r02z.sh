# r02z: the round's evidence with the final kernels -- rocprofv3 passes for profiles/ (tests/run_profiles.sh), then the four bench lines
TAG=r02 sh tests/run_profiles.sh > gpurun_out/r02z_profiles.log 2>&1; tail -3 gpurun_out/r02z_profiles.log
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python bench.py > $O/r02z_bench_decode.json 2> $O/r02z_bench_decode.err; cut -c1-300 $O/r02z_bench_decode.json
timeout 900 python bench.py --config dict > $O/r02z_bench_dict.json 2> $O/r02z_bench_dict.err; cut -c1-200 $O/r02z_bench_dict.json
timeout 900 python bench.py --config roundtrip > $O/r02z_bench_roundtrip.json 2> $O/r02z_bench_roundtrip.err; cut -c1-300 $O/r02z_bench_roundtrip.json
timeout 900 python bench.py --config compress > $O/r02z_bench_compress.json 2> $O/r02z_bench_compress.err; cut -c1-200 $O/r02z_bench_compress.json
