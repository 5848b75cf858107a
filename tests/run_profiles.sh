# r01c profile collection (run on the GPU box through gpurun): kernel trace + stats, then one PMC pass per counter
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
P=gpurun_out/prof; rm -rf $P; mkdir -p $P/kt $P/fetch $P/write
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- python bench.py --frames 32768 --steps 3 --warmup 1 --no-cpu-baseline > $P/kt/bench.json 2> $P/kt/err.log
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python bench.py --frames 32768 --steps 1 --warmup 1 --no-cpu-baseline > $P/fetch/bench.json 2> $P/fetch/err.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- python bench.py --frames 32768 --steps 1 --warmup 1 --no-cpu-baseline > $P/write/bench.json 2> $P/write/err.log
python tests/prof_summarize.py $P | tail -8
find $P -name "*.csv" ! -name "*.zhip.csv" -delete
timeout 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -1 gpurun_out/bench_full.json
timeout 300 python bench.py --direction compress --frames 32768 --steps 3 --warmup 1 > gpurun_out/bench_enc.json 2> gpurun_out/bench_enc.err; tail -1 gpurun_out/bench_enc.json
du -sh gpurun_out
