# r02zi: kernel trace of the dictionary config (262 144 x 4 KiB documents, trained dictionary) with the final build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
P=$O/prof_r02zi; rm -rf $P; mkdir -p $P
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python bench.py --config dict --no-cpu-baseline --steps 3 --warmup 1 > $P/bench.json 2> $P/err.log
f=$(find $P -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "zhip\|Name" "$f" | cut -d, -f1-7 > $O/r02zi_dict_kernel_stats.csv; cat $O/r02zi_dict_kernel_stats.csv
cp $P/bench.json $O/r02zi_bench_under_rocprof_dict.json
rm -rf $P
