# Profile collection for profiles/ (run on the GPU box:  gpurun --timeout 1500 -- 'TAG=r03 sh tests/run_profiles.sh'): one rocprofv3 pass
# per purpose as the guide prescribes -- kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ counters -- for both directions, each over
# FRAMES frames (default 65 536 = bench.py's default = ONE chunk per launch of every kernel), then tests/prof_traffic.py reduces them
# (per-kernel summaries + traffic.json) under gpurun_out/summary/; copy those into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
TAG=${TAG:-r03}
FRAMES=${FRAMES:-65536}
P=gpurun_out/prof; rm -rf $P gpurun_out/summary; mkdir -p $P
B="python bench.py --frames $FRAMES --warmup 1 --no-cpu-baseline --no-extra --no-host-api"
run() { d=$P/$1; shift; mkdir -p $d; timeout 500 rocprofv3 "$@" > $d/bench.json 2> $d/err.log; echo "$d rc $?"; }
run decode_kt --kernel-trace --stats --output-format csv -d $P/decode_kt -- $B --steps 3 --compress-frames 0
run decode_fetch --pmc FETCH_SIZE --output-format csv -d $P/decode_fetch -- $B --steps 1 --compress-frames 0
run decode_write --pmc WRITE_SIZE --output-format csv -d $P/decode_write -- $B --steps 1 --compress-frames 0
run decode_sq --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $P/decode_sq -- $B --steps 1 --compress-frames 0
run decode_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $P/decode_sq2 -- $B --steps 1 --compress-frames 0
run decode_sq3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $P/decode_sq3 -- $B --steps 1 --compress-frames 0
run decode_tcc --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $P/decode_tcc -- $B --steps 1 --compress-frames 0
run compress_kt --kernel-trace --stats --output-format csv -d $P/compress_kt -- $B --steps 4 --config compress
run compress_tcc --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $P/compress_tcc -- $B --steps 1 --warmup 0 --config compress
run compress_fetch --pmc FETCH_SIZE --output-format csv -d $P/compress_fetch -- $B --steps 1 --warmup 0 --config compress
run compress_write --pmc WRITE_SIZE --output-format csv -d $P/compress_write -- $B --steps 1 --warmup 0 --config compress
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/dict_kt -- python bench.py --config dict --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $P/dict_kt.err; echo "dict_kt rc $?"
# frames of several blocks (2 048 x 1 MiB): the several-block mode's kernels by name
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/blocks_kt -- python tests/multiblock_rate.py 2048 1024 > $P/blocks_kt.json 2> $P/blocks_kt.err; echo "blocks_kt rc $?"
python tests/prof_traffic.py $P $TAG $FRAMES
f=$(find $P/blocks_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" > gpurun_out/summary/${TAG}_blocks_2048x1MiB_kernel_stats.csv
for d in decode_kt compress_kt; do cp $P/$d/bench.json gpurun_out/summary/${TAG}_bench_under_rocprof_${d%_kt}_$FRAMES.json; f=$(find $P/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" > gpurun_out/summary/${TAG}_${d}_kernel_stats.csv; done
rm -rf $P
ls -la gpurun_out/summary
