# compress-direction profile collection (run on the GPU box through gpurun): kernel trace + stats, then one PMC pass per counter set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
F=${PROF_FRAMES:-32768}
P=gpurun_out/profc; rm -rf $P; mkdir -p $P/kt $P/fetch $P/write $P/sq $P/tcc
B="python bench.py --direction compress --frames $F --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- $B --steps 2 --warmup 1 > $P/kt/bench.json 2> $P/kt/err.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- $B --steps 1 --warmup 0 > $P/fetch/bench.json 2> $P/fetch/err.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- $B --steps 1 --warmup 0 > $P/write/bench.json 2> $P/write/err.log
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $P/sq -- $B --steps 1 --warmup 0 > $P/sq/bench.json 2> $P/sq/err.log
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $P/tcc -- $B --steps 1 --warmup 0 > $P/tcc/bench.json 2> $P/tcc/err.log
python tests/prof_summarize.py $P | tail -12
find $P -name "*.csv" ! -name "*.zhip.csv" -delete
for d in fetch write sq tcc; do echo "== $d"; for f in $(find $P/$d -name "*counter_collection.zhip.csv"); do python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    k = (r.get("Kernel_Name", "?")[:40], r.get("Counter_Name", "?"))
    acc[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
for k in sorted(acc): print(k[0], k[1], "total=%.4g" % acc[k], "dispatches=%d" % cnt[k])
PY
done; done
tail -2 $P/tcc/err.log
cat $P/kt/bench.json | cut -c1-1200
du -sh gpurun_out
