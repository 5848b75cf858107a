# r01e evidence (run on the GPU box through gpurun): the un-profiled bench lines, then rocprofv3 kernel traces for both directions
# and the FETCH_SIZE / WRITE_SIZE passes of the compress direction, one pass per purpose as the guide prescribes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r01e_bench_full_65536.json 2> gpurun_out/r01e_bench_full.err
timeout 400 python bench.py --direction compress --steps 3 --warmup 1 > gpurun_out/r01e_bench_compress_65536.json 2> gpurun_out/r01e_bench_compress.err
P=gpurun_out/profe; rm -rf $P; mkdir -p $P/ktd $P/ktc $P/fetch $P/write $P/dfetch $P/dwrite
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/ktd -- python bench.py --frames 32768 --steps 3 --warmup 1 --no-cpu-baseline --compress-frames 0 > $P/ktd/bench.json 2> $P/ktd/err.log
BC="python bench.py --direction compress --frames 32768 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/ktc -- $BC --steps 2 --warmup 1 > $P/ktc/bench.json 2> $P/ktc/err.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- $BC --steps 1 --warmup 0 > $P/fetch/bench.json 2> $P/fetch/err.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- $BC --steps 1 --warmup 0 > $P/write/bench.json 2> $P/write/err.log
BD="python bench.py --frames 32768 --no-cpu-baseline --compress-frames 0"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/dfetch -- $BD --steps 1 --warmup 1 > $P/dfetch/bench.json 2> $P/dfetch/err.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/dwrite -- $BD --steps 1 --warmup 1 > $P/dwrite/bench.json 2> $P/dwrite/err.log
python tests/prof_summarize.py $P | tail -16
find $P -name "*.csv" ! -name "*.zhip.csv" -delete
for d in fetch write dfetch dwrite; do echo "== $d"; for f in $(find $P/$d -name "*counter_collection.zhip.csv"); do python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(float); cnt = collections.Counter()
for r in rows:
    k = (r.get("Kernel_Name", "?")[:40], r.get("Counter_Name", "?"))
    acc[k] += float(r.get("Counter_Value", 0)); cnt[k] += 1
for k in sorted(acc): print(k[0], k[1], "total=%.4g" % acc[k], "dispatches=%d" % cnt[k])
PY
done; done
cat gpurun_out/r01e_bench_full_65536.json gpurun_out/r01e_bench_compress_65536.json | cut -c1-2600
cat $P/ktd/bench.json $P/ktc/bench.json | cut -c1-200
du -sh gpurun_out
