# r02o: K3 phase timers after the piece-load changes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
ZHIP_PROF=1 timeout 300 python bench.py --config decompress --frames 32768 --no-cpu-baseline --steps 1 --warmup 1 > $O/r02o_prof.json 2> $O/r02o_prof.err; grep zhip-prof $O/r02o_prof.err | tail -30
