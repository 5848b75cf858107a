# r02t: entropy-kernel phase timers after the wave-parallel table builders
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
ZHIP_PROF=1 timeout 300 python bench.py --config compress --frames 32768 --compress-frames 32768 --no-cpu-baseline --steps 1 --warmup 1 > $O/r02t_prof.json 2> $O/r02t_prof.err; grep "zhip-prof" $O/r02t_prof.err | grep -A9 "E2:" | tail -10
