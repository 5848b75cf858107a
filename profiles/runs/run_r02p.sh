# r02p: profile pass for profiles/ with the quad K2 + slimmer K3, then the round's decode bench line
TAG=r02 sh tests/run_profiles.sh
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r02p_bench.json 2> gpurun_out/r02p_bench.err; cat gpurun_out/r02p_bench.json | cut -c1-1500
