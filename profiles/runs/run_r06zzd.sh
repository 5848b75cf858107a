cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06zzd
for i in 1 2 3 4; do timeout 900 python -m pytest tests/test_gpu_compress.py -x -q -m gpu -k "table_placement_pick" 2>&1 | tail -25 | cut -c1-300 | tee -a gpurun_out/r06zzd/pick_test_reruns.txt; done
