# r02j: table-copy mode on the device -- GPU suite, the reference's own hot-path tests, dictionary bench (with the CPU leg)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/r02j_pytest.log 2>&1; tail -5 $O/r02j_pytest.log
sh tests/run_reference_hotpath_tests.sh > $O/r02j_ref.log 2>&1; tail -15 $O/r02j_ref.log
timeout 600 python bench.py --config dict > $O/r02j_dict.json 2> $O/r02j_dict.err; cat $O/r02j_dict.json
