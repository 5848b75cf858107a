# r02b: boundary work (dictionary type, format, explicit parameters, cext as the implementation) on hardware
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02b_pytest.log
tail -25 gpurun_out/r02b_pytest.log
