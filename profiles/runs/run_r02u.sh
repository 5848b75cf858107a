# r02u: the Python-visible multi_*_to_buffer calls (PCIe inclusive) with the round's final decode kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out; mkdir -p $O
timeout 900 python tests/host_api_rate.py 65536 > $O/r02u_host_api_65536.log 2>&1; tail -3 $O/r02u_host_api_65536.log
