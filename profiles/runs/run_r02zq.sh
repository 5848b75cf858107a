# r02zq: small compress batches -- the LDS-source match kernel against the flat kernel; then the compress + boundary GPU tests on the new default
mkdir -p gpurun_out
timeout 300 python tests/small_batch_latency.py > gpurun_out/r02zq_lds.json 2> gpurun_out/r02zq_lds.err; tail -1 gpurun_out/r02zq_lds.json
ZHIP_E1LDS_MAX=0 timeout 300 python tests/small_batch_latency.py > gpurun_out/r02zq_flat.json 2> gpurun_out/r02zq_flat.err; tail -1 gpurun_out/r02zq_flat.json
ZHIP_E1LDS_MAX=4096 timeout 300 python tests/small_batch_latency.py > gpurun_out/r02zq_lds4096.json 2> gpurun_out/r02zq_lds4096.err; tail -1 gpurun_out/r02zq_lds4096.json
timeout 600 python -m pytest tests/test_gpu_compress.py tests/test_cext_backend.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r02zq_pytest.log
