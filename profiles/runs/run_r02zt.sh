# r02zt: batches of small sources -- LDS-source match kernel with the LDS area sized for the batch (2 / 4 / 8 rounds of resident frames) against the flat kernel
mkdir -p gpurun_out
for r in 2 4 8; do ZHIP_E1LDS_ROUNDS=$r timeout 200 python tests/small_source_batches.py > gpurun_out/r02zt_rounds$r.json 2> gpurun_out/r02zt_rounds$r.err; tail -1 gpurun_out/r02zt_rounds$r.json; done
ZHIP_E1LDS_MAX=0 timeout 200 python tests/small_source_batches.py > gpurun_out/r02zt_flat.json 2> gpurun_out/r02zt_flat.err; tail -1 gpurun_out/r02zt_flat.json
timeout 300 python -m pytest tests/test_gpu_compress.py -m gpu -x -q 2>&1 | tail -2
