#!/bin/bash
# round 4, GPU session ZD: four probes per trip in the flat search (ze_dfast_flat_np) -- parity through the GPU compress tests with both kernels
# switched to it, the match kernel's time by batch size (two against four probes), one-shot / small-batch latencies with the LDS-source kernel
# at four probes, the host API at 8 192 sources
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04zd && O=gpurun_out/r04zd
export TMPDIR=/tmp
( time ZHIP_E1L_PROBES=4 ZHIP_FLAT4_MAX=262144 timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_boundary.py -m gpu -x -q > $O/pytest_gpu_four_probes.txt 2>&1 ) 2> $O/pytest.time; tail -3 $O/pytest_gpu_four_probes.txt
timeout 600 python tests/tools/flat4_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/flat4_sweep.txt
for p in 2 4; do ZHIP_E1L_PROBES=$p timeout 600 python tests/small_batch_latency.py > $O/small_batch_latency_probes$p.txt 2>&1; tail -12 $O/small_batch_latency_probes$p.txt | cut -c1-200; done
for m in 0 8192; do ZHIP_FLAT4_MAX=$m timeout 600 python tests/host_api_rate.py 8192 > $O/host_api_8192_flat4max$m.log 2>&1; tail -1 $O/host_api_8192_flat4max$m.log | cut -c1-300; done
