#!/bin/bash
# round 6, GPU session ZH: the LDS-source match kernel (small batches: one source per CU) with two / four probes per trip (-DZHIP_E1LDS_PROBES=4): one-shot and small-batch compress latency
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zh && O=gpurun_out/r06zh
export TMPDIR=/tmp
for i in 1 2; do for V in product e1lds4; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python tests/small_batch_latency.py 2>/dev/null | tail -1 | sed "s/^/$V /" | cut -c1-600 | tee -a $O/e1lds_probes_ab.txt
done; done
