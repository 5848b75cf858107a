#!/bin/bash
# round 6, GPU session ZS: KX -- content checksums verified after K3 by a lane per frame -- the decode-side GPU tests, the checksum cost again (r06zr: 6.8 -> 15.7 ms per 16 384
# frames with K3 / K1 hashing on one lane), and the headline (no checksums: KX returns at once)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zs && O=gpurun_out/r06zs
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_decompress.py tests/test_gpu_boundary.py tests/test_gpu_fullsize.py tests/test_cext_backend.py tests/test_gpu_multidevice.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_decode.txt
timeout 900 python tests/tools/decode_checksum_cost.py 16384 2>/dev/null | tail -1 | tee $O/decode_checksum_cost.txt
timeout 900 python tests/tools/decode_checksum_cost.py 65536 2>/dev/null | tail -1 | tee -a $O/decode_checksum_cost.txt
for i in 1 2; do timeout 600 python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('headline', d['value'], d['ms_per_step'], {n.replace('zhip_decode_','').replace('_kernel',''): round(v['avg_ms'],3) for n,v in k.items()})" | tee -a $O/decode_checksum_cost.txt; done
