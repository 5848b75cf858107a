#!/bin/bash
# round 6, GPU session ZU: KX's "a frame carries a checksum" word written once per K1 wave (not per frame): the GPU suite, the checksum cost of both directions on the final build
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zu && O=gpurun_out/r06zu
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout 900 python tests/tools/decode_checksum_cost.py 65536 2>/dev/null | tail -1 | sed 's/^/decode /' | tee $O/checksum_cost.txt
timeout 900 python tests/tools/decode_checksum_cost.py 16384 2>/dev/null | tail -1 | sed 's/^/decode /' | tee -a $O/checksum_cost.txt
timeout 900 python tests/tools/compress_checksum_cost.py 32768 2>/dev/null | tail -1 | sed 's/^/compress /' | tee -a $O/checksum_cost.txt
