#!/bin/bash
# round 6, GPU session ZY: a survey of the decode step over frame kinds no bench line times (levels, no content size, raw / RLE blocks, 16 KiB frames): is there another cliff like the checksums'?
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zy && O=gpurun_out/r06zy
export TMPDIR=/tmp
timeout 1500 python tests/tools/decode_kinds_survey.py 8192 2>&1 | tail -1 | tee $O/decode_kinds_survey.txt
