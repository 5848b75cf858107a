#!/bin/bash
# round 6, GPU session ZZA: a survey of the compress step (level 3) over source kinds no bench line times
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zza && O=gpurun_out/r06zza
export TMPDIR=/tmp
timeout 1500 python tests/tools/compress_kinds_survey.py 2>&1 | tail -1 | tee $O/compress_kinds_survey.txt
