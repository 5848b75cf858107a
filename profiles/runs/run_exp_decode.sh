#!/bin/sh
# GPU check of the CPython extension's compute paths
cd /root/repo
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_cext_backend.py -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/exp_cext.log 2>&1; cat gpurun_out/exp_cext.log
