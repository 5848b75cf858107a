#!/bin/sh
# GPU experiment: the default bench line
cd /root/repo
mkdir -p gpurun_out
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/exp_full.json 2> gpurun_out/exp_full.err
cut -c80-130 gpurun_out/exp_full.json; python -c "
import json; d=json.load(open('gpurun_out/exp_full.json')); print(d['compress'])"; tail -n 3 gpurun_out/exp_full.err
