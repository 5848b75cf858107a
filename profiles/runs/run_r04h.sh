#!/bin/bash
# round 4, GPU session H: the host pipeline's compress chunk slots (concurrent chunks on their own streams); full GPU suite incl. the full-size test; host API rates
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r04h && O=gpurun_out/r04h
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1 ) 2> $O/pytest_gpu.time; tail -6 $O/pytest_gpu.txt; cat $O/pytest_gpu.time | tail -3
for s in 4 2 1; do ZHIP_ESLOTS=$s timeout 600 python tests/host_api_rate.py 65536 > $O/host_api_65536_slots$s.log 2>&1; tail -1 $O/host_api_65536_slots$s.log | cut -c1-400; done
ZHIP_ESLOT_ITEMS=8192 timeout 600 python tests/host_api_rate.py 65536 > $O/host_api_65536_items8192.log 2>&1; tail -1 $O/host_api_65536_items8192.log | cut -c1-300
ZHIP_ESLOT_ITEMS=4096 timeout 600 python tests/host_api_rate.py 8192 > $O/host_api_8192_items4096.log 2>&1; tail -1 $O/host_api_8192_items4096.log | cut -c1-300
ZHIP_ESLOT_ITEMS=2048 timeout 600 python tests/host_api_rate.py 8192 > $O/host_api_8192_items2048.log 2>&1; tail -1 $O/host_api_8192_items2048.log | cut -c1-300
timeout 600 python tests/host_api_rate.py 8192 > $O/host_api_8192.log 2>&1; tail -1 $O/host_api_8192.log | cut -c1-300
