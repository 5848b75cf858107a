#!/bin/bash
# round 6, GPU session X: E2's three state chains with a full round (64 steps) unrolled -- the loop's counter, bound check and address arithmetic were 5 of its 17 instructions, and three
# busy lanes pay for each: 11.5 per step -- against the one-step loop (-DZE_CHAIN_UNROLL=0). Parity (compress tests, every frame against libzstd), then compress + dictionary runs alternating.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06x && O=gpurun_out/r06x
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_compress.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_compress_bare_states.txt
for i in 1 2; do for V in chain1 product; do
  L=$PWD/python-zstandard_amd/csrc/libzstd_hip.so; [ $V != product ] && L=$PWD/python-zstandard_amd/csrc/libzstd_hip_$V.so
  ZHIP_LIB=$L timeout 600 python bench.py --config compress --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V 65536', d['value'], d['ms_per_step'], {n.replace('zhip_encode_','').replace('_kernel',''): round(v['avg_ms'],2) for n,v in d['kernels'].items()})" | tee -a $O/e2_chain_bare_states_ab.txt
  ZHIP_LIB=$L timeout 600 python bench.py --config dict --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V dict', d['value'], d['ms_per_step'], {n.replace('zhip_encode_','').replace('_kernel',''): round(v['avg_ms'],2) for n,v in d['kernels'].items()})" | tee -a $O/e2_chain_bare_states_ab.txt
done; done
