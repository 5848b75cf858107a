#!/bin/bash
# round 6, GPU session ZZJ: the several-block flat search from 4 096 sources per 256 KiB (was 8 192): the whole GPU suite, the blocks stress, the shapes around the threshold once more (product only)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zzj && O=gpurun_out/r06zzj
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | cut -c1-300 | tee $O/pytest_gpu.txt
timeout 900 python tests/stress_gpu_blocks.py 6207 2>&1 | tail -1 | cut -c1-600 | tee $O/stress_gpu_blocks.txt
for sh in 4096:256 6144:256 8192:512 4096:512; do shape="${sh/:/ }"
  timeout 600 python tests/multiblock_rate.py $shape 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$shape', 'compress', d['compress_GBps'], 'GB/s', d['compress_ms'], 'ms  bit-exact', d['bit_exact_vs_libzstd'], ' decompress', d['decompress_GBps'])" | tee -a $O/mbc_after.txt
done
