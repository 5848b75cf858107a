#!/bin/bash
# round 6, GPU session ZM: is the bimodal 8 192-frame host-buffer decompress figure (20 or 30 GB/s run to run, rounds 5 and 6) the runtime's hardware queues? The same calls with
# GPU_MAX_HW_QUEUES at its default (4) and at 8 / 16, five processes each (standalone tool: host_api_rate.py 8192)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06zm && O=gpurun_out/r06zm
export TMPDIR=/tmp
for i in 1 2 3 4 5; do for Q in default 8 16; do
  if [ $Q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
  timeout 600 python tests/host_api_rate.py 8192 2>/dev/null | tail -1 | python -c "
import sys,re; l=sys.stdin.read(); m=re.search(r'\"decompress_GBps\": ([0-9.]+), \"compress_GBps\": ([0-9.]+)', l); print('queues=$Q 8192 frames: decompress', m.group(1), 'compress', m.group(2))" | tee -a $O/hw_queues.txt
done; done
