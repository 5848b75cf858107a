#!/bin/bash
# round 6, GPU session V (diagnostic build: git tag r06-k1-fine-timers, variant k1fine there): K1's phase timers in finer parts (-DZD_PROF_FINE, ZHIP_PROF=1): where its ~20 K instructions per frame go
# (slots as printed: "K3 load+scan" = the Huffman weights' distribution + FSE table, "-" = the weights' FSE decode on lane 0, "K3 literal/far fetch" = the Huffman table's build,
#  "K3 near rounds" = the three sequence distributions on lane 0, "K3 flush" = their FSE tables' builds; huf-table / seq-header+fse keep the rest)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r06v && O=gpurun_out/r06v
export TMPDIR=/tmp
ZHIP_LIB=$PWD/python-zstandard_amd/csrc/libzstd_hip_k1fine.so ZHIP_PROF=1 timeout 600 python bench.py --config decompress --compress-frames 0 --no-extra --no-host-api --no-cpu-baseline --steps 1 --warmup 0 2>&1 >/dev/null | grep "zhip-prof" | head -24 | tee $O/k1_fine_phase_timers.txt
