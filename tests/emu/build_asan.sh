#!/bin/sh
# AddressSanitizer build of the emulated kernels (debug aid): out-of-bounds reads / writes of the kernel sources show up on the host.
#   sh tests/emu/build_asan.sh && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
#       ZHIP_EMU_SO=/tmp/libzhip_emu_asan.so python tests/stress_emu_encode.py 1 p 3
set -e
cd "$(dirname "$0")"
g++ -O1 -g -fPIC -shared -std=c++17 -I. -fsanitize=address -fno-omit-frame-pointer -Wno-unused-function -Wno-unused-variable -o ${1:-/tmp/libzhip_emu_asan.so} zhemu.cpp emu_kernels.cpp
