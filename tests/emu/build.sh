#!/bin/sh
# builds tests/emu/libzhip_emu.so : product kernel sources compiled for the host wave emulator (debug aid)
set -e
cd "$(dirname "$0")"
g++ -O1 -g -fPIC -shared -std=c++17 -I. -Wall -Wno-unused-function -Wno-unused-variable -o libzhip_emu.so zhemu.cpp emu_kernels.cpp
