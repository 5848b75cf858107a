#!/bin/sh
# builds the gfx950 microbenchmarks (test infrastructure, not part of the product): ubench (wave primitives), membench (random gathers, the
# lane-per-frame executor mock), tablebench (the flat match kernel's table traffic)
set -e
cd "$(dirname "$0")"
for b in ubench membench tablebench allocbench; do ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -o $b $b.hip; done
