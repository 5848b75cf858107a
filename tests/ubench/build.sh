#!/bin/sh
# builds tests/ubench/ubench (gfx950 microbenchmarks; test infrastructure, not part of the product)
set -e
cd "$(dirname "$0")"
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O2 -std=c++17 -o ubench ubench.hip
