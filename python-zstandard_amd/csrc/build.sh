#!/bin/sh
# builds python-zstandard_amd/csrc/libzstd_hip.so for gfx950 (cross-compiles without a GPU)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o libzstd_hip.so zhip_lib.hip "$@"
