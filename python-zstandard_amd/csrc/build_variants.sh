#!/bin/sh
# builds the experimental kernel variants next to the product library, for A/B measurement on the GPU with ZHIP_LIB=<path>:
#   libzstd_hip_tab3.so     -DZE_TAB3        entropy kernel: the three sequence tables built by three lanes at once
#   libzstd_hip_longall.so  -DZP_K3_LONGALL  K3: every ready long match of a dependency round handled in that round
# Both are emulator-verified (tests/test_emu_kernels.py::test_experimental_kernel_variants_stay_correct) and not yet measured on hardware.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZE_TAB3 -o libzstd_hip_tab3.so zhip_lib.hip &
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DZP_K3_LONGALL -o libzstd_hip_longall.so zhip_lib.hip &
wait
ls -la libzstd_hip_*.so
