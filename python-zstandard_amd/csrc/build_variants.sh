#!/bin/sh
# builds kernel variants next to the product library, for A/B measurement on the GPU with ZHIP_LIB=<path> (tests/run_r02*.sh):
#   libzstd_hip_k2l{30,15,7}.so   -DZP_K2_LANES=n     K2: n frames per wave -> 2 / 4 / 8 one-wave workgroups per CU instead of 1
#   libzstd_hip_huf{8,4}.so       -DZP_HUF_FRAMES=n   K1b: n frames per wave -> 6 / 12 workgroups per CU instead of 3
#   libzstd_hip_longall.so        -DZP_K3_LONGALL     K3: every ready long match of a dependency round handled in that round
#   libzstd_hip_tab3.so           -DZE_TAB3           entropy kernel: the three sequence tables built by three lanes at once
#   libzstd_hip_c1.so / _c2.so    combinations
# All are emulator-verified (tests/test_emu_kernels.py: test_decode_shape_variants_stay_correct, test_experimental_kernel_variants_stay_correct).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
B="$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared"
$B -DZP_K2_LANES=30 -o libzstd_hip_k2l30.so zhip_lib.hip &
$B -DZP_K2_LANES=15 -o libzstd_hip_k2l15.so zhip_lib.hip &
$B -DZP_K2_LANES=7 -o libzstd_hip_k2l7.so zhip_lib.hip &
$B -DZP_HUF_FRAMES=8 -o libzstd_hip_huf8.so zhip_lib.hip &
wait
$B -DZP_HUF_FRAMES=4 -o libzstd_hip_huf4.so zhip_lib.hip &
$B -DZP_K3_LONGALL -o libzstd_hip_longall.so zhip_lib.hip &
$B -DZP_K2_LANES=15 -DZP_HUF_FRAMES=8 -o libzstd_hip_c1.so zhip_lib.hip &
$B -DZP_K2_LANES=7 -DZP_HUF_FRAMES=4 -o libzstd_hip_c2.so zhip_lib.hip &
wait
$B -DZE_TAB3 -o libzstd_hip_tab3.so zhip_lib.hip
ls -la libzstd_hip_*.so
