#!/bin/sh
# builds kernel variants next to the product library, for A/B measurement on the GPU with ZHIP_LIB=<path> (profiles/runs/run_r02*.sh):
#   libzstd_hip_k2l{30,15,7}.so   -DZP_K2_LANES=n     K2: n frames per wave -> 2 / 4 / 8 one-wave workgroups per CU instead of 1   (r02c: all slower)
#   libzstd_hip_huf{16,4}.so      -DZP_HUF_FRAMES=n   K1b: n frames per wave -> 3 / 12 workgroups per CU instead of 6
#   libzstd_hip_k3d{4,2}.so       -DZP_K3D_MINWAVES=n K3's dictionary instantiation with 4 / 2 waves per SIMD instead of 3 (r02y: 89 / 111 against 113 GB/s)
#   libzstd_hip_k3w3.so           -DZP_K3_MINWAVES=3  K3 (no dictionary) at three waves per SIMD / 168 registers (r02za)
#   libzstd_hip_longone.so        -DZP_K3_LONGONE     K3: one ready long match per dependency round (round 1's form)
#   libzstd_hip_nogld.so          -DZP_K3_NO_GLD      K3: the exact (predicated) piece loads from global memory everywhere (r02n)
#   libzstd_hip_zqf{1,0}.so       -DZQ_FENCES=n       K2: fewer / no scheduling fences around the hand-placed pipeline sections (r02q)
#   libzstd_hip_e1l{16,32,64}.so  -DZE_E1_LANES=n     lane-serial match kernel (dictionary / fast-strategy batches): n frames per wave instead of 8
#   libzstd_hip_pf.so             -DZP_K3_PREFETCH    K3: the next batch's far-match source lines touched a batch ahead
#   libzstd_hip_asm2k.so          -DZP_ASM_BYTES=2048 K3: 2 KiB batch assembly buffer (3.7 KiB of LDS per wave instead of 5.7)
#   libzstd_hip_co{36,40,44,48}.so  asm2k + -DZP_K2_LANES=n: a K2 wave of n frames and sixteen (fourteen, ...) K3 waves fit one CU together
# round 3 (profiles/r03*): -DZP_K3_MINWAVES=4|6 (default 5 = 96 VGPRs), -DZP_LIT_SHORT=32 -DZP_FAR_SHORT=32 (own-lane items up to 32 bytes: the round-2 shape, 113 VGPRs),
#   -DZP_K3_SORTED_ORDER (K3 takes frames in K2's order: slower), -DZP_K3_NT=1|3 (streaming loads: no gain / slower), -DZP_K3_DIAG_NOFAR (diagnostic, wrong bytes),
#   -DZQ_FRAMES=10..14 -DZP_ASM_BYTES=2048 [-DZP_K2_PRIO=3] with ZHIP_SPLIT=1 (co-resident K2 + K3: slower), -DZP_HUF_FRAMES=10|12|16 (K1b shapes: unchanged)
# All are emulator-verified (tests/test_emu_kernels.py: test_decode_shape_variants_stay_correct, test_experimental_kernel_variants_stay_correct).
# usage: build_variants.sh [name ...]   (no names: all)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
B="$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared"
want() { [ $# -eq 0 ] && return 0; }
build() { name=$1; shift; $B "$@" -o libzstd_hip_$name.so zhip_lib.hip; }
ALL="k2l30 k2l15 k2l7 huf16 huf4 longone asm2k co36 co40 co44 co48 pf co40p co44p co48p basep e1l16 e1l32 e1l64"
[ $# -gt 0 ] && ALL="$*"
for v in $ALL; do
  case $v in
    k2l30) build k2l30 -DZP_K2_LANES=30 & ;;
    k2l15) build k2l15 -DZP_K2_LANES=15 & ;;
    k2l7) build k2l7 -DZP_K2_LANES=7 & ;;
    huf16) build huf16 -DZP_HUF_FRAMES=16 & ;;
    huf4) build huf4 -DZP_HUF_FRAMES=4 & ;;
    longone) build longone -DZP_K3_LONGONE & ;;
    k3w3) build k3w3 -DZP_K3_MINWAVES=3 & ;;
    k3d4) build k3d4 -DZP_K3D_MINWAVES=4 & ;;
    k3d2) build k3d2 -DZP_K3D_MINWAVES=2 & ;;
    nogld) build nogld -DZP_K3_NO_GLD & ;;
    zqf1) build zqf1 -DZQ_FENCES=1 & ;;
    zqf0) build zqf0 -DZQ_FENCES=0 & ;;
    co40p) build co40p -DZP_ASM_BYTES=2048 -DZP_K2_LANES=40 -DZP_K2_PRIO=3 & ;;
    co44p) build co44p -DZP_ASM_BYTES=2048 -DZP_K2_LANES=44 -DZP_K2_PRIO=3 & ;;
    co48p) build co48p -DZP_ASM_BYTES=2048 -DZP_K2_LANES=48 -DZP_K2_PRIO=3 & ;;
    basep) build basep -DZP_K2_PRIO=3 & ;;
    e1l16) build e1l16 -DZE_E1_LANES=16 & ;;
    e1l32) build e1l32 -DZE_E1_LANES=32 & ;;
    e1l64) build e1l64 -DZE_E1_LANES=64 & ;;
    pf) build pf -DZP_K3_PREFETCH & ;;
    asm2k) build asm2k -DZP_ASM_BYTES=2048 & ;;
    co36) build co36 -DZP_ASM_BYTES=2048 -DZP_K2_LANES=36 & ;;
    co40) build co40 -DZP_ASM_BYTES=2048 -DZP_K2_LANES=40 & ;;
    co44) build co44 -DZP_ASM_BYTES=2048 -DZP_K2_LANES=44 & ;;
    co48) build co48 -DZP_ASM_BYTES=2048 -DZP_K2_LANES=48 & ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
wait
ls -la libzstd_hip_*.so
