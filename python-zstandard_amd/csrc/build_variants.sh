#!/bin/sh
# builds kernel variants next to the product library, for A/B measurement on the GPU with ZHIP_LIB=<path>. What is left to vary at build time
# after round 5's pruning (the forms that lost in rounds 1-4 -- lane-per-frame K2, co-resident K2 + K3, K3 / K1b rewrites, the link search --
# are in git tag r04-experiments, with their measurements in DESIGN.md section 4 and profiles/):
#   huf{16,10,8,4} -DZP_HUF_FRAMES=n K1b: n frames per wave -> 3 / 4 / 6 / 12 workgroups per CU instead of the product's 4 of 12 frames (rounds 2-5: 8)
#   zqf{1,0}    -DZQ_FENCES=n        K2: fewer / no scheduling fences around the hand-placed pipeline sections
#   zq{9,12}    -DZQ_FRAMES=n        K2: n frames per wave
#   k3w{4,5}    -DZP_K3_MINWAVES=n   K3 at n waves per SIMD (default 6)
#   k3d{2,3}    -DZP_K3D_MINWAVES=n  K3's dictionary instantiation (default 4)
#   asm2k       -DZP_ASM_BYTES=2048  K3: 2 KiB batch assembly buffer
#   own32       -DZP_LIT_SHORT=32 -DZP_FAR_SHORT=32   K3: own-lane items up to 32 bytes (round 2's shape)
#   nohist      -DZP_HIST_KEEP=0u -DZP_HIST_SLIDE=0u  K3 without the LDS history in front of the batch (round 6 A/B)
#   nok0        -DZHIP_K0=0          decode pipeline without K0 (the lane-per-frame parser pass in front of K1)
#   dchunk32k   -DZHIP_DCHUNK=32768  decode pipeline in chunks of 32 768 frames (two slot streams) instead of one of 65 536
#   e1l{16,32,64} -DZE_E1_LANES=n    lane-serial match kernel (fast-strategy batches): n frames per wave instead of 8
# Emulator-verified shapes: tests/test_emu_kernels.py::test_decode_shape_variants_stay_correct.
# usage: build_variants.sh name [name ...]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
B="$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared"
build() { name=$1; shift; $B "$@" -o libzstd_hip_$name.so zhip_lib.hip; }
for v in "$@"; do
  case $v in
    huf16) build huf16 -DZP_HUF_FRAMES=16 & ;;
    huf4) build huf4 -DZP_HUF_FRAMES=4 & ;;
    huf8) build huf8 -DZP_HUF_FRAMES=8 & ;;
    huf10) build huf10 -DZP_HUF_FRAMES=10 & ;;
    zqf1) build zqf1 -DZQ_FENCES=1 & ;;
    zqf0) build zqf0 -DZQ_FENCES=0 & ;;
    zq9) build zq9 -DZQ_FRAMES=9 & ;;
    zq12) build zq12 -DZQ_FRAMES=12 & ;;
    k3w4) build k3w4 -DZP_K3_MINWAVES=4 & ;;
    k3w5) build k3w5 -DZP_K3_MINWAVES=5 & ;;
    k3d2) build k3d2 -DZP_K3D_MINWAVES=2 & ;;
    k3d3) build k3d3 -DZP_K3D_MINWAVES=3 & ;;
    asm2k) build asm2k -DZP_ASM_BYTES=2048 & ;;
    nt1) build nt1 -DZP_K3_NT=1 & ;;                      # K3: sequences and decoded literals read with streaming (nt) loads
    zqnost) build zqnost -DZQ_DIAG_NOSTORE & ;;             # DIAGNOSTIC: K2 without its sequence stores
    zqnt) build zqnt -DZQ_NT_STORE & ;;                    # K2: sequences stored with non-temporal stores
    hufnost) build hufnost -DZP_K1B_DIAG_NOSTORE & ;;      # DIAGNOSTIC: K1b without its literal stores
    hufnt) build hufnt -DZP_K1B_NT_STORE & ;;
    nt1w5) build nt1w5 -DZP_K3_NT=1 -DZP_K3_MINWAVES=5 & ;;
    nt3) build nt3 -DZP_K3_NT=3 & ;;
    own32) build own32 -DZP_LIT_SHORT=32 -DZP_FAR_SHORT=32 & ;;
    nohist) build nohist -DZP_HIST_KEEP=0u -DZP_HIST_SLIDE=0u & ;;   # K3 without the LDS history of its flushed output (rounds 1-5's form; A/B of round 6: profiles/r06g_k3_history_*)
    floor) build floor -DZP_K3_DIAG_FLOOR & ;;
    floorw4) build floorw4 -DZP_K3_DIAG_FLOOR -DZP_K3_MINWAVES=4 -DZP_FLOOR_LDSPAD=4096 & ;;                                  # the occupancy of an 8 KiB buffer, no window
    floorw4win) build floorw4win -DZP_K3_DIAG_FLOOR -DZP_K3_MINWAVES=4 -DZP_FLOOR_LDSPAD=4096 -DZP_FLOOR_WIN=8192 -DZP_ASM_BYTES=2048 & ;;
    floorw5) build floorw5 -DZP_K3_DIAG_FLOOR -DZP_K3_MINWAVES=5 -DZP_FLOOR_LDSPAD=2048 & ;;
    floorw5win) build floorw5win -DZP_K3_DIAG_FLOOR -DZP_K3_MINWAVES=5 -DZP_FLOOR_LDSPAD=2048 -DZP_FLOOR_WIN=6144 -DZP_FLOOR_KEEP=3072 -DZP_ASM_BYTES=2048 & ;;
    floorw6win) build floorw6win -DZP_K3_DIAG_FLOOR -DZP_FLOOR_WIN=4096 -DZP_FLOOR_KEEP=2048 -DZP_ASM_BYTES=1536 & ;;
    floorw3) build floorw3 -DZP_K3_DIAG_FLOOR -DZP_K3_MINWAVES=3 -DZP_FLOOR_LDSPAD=8192 & ;;
    floorw2) build floorw2 -DZP_K3_DIAG_FLOOR -DZP_K3_MINWAVES=2 -DZP_FLOOR_LDSPAD=14336 & ;;
    floorw4free) build floorw4free -DZP_K3_DIAG_FLOOR -DZP_K3_MINWAVES=4 -DZP_FLOOR_LDSPAD=4096 -DZP_FLOOR_WIN=8192 -DZP_FLOOR_KEEP=5120 -DZP_FLOOR_FREE & ;;        # an ideal 8 KiB ring: ~5-7 KiB behind the batch, no LDS work of its own
    floorw6free) build floorw6free -DZP_K3_DIAG_FLOOR -DZP_FLOOR_WIN=8192 -DZP_FLOOR_KEEP=5120 -DZP_FLOOR_FREE & ;;                                                 # the same requests dropped at six waves per SIMD (not buildable: shows what occupancy is worth)
    floorw3free) build floorw3free -DZP_K3_DIAG_FLOOR -DZP_K3_MINWAVES=3 -DZP_FLOOR_LDSPAD=8192 -DZP_FLOOR_WIN=12288 -DZP_FLOOR_KEEP=9216 -DZP_FLOOR_FREE & ;;
    floorw2free) build floorw2free -DZP_K3_DIAG_FLOOR -DZP_K3_MINWAVES=2 -DZP_FLOOR_LDSPAD=14336 -DZP_FLOOR_WIN=20480 -DZP_FLOOR_KEEP=17408 -DZP_FLOOR_FREE & ;;
    chain1) build chain1 -DZE_CHAIN_UNROLL=0 & ;;          # E2's state chains one step per loop trip (rounds 2-5: 17 instructions per step; round 6 unrolls a full round: 11.5)
    e2prof) build e2prof -DZE_PROF_STREAM & ;;             # E2 with its sequence-stream rounds timed in four parts (ZHIP_PROF=1)
    e2w3) build e2w3 -DZE_E2_MINWAVES=3 & ;;               # E2 (entropy stage) at three / five waves per SIMD (default 4: 127 VGPRs + 160 bytes of scratch)
    e2w5) build e2w5 -DZE_E2_MINWAVES=5 & ;;
    e1lds4) build e1lds4 -DZHIP_E1LDS_PROBES=4 & ;;        # the LDS-source match kernel (small batches) with four probes per trip
    noepoch) build noepoch -DZHIP_TABLE_EPOCHS=0 & ;;      # the flat searches with their tables zeroed every launch (rounds 1-5; A/B of round 6: profiles/r06ze_*, r06zg_*)
    nofastwide) build nofastwide -DZHIP_FAST_WIDE=0 & ;;   # fast-strategy batches in chunks of 32 768 at eight sources per wave (rounds 2-5)
    notrailer) build notrailer -DZHIP_TRAILER_LATER=0 & ;;   # compress with write_checksum: the entropy kernel hashes the source on one lane (rounds 1-5; A/B of round 6: profiles/r06zt_*)
    noside) build noside -DZHIP_SIDE=0 -DZHIP_K0=0 & ;;    # the decode step as rounds 1-5 ran it: one stream, no K0
    pick3) build pick3 -DZHIP_PICK_CANDIDATES=3 & ;;       # the placement pick over three candidates (rounds 5-6 until r06zzv)
    pick48k) build pick48k -DZHIP_PICK_MIN=49152 & ;;      # the placement pick from 49 152 sources per launch on only (rounds 5-6 until r06zzt: the host-buffer API's chunks of 32 768 never picked)
    pickstudy24) build pickstudy24 -DZHIP_PICK_STUDY=24 & ;;   # the same over 24 candidates (whole launches for the first eight)
    pickstudy) build pickstudy -DZHIP_PICK_STUDY=1 & ;;   # DIAGNOSTIC: the placement pick prints eight candidates' times, whole launches and probes over the sources' first bytes
    nok0) build nok0 -DZHIP_K0=0 & ;;                    # decode without K0: K1's lane 0 parses the Huffman weights and the sequence distributions itself (rounds 1-5; A/B of round 6: profiles/r06w_*)
    dchunk32k) build dchunk32k -DZHIP_DCHUNK=32768 & ;;    # decode: chunks of 32 768 frames on the slot streams (round 6 A/B with K1b beside K2: profiles/r06q_*)
    e1l16) build e1l16 -DZE_E1_LANES=16 & ;;
    e1l32) build e1l32 -DZE_E1_LANES=32 & ;;
    e1l64) build e1l64 -DZE_E1_LANES=64 & ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done
wait
ls -la libzstd_hip_*.so 2>/dev/null || true
