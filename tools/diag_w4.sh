#!/bin/bash
# CPU-side study aid for the spill investigation.  Cross-compiles lk.hip with the two-keypoint
# tracker held at 4 waves per SIMD by the compiler (-DLK_X2_WAVES_PER_EU=4: spills to scratch memory) to ASSEMBLY ONLY, outside
# the repository, so that the spill code can be read against the EXEC state.  It builds no library: the spill build is not
# to be run on the shared GPU pool (its round-2 sibling faulted a GPU and the cause is not established).
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/vofe_diag_w4}
mkdir -p "$OUT"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Xclang -target-feature -Xclang -dot4-insts \
  -DLK_X2_WAVES_PER_EU=4 -Iinclude -Rpass-analysis=kernel-resource-usage --cuda-device-only -S -o "$OUT/lk_w4.s" \
  visual-odometry_amd/csrc/lk.hip 2> "$OUT/w4_resources.txt"
grep -A9 "lk_track_kernel_x2ILi2" "$OUT/w4_resources.txt" | grep -E "VGPRs|Scratch|Spill|Occupancy"
echo "assembly: $OUT/lk_w4.s"
