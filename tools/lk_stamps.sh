#!/bin/bash
# Builds the diagnostic library (s_memtime stamps per phase of lk_track_kernel<R,WC>; never shipped) for tools/lk_stamps.py.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/diag
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-value \
    -Xclang -target-feature -Xclang -dot4-insts -DLK_STAMPS -o tools/diag/libvofe_stamps.so \
    visual-odometry_amd/csrc/{api,pyramid,lk,corners,bookkeeping}.hip
echo tools/diag/libvofe_stamps.so
