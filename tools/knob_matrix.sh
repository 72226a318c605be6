#!/bin/bash
# The detector's result must not depend on the selection schedule: run its GPU tests under extreme knob settings.
cd "$(dirname "$0")/.."
for cfg in "1 1 1 0 1 1" "7 3 1 1 1 3" "64 8 8 0 4 1000000" "4096 1 1 200 3 2" "1024 2 4 4 2 6"; do
  set -- $cfg
  echo "== VOFE_SEL_BLOCKS=$1 LAUNCHES=$2 PASSES=$3 SPINS=$4 PREFIX=$5 FULL=$6"
  VOFE_SEL_BLOCKS=$1 VOFE_SEL_LAUNCHES=$2 VOFE_SEL_PASSES=$3 VOFE_SEL_SPINS=$4 VOFE_SEL_PREFIX=$5 VOFE_SEL_FULL=$6 \
    timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -x -q --timeout 500 --timeout-method thread \
    -k "gftt or real_texture or full_size or handle_candidates" 2>&1 | tail -2 || exit 1
done
