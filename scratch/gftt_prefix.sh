#!/bin/bash
# first-prefix factor (VOFE_SEL_PREFIX) x whole-list ratio (VOFE_SEL_FULL) over the detector workloads
cd "$(dirname "$0")/.."
for pf in 2 3 4; do for full in 1 2 3 4 6; do
  echo "== VOFE_SEL_PREFIX=$pf VOFE_SEL_FULL=$full"
  VOFE_SEL_PREFIX=$pf VOFE_SEL_FULL=$full timeout -k 10 100 python3 scratch/gftt_frames.py 2>&1 | grep -E "frame 0|frame 3|cycling" || exit 1
  VOFE_SEL_PREFIX=$pf VOFE_SEL_FULL=$full timeout -k 10 100 python3 scratch/gftt_bench.py 2>&1 | grep -E "GPU" | sed 's/(device.*oracle/ oracle/' || exit 1
done; done
