#!/bin/bash
# Parity soak on the GPU: the two randomised differential sweeps of tests/test_gpu_parity.py with fresh seeds and
# many more cases than the test suite runs (usage: tools/soak.sh [cases] [seed ...]).
cd "$(dirname "$0")/.."
cases=${1:-400}; shift
for seed in "${@:-11 12 13}"; do
  for s in $seed; do
    echo "== seed $s, $cases cases per sweep"
    VOFE_SWEEP_CASES=$cases VOFE_SWEEP_SEED=$s timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py -x -q \
        --timeout 800 --timeout-method thread -k "${SOAK_TESTS:-randomised}" 2>&1 | tail -3 || exit 1
  done
done
