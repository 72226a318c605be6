#!/bin/bash
# A/B of non-temporal stores per streaming kernel (diagnostic libraries built with -DVOFE_NT_<KERNEL>): the bench's
# HIP-event-timed streaming rooflines and headline, and the detector's per-kernel split.
cd "$(dirname "$0")/.."
for v in base COPY PYRDOWN SCORE NMS; do
  lib=visual-odometry_amd/libvofe.so; [ $v != base ] && lib=tools/diag/libvofe_nt_$v.so
  echo "== $v"
  VOFE_LIB=$lib timeout -k 10 240 python bench.py --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'])
for e in d['roofline_streaming']: print(' ', e['kernel'], e['avg_launch_ms'], e['achieved'], e['frac'])
" || exit 1
  VOFE_LIB=$lib timeout -k 10 120 python tools/gftt_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 || exit 1
done
