#!/bin/bash
# Pyramid chain with and without the pipeline's tail kernel, by batch size (HIP events per kernel, tools/stream_rooflines.py).
# usage: tools/pyr_tail_ab.sh [frames ...]     output: gpurun_out/pyr_tail_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pyr_tail_ab.txt
: > "$out"
for n in ${*:-16 32 64 128}; do
  for mode in tail notail; do
    if [ $mode = notail ]; then export VOFE_PYR_TAIL=0; else unset VOFE_PYR_TAIL; fi
    echo "== $n frames, $mode" >> "$out"
    VOFE_PYRAMID=levels timeout -k 10 120 python3 -u "$R/tools/stream_rooflines.py" $n 2>/dev/null | grep -E "pyr_|pad_|scharr" >> "$out" || exit 1
  done
done
python3 - "$out" <<'PY'
import sys, re
tot = {}
key = None
for ln in open(sys.argv[1]):
    if ln.startswith("=="):
        key = ln.strip("= \n"); tot[key] = 0.0
    else:
        m = re.search(r"([\d.]+) us", ln)
        if m: tot[key] += float(m.group(1))
for k, v in tot.items():
    print("%-24s chain %.1f us" % (k, v))
PY
