#!/bin/bash
# Headline A/B of library builds: tools/lib_ab.sh <lib-suffix|-> ...   ("-" = the shipped libvofe.so; suffix w3 = libvofe_w3.so)
# bench.py --no-cpu --no-extras --steps 10 per entry, in the order given; prints frames/s and the tracker launch's average.
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = "-" ]; then unset VOFE_LIB; else export VOFE_LIB=$R/visual-odometry_amd/libvofe_$v.so; fi
  timeout -k 10 200 python3 "$R/bench.py" --no-cpu --no-extras --steps 10 > "$R/gpurun_out/lib_ab.json" 2>/dev/null || exit 1
  python3 - "$v" "$R/gpurun_out/lib_ab.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("variant %-6s %9.1f frames/s   tracker launch %.4f ms" % (sys.argv[1], d["value"], d["roofline"]["avg_launch_ms"]))
PY
done
