#!/bin/bash
# sweep of the selection knobs (read once per process) over the four GFTT workloads of gftt_bench.py
cd /root/repo
for cfg in "1024 1 2 64 2" "1024 1 2 64 2" "1024 1 3 64 2" "1024 1 2 128 2" "1024 2 2 64 2" "1024 1 4 32 2" "1024 1 2 48 2" "2048 1 2 64 2" "512 1 2 64 2"; do
  set -- $cfg
  echo "BLOCKS=$1 LAUNCHES=$2 PASSES=$3 SPINS=$4 PREFIX=$5"
  VOFE_SEL_BLOCKS=$1 VOFE_SEL_LAUNCHES=$2 VOFE_SEL_PASSES=$3 VOFE_SEL_SPINS=$4 VOFE_SEL_PREFIX=$5 timeout -k 10 100 python3 scratch/gftt_bench.py 2>/dev/null | sed 's/(device-resident image, incl. D2H of corners), //' | cut -c1-110 || exit 1
done
