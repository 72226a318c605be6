for cfg in "1024 2 64" "2048 2 64" "4096 2 64" "512 2 64" "1024 4 64" "1024 2 8" "1024 2 256" "2048 1 256" "1024 1 1024"; do
set -- $cfg
echo "== blocks $1 passes $2 spins $3"
VOFE_SEL_BLOCKS=$1 VOFE_SEL_PASSES=$2 VOFE_SEL_SPINS=$3 timeout -k 10 120 python tools/gftt_bench.py 2>&1 | grep -E "^(malaga|stress|kitti )" | sed -E 's/kernels.*select_round ([0-9.]+) x([0-9.]+).*/select_round \1 x\2/' || exit 1
done
