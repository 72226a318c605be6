for cfg in kitti parking; do
for n in 300 500 1000 1280 1600 2000 3000; do
for k in x1 lp; do
echo "== $cfg n=$n kernel=$k"
LKB_N=$n VOFE_LK=$k timeout -k 10 100 python tools/lk_bench.py $cfg 2>&1 | grep -E "exact|lk fwd" || exit 1
done; done; done
