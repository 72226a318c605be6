"""Where a wavefront of the single-keypoint LK kernel spends its time (diagnostic build only).

Build the library with -DLK_STAMPS into tools/diag/libvofe_stamps.so (see tools/lk_stamps.sh), then
    VOFE_LIB=tools/diag/libvofe_stamps.so python tools/lk_stamps.py [config] [n_points]
Prints the s_memtime deltas per phase, averaged over the first 256 wavefronts of one fwd+bwd launch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic, _lib                    # noqa: E402
from visual_odometry_amd.frontend import FrontEnd, Pyramid          # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "kitti"
H, W, seed, n, win, lvl = synthetic.CONFIGS[cfg]
if len(sys.argv) > 2:
    n = int(sys.argv[2])
fe = FrontEnd(0)
L = fe.L
a, b = synthetic.sequence(H, W, seed, 2)
d = torch.from_numpy(np.stack([a, b])).cuda()
c = fe.goodFeaturesToTrack(d[0], n, 0.01, 7)
pts = synthetic.pad_points(c, n, H, W, seed)
dp = torch.from_numpy(pts).cuda()
dn = torch.empty_like(dp)
ds = torch.empty(n, dtype=torch.uint8, device="cuda")
de = torch.empty(n, device="cuda")
pyr = Pyramid(fe, 2, H, W, win, lvl, True)


def build():
    _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2, d.data_ptr(), W, H * W))


def lk():
    _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, 1, dp.data_ptr(), None, 1, dn.data_ptr(), ds.data_ptr(), de.data_ptr(),
                               n, 3, 30, 0.01, 1e-3, 1.0))


for _ in range(5):
    build(); lk()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
for x, y in ev:
    build(); x.record(); lk(); y.record()
torch.cuda.synchronize()
ts = sorted(x.elapsed_time(y) for x, y in ev)
print("%s, %d points: launch median %.1f us (without stamps being stored)" % (cfg, n, ts[25] * 1e3))
NS = 12
cnt = torch.zeros(8 + NS * 256, dtype=torch.int64, device="cuda")
_lib.check(L.vofe_lk_set_counters(fe.ctx, cnt.data_ptr()))
build()
x, y = ev[0]
x.record(); lk(); y.record()
torch.cuda.synchronize()
_lib.check(L.vofe_lk_set_counters(fe.ctx, None))
print("stamped launch %.1f us" % (x.elapsed_time(y) * 1e3))
c = cnt.cpu().numpy()
print("level passes per point %.2f, iterations per level pass %.2f" % (c[1] / n, c[2] / max(c[1], 1)))
s = c[8:].reshape(256, NS).astype(np.float64)
if not s[:, 7].any():
    sys.exit("no stamps: the library was not built with -DLK_STAMPS")
names = ["0 template loads + structure tensor", "1 min-eig / determinant", "2 search region -> LDS", "3 pixel loop",
         "4 window sums", "5 2x2 solve", "6 convergence tests + iteration head", "7 whole wavefront", "8 level prologue (descriptors, bounds)",
         "9 err + end-of-level barrier", "10 kernel prologue (point, image index)", "11 -"]
tot = s[:, 7].mean()
acc = 0.0
for k, nm in enumerate(names):
    m = s[:, k].mean()
    if k != 7:
        acc += m
    print("  %-42s mean %9.0f ticks  %5.1f %%   (min %7.0f  max %8.0f)" % (nm, m, 100 * m / tot, s[:, k].min(), s[:, k].max()))
print("  %-42s mean %9.0f ticks  %5.1f %%" % ("other (last convergence test of a level, FB check, stores)", tot - acc, 100 * (tot - acc) / tot))
print("  slowest / fastest wavefront: %.0f / %.0f ticks" % (s[:, 7].max(), s[:, 7].min()))
fe.close()
