"""Times the fused fwd+bwd LK launch alone (HIP events) and checks bit-exactness vs the oracle."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import oracle
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
cfg = sys.argv[1] if len(sys.argv) > 1 else "kitti"
import os
H, W, seed, n, win, lvl = synthetic.CONFIGS[cfg]
if os.environ.get("LKB_WIN"): win = (int(os.environ["LKB_WIN"]),) * 2       # overrides for what-if runs
if os.environ.get("LKB_N"): n = int(os.environ["LKB_N"])
fe = FrontEnd(0); L = fe.L
a, b = synthetic.sequence(H, W, seed, 2)
d = torch.from_numpy(np.stack([a, b])).cuda()
c = fe.goodFeaturesToTrack(d[0], n, 0.01, 7 if cfg != "stress" else 10)
pts = synthetic.pad_points(c, n, H, W, seed)
dp = torch.from_numpy(pts).cuda()
dn = torch.empty_like(dp); ds = torch.empty(n, dtype=torch.uint8, device="cuda"); de = torch.empty(n, device="cuda")
pyr = Pyramid(fe, 2, H, W, win, lvl, True)
def build(): _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2, d.data_ptr(), W, H * W))
def lk(): _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, 1, dp.data_ptr(), None, 1, dn.data_ptr(), ds.data_ptr(), de.data_ptr(), n, 3, 30, 0.01, 1e-3, 1.0))
build(); lk(); torch.cuda.synchronize()
if cfg != "stress" or "--check" in sys.argv:
    t0, s0, e0 = oracle.track_keypoints(pts.T, a, b, win, lvl, (3, 30, 0.01), 1e-3, 1.0, 16)
    print(cfg, "exact:", np.array_equal(dn.cpu().numpy().T, t0), np.array_equal(ds.cpu().numpy(), s0), np.array_equal(de.cpu().numpy(), e0), "tracked", int(s0.sum()))
for name, fn in (("pyramid x2", build), ("lk fwd+bwd", lk)):
    for _ in range(10): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for x, y in ev:
        x.record(); fn(); y.record()
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in ev)
    print("%s: median %.1f us  min %.1f us" % (name, ts[50] * 1e3, ts[0] * 1e3))
cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
_lib.check(L.vofe_lk_set_counters(fe.ctx, cnt.data_ptr())); lk(); torch.cuda.synchronize()
_lib.check(L.vofe_lk_set_counters(fe.ctx, None))
c = cnt.cpu().numpy()
print("points %d: %.2f level passes per point, %.2f Gauss-Newton iterations per level pass" % (n, c[1] / n, c[2] / max(c[1], 1)))
fe.close()
