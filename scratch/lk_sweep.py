import sys, numpy as np, torch
sys.path.insert(0, '.')
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
fe = FrontEnd(0); L = fe.L
a, b = synthetic.sequence(H, W, seed, 2)
d = torch.from_numpy(np.stack([a, b])).cuda()
c = fe.goodFeaturesToTrack(d[0], n, 0.01, 7)
pts = synthetic.pad_points(c, n, H, W, seed)
dp = torch.from_numpy(pts).cuda()
dn = torch.empty_like(dp); ds = torch.empty(n, dtype=torch.uint8, device="cuda"); de = torch.empty(n, device="cuda")
cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
_lib.check(L.vofe_lk_set_counters(fe.ctx, cnt.data_ptr()))
for lv in (3, 0):
    pyr = Pyramid(fe, 2, H, W, win, lv, True)
    _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2, d.data_ptr(), W, H * W))
    for npts in (2000, 500, 64):
        for mc in (1, 2, 5, 10, 30):
            def lk(): _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, 1, dp.data_ptr(), None, dn.data_ptr(), ds.data_ptr(), de.data_ptr(), npts, 1 if mc < 30 else 3, mc, 0.01, 1e-3, 1.0))
            for _ in range(5): lk()
            cnt.zero_()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
            for x, y in ev:
                x.record(); lk(); y.record()
            torch.cuda.synchronize()
            ts = sorted(x.elapsed_time(y) for x, y in ev)
            k = cnt.cpu().numpy() / 50.0
            print("levels %d pts %4d maxcount %2d: median %.1f us; per launch: point-levels %.0f iterations %.0f" % (lv + 1, npts, mc, ts[25] * 1e3, k[1], k[2]))
fe.close()
