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
cnt = torch.zeros(8 + 8 * 256, dtype=torch.int64, device="cuda")
pyr = Pyramid(fe, 2, H, W, win, lvl, True)
_lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2, d.data_ptr(), W, H * W))
for npts in (2000, 64):
    for _ in range(3):
        _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, 1, dp.data_ptr(), None, dn.data_ptr(), ds.data_ptr(), de.data_ptr(), npts, 3, 30, 0.01, 1e-3, 1.0))
    _lib.check(L.vofe_lk_set_counters(fe.ctx, cnt.data_ptr())); cnt.zero_()
    _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, 1, dp.data_ptr(), None, dn.data_ptr(), ds.data_ptr(), de.data_ptr(), npts, 3, 30, 0.01, 1e-3, 1.0))
    torch.cuda.synchronize(); _lib.check(L.vofe_lk_set_counters(fe.ctx, None))
    k = cnt.cpu().numpy()
    nb = min(npts, 256)
    st = k[8:8 + 8 * nb].reshape(nb, 8).astype(np.float64)
    names = ["template loads+compute+reduce", "minEig/D float", "stage write+sync", "iter: lds+px compute", "iter: reduce", "iter: solve", "iter: tail(check)+head", "-"]
    print("points", npts, "levels/pt %.1f iters/pt %.1f" % (k[1] / k[0], k[2] / k[0]), " total cycles/pt (100MHz ticks?) %.0f" % st.sum(1).mean())
    for i, nm in enumerate(names[:7]):
        print("   %-32s mean %8.0f  share %5.1f%%" % (nm, st[:, i].mean(), 100 * st[:, i].sum() / st.sum()))
fe.close()
