import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
fe = FrontEnd(0); L = fe.L
for cfg in ("kitti", "stress"):
    H, W, seed, n, win, lvl = synthetic.CONFIGS[cfg]
    a, b = synthetic.sequence(H, W, seed, 2)
    d = torch.from_numpy(np.stack([a, b])).cuda()
    c = fe.goodFeaturesToTrack(d[0], n, 0.01, 7 if cfg != "stress" else 10)
    pts = synthetic.pad_points(c, n, H, W, seed)
    dp = torch.from_numpy(pts).cuda()
    dn = torch.empty_like(dp); ds = torch.empty(n, dtype=torch.uint8, device="cuda"); de = torch.empty(n, device="cuda")
    pyr = Pyramid(fe, 2, H, W, win, lvl, True)
    for _ in range(5):
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2, d.data_ptr(), W, H * W))
        _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, 1, dp.data_ptr(), None, dn.data_ptr(), ds.data_ptr(), de.data_ptr(), n, 3, 30, 0.01, 1e-3, 1.0))
        fe.goodFeaturesToTrack(d[1], n, 0.01, 7 if cfg != "stress" else 10)
    torch.cuda.synchronize()
    print(cfg, "done", flush=True)
    pyr.free()
fe.close()
