"""LK launch time vs point count for the one- and two-keypoint-per-wavefront kernels (where should LK_X2_MIN_POINTS sit?)."""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
fe = FrontEnd(0); L = fe.L
for B in (1, 2, 3, 4, 5, 6, 8):
    imgs, pts = [], []
    for s in range(min(B, 4)):
        a, b = synthetic.sequence(H, W, seed + s, 2)
        imgs.append((a, b)); pts.append(synthetic.pad_points(fe.goodFeaturesToTrack(a, n, 0.01, 7), n, H, W, seed + s))
    d = torch.from_numpy(np.concatenate([np.stack([imgs[s % len(imgs)][0] for s in range(B)]), np.stack([imgs[s % len(imgs)][1] for s in range(B)])])).cuda()
    dp = torch.from_numpy(np.concatenate([pts[s % len(pts)] for s in range(B)])).cuda()
    idx = torch.arange(B, dtype=torch.int32).repeat_interleave(n).cuda()
    N = n * B
    dn = torch.empty_like(dp); ds = torch.empty(N, dtype=torch.uint8, device="cuda"); de = torch.empty(N, device="cuda")
    pyr = Pyramid(fe, 2 * B, H, W, win, lvl, True)
    _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2 * B, d.data_ptr(), W, H * W))
    res = {}
    for kern in ("x1", "x2"):
        os.environ["VOFE_LK"] = kern
        def lk(): _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, dp.data_ptr(), idx.data_ptr(), dn.data_ptr(), ds.data_ptr(), de.data_ptr(), N, 3, 30, 0.01, 1e-3, 1.0))
        for _ in range(5): lk()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for x, y in ev:
            x.record(); lk(); y.record()
        torch.cuda.synchronize()
        res[kern] = sorted(x.elapsed_time(y) for x, y in ev)[20] * 1e3
    print("N = %5d: x1 %.1f us, x2 %.1f us" % (N, res["x1"], res["x2"]), flush=True)
    pyr.free()
fe.close()
