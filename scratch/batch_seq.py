"""B independent KITTI sequences batched into single launches (slots + img_idx)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
fe = FrontEnd(0); L = fe.L
for B in (1, 4, 16, 64):
    imgs = []; pts = []
    for s in range(min(B, 4)):
        a, b = synthetic.sequence(H, W, seed + s, 2)
        c = fe.goodFeaturesToTrack(a, n, 0.01, 7)
        imgs.append((a, b)); pts.append(synthetic.pad_points(c, n, H, W, seed + s))
    # slots: prev images 0..B-1, next images B..2B-1
    prev = np.stack([imgs[s % len(imgs)][0] for s in range(B)]); nxt = np.stack([imgs[s % len(imgs)][1] for s in range(B)])
    d = torch.from_numpy(np.concatenate([prev, nxt])).cuda()
    dp = torch.from_numpy(np.concatenate([pts[s % len(pts)] for s in range(B)])).cuda()
    idx = torch.arange(B, dtype=torch.int32).repeat_interleave(n).cuda()
    N = n * B
    dn = torch.empty_like(dp); ds = torch.empty(N, dtype=torch.uint8, device="cuda"); de = torch.empty(N, device="cuda")
    pyr = Pyramid(fe, 2 * B, H, W, win, lvl, True)
    def step():
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2 * B, d.data_ptr(), W, H * W))
        _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, dp.data_ptr(), idx.data_ptr(), dn.data_ptr(), ds.data_ptr(), de.data_ptr(), N, 3, 30, 0.01, 1e-3, 1.0))
    for _ in range(3): step()
    torch.cuda.synchronize(); K = max(4, 200 // B); t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record(); _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2 * B, d.data_ptr(), W, H * W)); ev[1].record()
    _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, dp.data_ptr(), idx.data_ptr(), dn.data_ptr(), ds.data_ptr(), de.data_ptr(), N, 3, 30, 0.01, 1e-3, 1.0)); ev[2].record()
    torch.cuda.synchronize()
    print("B=%3d: %.1f us/step -> %.0f frames/s (pyramids %.1f us, LK %.1f us), tracked %.3f" % (B, dt * 1e6, B / dt, ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3, float(ds.float().mean())))
    pyr.free()
fe.close()
