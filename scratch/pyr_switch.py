"""Fused vs per-level pyramid build over build sizes (where should the size switch sit?)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
fe = FrontEnd(0); L = fe.L
for name, counts in (("kitti", (1, 2, 4, 5, 6, 8, 16, 32)), ("malaga", (1, 2, 4, 5, 6)), ("stress", (1, 2))):
    H, W, seed, n, win, lvl = synthetic.CONFIGS[name]
    a, b = synthetic.sequence(H, W, seed, 2)
    for B in counts:
        d = torch.from_numpy(np.stack([a, b] * ((B + 1) // 2))[:B]).cuda()
        pyr = Pyramid(fe, B, H, W, win, lvl, True)
        res = {}
        for path in ("fused", "levels"):
            if path == "fused" and lvl + 1 > 5: continue
            os.environ["VOFE_PYRAMID"] = path
            for _ in range(5): _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, B, d.data_ptr(), W, H * W))
            torch.cuda.synchronize(); K = 200; t0 = time.perf_counter()
            for _ in range(K): _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, B, d.data_ptr(), W, H * W))
            torch.cuda.synchronize(); res[path] = (time.perf_counter() - t0) / K * 1e6
        print("%-7s B=%2d (%.2f Mpx): %s" % (name, B, B * H * W / 1e6, "  ".join("%s %.1f us" % kv for kv in res.items())), flush=True)
        pyr.free()
fe.close()
