"""Streaming kernels on large launches (roofline fractions are quoted on the stress frame and on
batched launches): builds pyramids of 128 KITTI frames and of 2 4K frames with the per-level kernels and
the score map of the same inputs.  Run under `rocprofv3 --kernel-trace`; bench.py's roofline_streaming block reports the same
kernels in GB/s with the algorithmic bytes of SURVEY 8d."""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
os.environ["VOFE_PYRAMID"] = "levels"
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
fe = FrontEnd(0); L = fe.L
for name, B in (("kitti", 128), ("stress", 2)):
    H, W, seed, n, win, lvl = synthetic.CONFIGS[name]
    a, b = synthetic.sequence(H, W, seed, 2)
    d = torch.from_numpy(np.stack([a, b] * (B // 2))).cuda()
    pyr = Pyramid(fe, B, H, W, win, lvl, True)
    score = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    for _ in range(5):
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, B, d.data_ptr(), W, H * W))
        _lib.check(L.vofe_corner_score(fe.ctx, d.data_ptr(), B, H, W, W, H * W, 3, 3, 0, 0.04, score.data_ptr()))
        torch.cuda.synchronize()
    pyr.free()
fe.close()
