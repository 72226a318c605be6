"""How the streaming kernels' bandwidth grows with the launch size: per-level pyramid kernels and the score map on
B = 128, 512, 2048 KITTI frames (60 MB .. 1 GB of input), HIP events around each call pair (build, score) are
not fine enough, so run under `rocprofv3 --kernel-trace --output-format csv` and summarise with
scratch/stream_scale_sum.py <dir>."""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
os.environ["VOFE_PYRAMID"] = "levels"
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
fe = FrontEnd(0); L = fe.L
H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
a, b = synthetic.sequence(H, W, seed, 2)
for B in [int(v) for v in os.environ.get("SCALE_B", "128,512,2048").split(",")]:
    d = torch.from_numpy(np.stack([a, b])).cuda().repeat(B // 2, 1, 1).contiguous()
    pyr = Pyramid(fe, B, H, W, win, lvl, True)
    score = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    for _ in range(4):
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, B, d.data_ptr(), W, H * W))
        _lib.check(L.vofe_corner_score(fe.ctx, d.data_ptr(), B, H, W, W, H * W, 3, 3, 0, 0.04, score.data_ptr()))
        torch.cuda.synchronize()
    pyr.free(); del score, d
    torch.cuda.empty_cache()
fe.close()
