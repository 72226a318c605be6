"""Level-0 copy alone (maxLevel 0, no derivative planes) on 128 frames of width 1241 / 1240 / 1248: does the
alignment of the source rows matter?"""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
os.environ["VOFE_PYRAMID"] = "levels"
from visual_odometry_amd import _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
fe = FrontEnd(0); L = fe.L
B, H = 128, 376
for W in (1241, 1240, 1248, 1280):
    d = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device="cuda")
    pyr = Pyramid(fe, B, H, W, (21, 21), 0, False)
    ts = []
    for _ in range(12):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, B, d.data_ptr(), W, H * W))
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print("W=%d: min %.1f us  %.2f TB/s (2 B/px)" % (W, min(ts), 2.0 * B * H * W / min(ts) / 1e6), flush=True)
    pyr.free()
fe.close()
