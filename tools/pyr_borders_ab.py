"""Fused pyramid build of one small frame: border bands written inside the launch (default) against the second launch
(VOFE_PYR_BORDERS=separate).  Event-bracketed kernels, host-to-host build time, and the single-pair tracker call.
usage: python3 tools/pyr_borders_ab.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic, _lib                         # noqa: E402
from visual_odometry_amd.frontend import FrontEnd, Pyramid              # noqa: E402

for mode in ("own", "separate", "own", "separate"):
    if mode == "separate":
        os.environ["VOFE_PYR_BORDERS"] = "separate"
    else:
        os.environ.pop("VOFE_PYR_BORDERS", None)
    fe = FrontEnd(0)
    L = fe.L
    out = []
    for name in ("kitti", "parking", "malaga"):
        H, W, seed, n, win, lvl = synthetic.CONFIGS[name]
        a, b = synthetic.sequence(H, W, seed, 2)
        d = torch.from_numpy(np.stack([a, b])).cuda()
        pyr = Pyramid(fe, 2, H, W, win, lvl, True)
        for _ in range(20):
            pyr.build(d[1:2], 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            pyr.build(d[1:2], 1)
            fe.synchronize()
        t_build = (time.perf_counter() - t0) / 300 * 1e6
        c = fe.goodFeaturesToTrack(a, n, 0.01, 7)
        pts = synthetic.pad_points(c, n, H, W, seed)
        for _ in range(5):
            fe.track_keypoints(pts.T, d[0], d[1], winSize=win, maxLevel=lvl)
        t0 = time.perf_counter()
        for _ in range(200):
            fe.track_keypoints(pts.T, d[0], d[1], winSize=win, maxLevel=lvl)
        t_pair = (time.perf_counter() - t0) / 200 * 1e6
        out.append("%s build %.1f us, track_keypoints (device frames) %.1f us" % (name, t_build, t_pair))
        pyr.free()
    print("bands %-8s  %s" % (mode, "; ".join(out)), flush=True)
    fe.close()
