"""Pyramid builds only (128 KITTI frames per build, per-level path), for `rocprofv3 --pmc` / `--kernel-trace` passes over the
pyramid chain:  PMC_WORKLOAD=tools/pyr_workload.py tools/pmc_passes.sh <tag> inst wait lds"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic, _lib                      # noqa: E402
from visual_odometry_amd.frontend import FrontEnd, Pyramid           # noqa: E402

H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
N = int(os.environ.get("PYR_FRAMES", 128))
fe = FrontEnd(0)
frames = synthetic.sequence(H, W, seed, 8)
d_frames = torch.from_numpy(np.stack(frames)).cuda()
imgs = d_frames[[k % 8 for k in range(N)]].contiguous()
pyr = Pyramid(fe, N, H, W, win, lvl, True)
for _ in range(int(os.environ.get("PMC_REPS", 5))):
    _lib.check(fe.L.vofe_pyramid_build(fe.ctx, pyr._h, 0, N, imgs.data_ptr(), W, H * W))
    torch.cuda.synchronize()
pyr.free()
fe.close()
