"""A short, fixed sequence of the launches bench.py times, for `rocprofv3 --pmc` passes (one counter set per run):

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o x -- python3 tools/pmc_workload.py
    python3 tools/pmc_sum.py gpurun_out/pmc_fetch ""          # per kernel: counters per dispatch

  * 3 launch sets of the bench shape: pyramid build of 128 KITTI frames (per-level kernels) + ONE tracker launch of
    64 pairs x 2000 points (lk_track_kernel_x2<2>)
  * 3 single-pair launches (fused pyramid + lk_track_kernel<7,21>, 2000 points)
  * 3 score maps of 128 frames (corner_score_strip<16>) and 3 detector calls on one frame
  * with PMC_BIG=1 also the streaming kernels on 2048 frames per launch (1 GB of input: beyond the Infinity Cache)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic, _lib                      # noqa: E402
from visual_odometry_amd.frontend import FrontEnd, Pyramid           # noqa: E402

H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
B = 64
REPS = int(os.environ.get("PMC_REPS", 3))
fe = FrontEnd(0)
L = fe.L
frames = synthetic.sequence(H, W, seed, 8)
d_frames = torch.from_numpy(np.stack(frames)).cuda()
pts = [synthetic.pad_points(fe.goodFeaturesToTrack(d_frames[t], n, 0.01, 7), n, H, W, seed + t) for t in range(7)]
pairs = [k % 7 for k in range(B)]
imgs = torch.cat([d_frames[pairs], d_frames[[q + 1 for q in pairs]]]).contiguous()
d_p = torch.from_numpy(np.concatenate([pts[q] for q in pairs])).cuda()
idx = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(n).contiguous()
NB = n * B
nxt = torch.empty((NB, 2), dtype=torch.float32, device="cuda")
st = torch.empty(NB, dtype=torch.uint8, device="cuda")
err = torch.empty(NB, dtype=torch.float32, device="cuda")
pyr = Pyramid(fe, 2 * B, H, W, win, lvl, True)
score = torch.empty((2 * B, H, W), dtype=torch.float32, device="cuda")
d_one = torch.from_numpy(pts[0]).cuda()
for _ in range(REPS):
    _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2 * B, imgs.data_ptr(), W, H * W))
    _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, d_p.data_ptr(), idx.data_ptr(), B, nxt.data_ptr(), st.data_ptr(),
                               err.data_ptr(), NB, 3, 30, 0.01, 1e-3, 1.0))
    torch.cuda.synchronize()
for _ in range(REPS):
    _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2, d_frames.data_ptr(), W, H * W))
    _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, 1, d_one.data_ptr(), None, 1, nxt.data_ptr(), st.data_ptr(),
                               err.data_ptr(), n, 3, 30, 0.01, 1e-3, 1.0))
    torch.cuda.synchronize()
for _ in range(REPS):
    _lib.check(L.vofe_corner_score(fe.ctx, imgs.data_ptr(), 2 * B, H, W, W, H * W, 3, 3, 0, 0.04, score.data_ptr()))
    torch.cuda.synchronize()
for t in range(REPS):
    fe.goodFeaturesToTrack(d_frames[t], n, 0.01, 7)
pyr.free()
del score
if os.environ.get("PMC_BIG"):
    NBIG = 2048
    big = d_frames[[k % 8 for k in range(NBIG)]].contiguous()
    pyr = Pyramid(fe, NBIG, H, W, win, lvl, True)
    for _ in range(2):
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, NBIG, big.data_ptr(), W, H * W))
        torch.cuda.synchronize()
    pyr.free()
fe.close()
