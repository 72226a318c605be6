"""Do the streaming kernels run UNDER a tracker launch, or beside it?  The two-keypoint tracker kernel holds 3 waves of
160 VGPRs per SIMD, which leaves 32 of the 512 registers free: a kernel whose waves need no more than that could share the
SIMDs with it.  Experiment: one 64-pair tracker launch on stream T while stream P runs N level-0 copies (10 VGPRs) or N
full pyramid builds (Scharr 36-40, pyrDown 66 VGPRs); wall time of each alone and of both together.
usage: python3 tools/coresidency.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic, _lib                         # noqa: E402
from visual_odometry_amd.frontend import FrontEnd, Pyramid              # noqa: E402

H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
B = 64
L = _lib.load()
dev = torch.device("cuda", 0)
fr = synthetic.sequence(H, W, seed, 2)
PRIO = int(os.environ.get("COR_PRIO", "0"))      # -1: the side stream gets the high priority
sT, sP = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=PRIO)
print("side stream priority", PRIO, "range", torch.cuda.Stream.priority_range())
feT, feP = FrontEnd(0, stream=sT), FrontEnd(0, stream=sP)
c = feT.goodFeaturesToTrack(fr[0], n, 0.01, 7)
pts = synthetic.pad_points(c, n, H, W, seed)
imgs = torch.from_numpy(np.stack([fr[0]] * B + [fr[1]] * B)).to(dev)
d_pts = torch.from_numpy(np.tile(pts, (B, 1))).to(dev)
d_idx = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(n).contiguous()
NB = n * B
with torch.cuda.stream(sT):
    pyrT = Pyramid(feT, 2 * B, H, W, win, lvl, True)
    pyrT.build(imgs, 0)
    nxt = torch.empty((NB, 2), dtype=torch.float32, device=dev)
    st = torch.empty(NB, dtype=torch.uint8, device=dev)
    err = torch.empty(NB, dtype=torch.float32, device=dev)
with torch.cuda.stream(sP):
    pyr_copy = Pyramid(feP, 2 * B, H, W, win, 0, False)        # level 0 only, no derivative planes: the build is the copy kernel
    pyr_full = Pyramid(feP, 2 * B, H, W, win, lvl, True)
torch.cuda.synchronize()


def lk():
    _lib.check(L.vofe_track_fb(feT.ctx, pyrT._h, 0, pyrT._h, B, d_pts.data_ptr(), d_idx.data_ptr(), B, nxt.data_ptr(), st.data_ptr(),
                               err.data_ptr(), NB, 3, 30, 0.01, 1e-3, 1.0))


def side(pyr, reps):
    for _ in range(reps):
        _lib.check(L.vofe_pyramid_build(feP.ctx, pyr._h, 0, 2 * B, imgs.data_ptr(), W, H * W))


def wall(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6


lk(); side(pyr_copy, 2); side(pyr_full, 2)
t_lk = wall(lk)
for name, pyr, reps in (("level-0 copy (10 VGPRs)", pyr_copy, 30), ("full pyramid build (copy 10, pyrDown 66, Scharr 36-40 VGPRs)", pyr_full, 6)):
    t_side = wall(lambda: side(pyr, reps))

    def both():
        lk()
        side(pyr, reps)
    t_both = wall(both)
    print("%s x %d: tracker launch alone %.0f us, side work alone %.0f us, together %.0f us (sum %.0f): %.0f %% of the side work ran under the launch" % (
        name, reps, t_lk, t_side, t_both, t_lk + t_side, 100.0 * (t_lk + t_side - t_both) / t_side))
