"""Score map and detector call by blockSize / gradientSize (cv2's optional arguments): block 3 / aperture 3 runs the dedicated
register-strip kernels, every other pair corner_score_generic (run-time geometry, LDS tile).  One KITTI frame and 128 frames per
launch, torch CUDA events around 20 calls.  usage: python3 tools/score_geom_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic, _lib                         # noqa: E402
from visual_odometry_amd.frontend import FrontEnd                       # noqa: E402

fe = FrontEnd(0)
H, W, seed, _, _, _ = synthetic.CONFIGS["kitti"]
fr = synthetic.sequence(H, W, seed, 8)
d = torch.from_numpy(np.stack(fr)).cuda()
L = fe.L


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    fe.synchronize()
    with torch.cuda.stream(fe.stream):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(fe.stream)
        for _ in range(reps):
            fn()
        b.record(fe.stream)
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for frames in (1, 128):
    imgs = d[[k % 8 for k in range(frames)]].contiguous()
    score = torch.empty((frames, H, W), dtype=torch.float32, device="cuda")
    for bs, ks in ((3, 3), (5, 3), (7, 3), (3, 5), (7, 7), (15, 5), (31, 7)):
        us = timed(lambda: _lib.check(L.vofe_corner_score(fe.ctx, imgs.data_ptr(), frames, H, W, W, H * W, bs, ks, 0, 0.04, score.data_ptr())))
        print("%4d frames  blockSize %2d gradientSize %d  score map %9.1f us  (%.2f TB/s on 5 B/px)" % (frames, bs, ks, us, frames * H * W * 5 / us / 1e6))
    del imgs, score
import time
for bs, ks in ((3, 3), (7, 3), (7, 7)):
    t0 = time.perf_counter()
    for _ in range(50):
        fe.goodFeaturesToTrack(d[0], 1000, 0.01, 10, blockSize=bs, gradientSize=ks)
    print("goodFeaturesToTrack(1000, 0.01, 10, blockSize=%d, gradientSize=%d): %.1f us per call (device image in, corners to the host)" % (bs, ks, (time.perf_counter() - t0) / 50 * 1e6))
fe.close()
