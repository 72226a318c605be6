"""The score-map kernel alone: HIP events around `reps` launches, 128 KITTI frames per launch and one 4K frame, bit
parity of one frame against the oracle, for the kernel variant the VOFE_SCORE_* switches select.
usage: [VOFE_SCORE_NC=2] [VOFE_SCORE_PF=1] [VOFE_SCORE_RS=8] python3 tools/score_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                                           # noqa: E402
from visual_odometry_amd import synthetic, _lib                         # noqa: E402
from visual_odometry_amd.frontend import FrontEnd                       # noqa: E402

fe = FrontEnd(0)
L = fe.L
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("VOFE_SCORE")) or "default"
for name, H, W, seed, frames in (("kitti x128", 376, 1241, 0, 128), ("4K x1", 2160, 3840, 2, 1), ("kitti x1", 376, 1241, 0, 1)):
    fr = synthetic.sequence(H, W, seed, 2)
    imgs = torch.from_numpy(np.stack([fr[k % 2] for k in range(frames)])).cuda()
    score = torch.empty((frames, H, W), dtype=torch.float32, device="cuda")

    def once():
        _lib.check(L.vofe_corner_score(fe.ctx, imgs.data_ptr(), frames, H, W, W, H * W, 3, 3, 0, 0.04, score.data_ptr()))
    once()
    torch.cuda.synchronize()
    ok = np.array_equal(score[0].cpu().numpy(), oracle.corner_score(fr[0], nthreads=16))
    ok = ok and np.array_equal(score[frames - 1].cpu().numpy(), oracle.corner_score(fr[(frames - 1) % 2], nthreads=16))
    for _ in range(5):
        once()
    reps = 30
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    s = torch.cuda.current_stream()
    for a, b in ev:
        a.record(s); once(); b.record(s)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    med = ts[len(ts) // 2] * 1e3
    gbs = frames * H * W * 5 / (med * 1e-6) / 1e9
    print("%-28s %-12s median %8.1f us  min %8.1f us  %7.1f GB/s = %4.1f %% of 8 TB/s  exact=%s" % (
        tag, name, med, ts[0] * 1e3, gbs, gbs / 80.0, ok), flush=True)
fe.close()
