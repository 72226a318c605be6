"""Where the launch-set period of the headline benchmark goes: HIP events around every tracker launch of a short run of
bench.py's two-stream schedule (64 KITTI pairs per launch set), then per set: tracker launch duration, gap between the end
of one tracker launch and the start of the next (they are chained across the two streams by an event), and the period.
usage: python3 tools/overlap_gaps.py [sets] [streams]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic, _lib                         # noqa: E402
from visual_odometry_amd.frontend import FrontEnd, Pyramid              # noqa: E402

SETS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
MODE = sys.argv[3] if len(sys.argv) > 3 else "chain"                    # chain | free
H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
B = 64
L = _lib.load()
dev = torch.device("cuda", 0)
fr = synthetic.sequence(H, W, seed, 3)
fe0 = FrontEnd(0)
c = fe0.goodFeaturesToTrack(fr[0], n, 0.01, 7)
pts = synthetic.pad_points(c, n, H, W, seed)
imgs = torch.from_numpy(np.stack([fr[0]] * B + [fr[1]] * B)).to(dev)
d_pts = torch.from_numpy(np.tile(pts, (B, 1))).to(dev)
d_idx = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(n).contiguous()
NB = n * B


class Worker:
    def __init__(self):
        self.stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(self.stream):
            self.fe = FrontEnd(0)
            self.pyr = Pyramid(self.fe, 2 * B, H, W, win, lvl, True)
            self.nxt = torch.empty((NB, 2), dtype=torch.float32, device=dev)
            self.st = torch.empty(NB, dtype=torch.uint8, device=dev)
            self.err = torch.empty(NB, dtype=torch.float32, device=dev)


workers = [Worker() for _ in range(S)]
chain = [None]
ev = []


def launch(i, record):
    w = workers[i % S]
    _lib.check(L.vofe_pyramid_build(w.fe.ctx, w.pyr._h, 0, 2 * B, imgs.data_ptr(), W, H * W))
    p0 = torch.cuda.Event(enable_timing=True)
    if record:
        p0.record(w.stream)                      # pyramid kernels of this set are done
    if chain[0] is not None and MODE == "chain":
        w.stream.wait_event(chain[0])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if record:
        a.record(w.stream)
    _lib.check(L.vofe_track_fb(w.fe.ctx, w.pyr._h, 0, w.pyr._h, B, d_pts.data_ptr(), d_idx.data_ptr(), B, w.nxt.data_ptr(),
                               w.st.data_ptr(), w.err.data_ptr(), NB, 3, 30, 0.01, 1e-3, 1.0))
    b.record(w.stream)
    chain[0] = b
    if record:
        ev.append((p0, a, b))


for i in range(2 * S):
    launch(i, False)
torch.cuda.synchronize()
for i in range(SETS):
    launch(i, True)
torch.cuda.synchronize()
lk = [a.elapsed_time(b) * 1e3 for _, a, b in ev]
gap = [ev[i][2].elapsed_time(ev[i + 1][1]) * 1e3 for i in range(len(ev) - 1)]
period = [ev[i][2].elapsed_time(ev[i + 1][2]) * 1e3 for i in range(len(ev) - 1)]
pyr_lead = [ev[i + 1][0].elapsed_time(ev[i][2]) * 1e3 for i in range(len(ev) - 1)]     # > 0: the next set's pyramids were ready before this launch ended
sl = slice(4, None)
print("streams %d, %s: tracker launch %.1f us (min %.1f), gap end->next start %.1f us (min %.1f, max %.1f), period %.1f us = %.0f frames/s, "
      "next pyramids ready %.0f us before the launch ends" % (S, MODE, np.mean(lk[sl]), np.min(lk[sl]), np.mean(gap[sl]), np.min(gap[sl]), np.max(gap[sl]),
                                                            np.mean(period[sl]), B / np.mean(period[sl]) * 1e6, np.mean(pyr_lead[sl])))
