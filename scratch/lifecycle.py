"""Context / pyramid lifecycle: repeated create, use, destroy must not leak device memory or pinned buffers."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import visual_odometry_amd as voa
from visual_odometry_amd import synthetic
from visual_odometry_amd.frontend import FrontEnd
a, b = synthetic.sequence(376, 1241, 0, 2)
pts = None
free0 = None
marks = {}
for it in range(600):
    fe = FrontEnd(0)
    if pts is None:
        pts = fe.goodFeaturesToTrack(a, 500, 0.01, 7).reshape(-1, 2).T
    trk = voa.SequenceTracker(376, 1241, frontend=fe)
    trk.reset(a)
    res, corners = trk.track_and_detect(b, (pts,), 300, 0.01, 10)
    fe.track_keypoints(pts, a, b, winSize=(31, 31), maxLevel=2)
    fe.goodFeaturesToTrack(b, 0, 0.05, 3)
    trk.close(); fe.close()
    torch.cuda.synchronize()
    if it in (10, 100, 200, 300, 400, 500): marks[it] = torch.cuda.mem_get_info()[0]
marks[600] = torch.cuda.mem_get_info()[0]
print("free device memory (MiB) after n lifecycles:", {k: round(v / 2**20, 1) for k, v in marks.items()},
      "torch reserved %.1f MiB" % (torch.cuda.memory_reserved() / 2**20))
assert marks[200] - marks[600] < 96 * 2**20, "device memory keeps shrinking"
print("ok")
