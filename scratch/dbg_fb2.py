import sys, numpy as np
sys.path.insert(0, '.')
import oracle
from visual_odometry_amd import synthetic
from visual_odometry_amd.frontend import FrontEnd
fe = FrontEnd(0)
h, w = 376, 1241
a, b = synthetic.sequence(h, w, 0, 2)
pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 2000, 0.01, 7), 2000, h, w, 0)
f0, fs0, _ = oracle.calcOpticalFlowPyrLK(a, b, pts, None, (21, 21), 3, (3, 30, 0.01), 0, 1e-3)
sel = f0[[866, 1671, 1753, 5]].reshape(-1, 2).copy()
for lvl in (0, 1, 2, 3):
    for cnt in (1, 2, 3, 4, 5, 6, 8, 10, 15, 30):
        g, gs, ge = fe.calcOpticalFlowPyrLK(b, a, sel, None, (21, 21), lvl, (1, cnt, 0.01), 0, 1e-3)
        c, cs, ce = oracle.calcOpticalFlowPyrLK(b, a, sel, None, (21, 21), lvl, (1, cnt, 0.01), 0, 1e-3)
        if not np.array_equal(g, c):
            print("lvl", lvl, "cnt", cnt, "gpu", g.reshape(-1, 2)[:3].tolist(), "cpu", c.reshape(-1, 2)[:3].tolist())
            break
    else:
        print("lvl", lvl, "all equal")
fe.close()
