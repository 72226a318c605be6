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
for (x, y, p, name) in ((a, b, pts, "fwd"), (b, a, f0.reshape(-1, 2).copy(), "bwd")):
    g, gs, ge = fe.calcOpticalFlowPyrLK(x, y, p, None, (21, 21), 3, (3, 30, 0.01), 0, 1e-3)
    c, cs, ce = oracle.calcOpticalFlowPyrLK(x, y, p, None, (21, 21), 3, (3, 30, 0.01), 0, 1e-3, nthreads=8)
    bad = np.nonzero(np.any(g.reshape(-1, 2) != c.reshape(-1, 2), axis=1) & np.isfinite(c.reshape(-1,2)).all(1))[0]
    print(name, "n mismatching points:", len(bad), bad[:10])
fe.close()
