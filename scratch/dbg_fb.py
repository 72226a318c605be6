import sys, numpy as np
sys.path.insert(0, '.')
import oracle
from visual_odometry_amd import synthetic
from visual_odometry_amd.frontend import FrontEnd
fe = FrontEnd(0)
h, w = 376, 1241
a, b = synthetic.sequence(h, w, 0, 2)
pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 2000, 0.01, 7), 2000, h, w, 0)
t1, s1, e1 = fe.track_keypoints(pts.T, a, b, (21, 21), 3, (3, 30, 0.01), 1e-3)
t0, s0, e0 = oracle.track_keypoints(pts.T, a, b, (21, 21), 3, (3, 30, 0.01), 1e-3, nthreads=8)
bad = np.nonzero(s0 != s1)[0]
print("mismatch", bad, "err eq", np.array_equal(e0, e1))
f1, fs1, fe1 = fe.calcOpticalFlowPyrLK(a, b, pts, None, (21, 21), 3, (3, 30, 0.01), 0, 1e-3)
bk1, bs1, _ = fe.calcOpticalFlowPyrLK(b, a, f1, None, (21, 21), 3, (3, 30, 0.01), 0, 1e-3)
f0, fs0, _ = oracle.calcOpticalFlowPyrLK(a, b, pts, None, (21, 21), 3, (3, 30, 0.01), 0, 1e-3)
bk0, bs0, _ = oracle.calcOpticalFlowPyrLK(b, a, f0, None, (21, 21), 3, (3, 30, 0.01), 0, 1e-3)
print("fwd eq", np.array_equal(f1, f0), np.array_equal(fs1, fs0), "bwd eq", np.array_equal(bk1, bk0, equal_nan=True))
for i in bad[:10]:
    print(i, pts[i], "fwd", f1[i, 0], fs1[i, 0], fs0[i,0], "bk gpu", bk1[i, 0], "bk cpu", bk0[i, 0], "s gpu/cpu", s1[i], s0[i],
          "d", np.abs(pts[i] - bk0[i, 0]).max())
