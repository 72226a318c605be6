"""goodFeaturesToTrack on one synthetic config (argv: config maxCorners minDistance), device-resident image."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic
from visual_odometry_amd.frontend import FrontEnd
cfg, maxc, md = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
fe = FrontEnd(0)
H, W, seed, n, win, lvl = synthetic.CONFIGS[cfg]
fr = synthetic.sequence(H, W, seed, 3)
d = torch.from_numpy(np.stack(fr)).cuda()
for t in range(3):
    for _ in range(5): c = fe.goodFeaturesToTrack(d[t], maxc, 0.01, md)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): c = fe.goodFeaturesToTrack(d[t], maxc, 0.01, md)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("%s frame %d maxCorners %d minDistance %g: %.1f us/call, %d corners" % (cfg, t, maxc, md, dt * 1e6, 0 if c is None else len(c)), flush=True)
import os
os.environ["VOFE_GFTT_DEBUG"] = "1"
fe.goodFeaturesToTrack(d[1], maxc, 0.01, md)
fe.close()
