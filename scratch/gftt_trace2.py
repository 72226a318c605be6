import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic
from visual_odometry_amd.frontend import FrontEnd
fe = FrontEnd(0)
H, W, seed, n, win, lvl = synthetic.CONFIGS["stress"]
img = synthetic.sequence(H, W, seed, 1)[0]
d = torch.from_numpy(img).cuda()
c = fe.goodFeaturesToTrack(d, 0, 0.01, 0)
print("stress candidates:", 0 if c is None else len(c))
c = fe.goodFeaturesToTrack(d, 0, 0.01, 10)
print("stress accepted (md 10):", 0 if c is None else len(c))
for _ in range(3): c = fe.goodFeaturesToTrack(d, 20000, 0.01, 10)
torch.cuda.synchronize()
fe.close()
