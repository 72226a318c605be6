import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic
from visual_odometry_amd.frontend import FrontEnd
fe = FrontEnd(0)
H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
img = synthetic.sequence(H, W, seed, 1)[0]
d = torch.from_numpy(img).cuda()
for _ in range(6): s = fe.corner_score(d)          # no header: no atomicMax
for _ in range(6): c = fe.goodFeaturesToTrack(d, 2000, 0.01, 7)
torch.cuda.synchronize()
fe.close()
