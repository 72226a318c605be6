import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic
from visual_odometry_amd.frontend import FrontEnd
fe = FrontEnd(0)
for cfg, maxc, md in (("kitti", 1000, 10), ("malaga", 3000, 10), ("stress", 20000, 10)):
    H, W, seed, n, win, lvl = synthetic.CONFIGS[cfg]
    img = synthetic.sequence(H, W, seed, 1)[0]
    d = torch.from_numpy(img).cuda()
    for _ in range(3): c = fe.goodFeaturesToTrack(d, maxc, 0.01, md)
    torch.cuda.synchronize()
fe.close()
