import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
fe = FrontEnd(0); L = fe.L
H, W, seed, n, win, lvl = synthetic.CONFIGS["stress"]
a, b = synthetic.sequence(H, W, seed, 2)
d = torch.from_numpy(np.stack([a, b])).cuda()
pyr = Pyramid(fe, 2, H, W, win, lvl, True)
for _ in range(4):
    _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2, d.data_ptr(), W, H * W))
torch.cuda.synchronize()
pyr.free(); fe.close()
