import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import oracle
from visual_odometry_amd import synthetic
from visual_odometry_amd.frontend import FrontEnd
fe = FrontEnd(0)
for cfg, maxc, md in (("kitti", 1000, 10), ("kitti", 2000, 7), ("malaga", 3000, 10), ("stress", 20000, 10)):
    H, W, seed, n, win, lvl = synthetic.CONFIGS[cfg]
    img = synthetic.sequence(H, W, seed, 1)[0]
    d = torch.from_numpy(img).cuda()
    for _ in range(3): c = fe.goodFeaturesToTrack(d, maxc, 0.01, md)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 20
    for _ in range(K): c = fe.goodFeaturesToTrack(d, maxc, 0.01, md)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    t1 = time.perf_counter(); r = oracle.goodFeaturesToTrack(img, maxc, 0.01, md, nthreads=16); tc = time.perf_counter() - t1
    print("%-7s maxCorners %5d minDist %2d: GPU %.1f us/call (device-resident image, incl. D2H of corners), oracle 16 thr %.1f ms, n=%d equal=%s" % (cfg, maxc, md, dt * 1e6, tc * 1e3, len(c), np.array_equal(c, r)))
fe.close()
