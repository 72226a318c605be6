import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from scipy.spatial.distance import cdist
from visual_odometry_amd.frontend import FrontEnd
from visual_odometry_amd import _lib
fe = FrontEnd(0)
rng = np.random.default_rng(0)
for nf, ne in ((1000, 2000), (3000, 6000), (20000, 40000)):
    f = (rng.random((nf, 2)) * [1241, 376]).astype(np.float32); e = (rng.random((ne, 2)) * [1241, 376]).astype(np.float32)
    df, de = torch.from_numpy(f).cuda(), torch.from_numpy(e).cuda(); m = torch.empty(nf, dtype=torch.uint8, device="cuda")
    def run(): _lib.check(fe.L.vofe_new_feature_mask(fe.ctx, df.data_ptr(), nf, de.data_ptr(), ne, 15.0, m.data_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): run()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter(); fe.new_feature_mask(f, e, 15); th = time.perf_counter() - t0
    t0 = time.perf_counter(); ref = np.min(cdist(f, e), axis=1) > 15; tc = time.perf_counter() - t0
    print("%d x %d: GPU %.1f us device-resident, %.1f us host-to-host; scipy cdist+min %.2f ms; equal=%s" % (nf, ne, tg * 1e6, th * 1e6, tc * 1e3, np.array_equal(m.cpu().numpy().astype(bool), ref)))
fe.close()
