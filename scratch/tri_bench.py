"""vofe_triangulate_candidates / vofe_candidate_filter vs the reference's Python loops (numpy + LAPACK SVD)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle
from test_oracle_cpu import triangulation_scene
from visual_odometry_amd.frontend import FrontEnd
fe = FrontEnd(0)
for fdt in (np.float32, np.float64):
    K, pose, Ci, Fi, Ti, flags = triangulation_scene(1000, 9, fdt)
    t0 = time.perf_counter(); rk, rp = oracle.triangulate_candidates(K, pose, Ci, Fi, Ti, flags); t_ref = time.perf_counter() - t0
    for _ in range(3): k, p = fe.triangulate_candidates(K, pose, Ci, Fi, Ti, flags)
    t0 = time.perf_counter()
    for _ in range(20): k, p = fe.triangulate_candidates(K, pose, Ci, Fi, Ti, flags)
    t_gpu = (time.perf_counter() - t0) / 20
    both = k & rk
    a = p[:, np.cumsum(k)[both] - 1]; b = rp[:, np.cumsum(rk)[both] - 1]
    print("%s: loop %.1f ms, device path host-to-host %.3f ms; kept %d/%d (same decisions: %s); max rel diff of X %.2e" % (
        fdt.__name__, t_ref * 1e3, t_gpu * 1e3, k.sum(), flags.sum(), np.array_equal(k, rk), np.abs(a / b - 1).max()))
# kernel alone
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
dK, dP, dC, dF, dT, dI = d(K), d(pose), d(Ci.T), d(Fi.T), d(Ti.T), d(flags.view(np.uint8))
dO = torch.empty(1000, dtype=torch.uint8, device="cuda"); dX = torch.empty((1000, 3), dtype=torch.float64, device="cuda")
from visual_odometry_amd import _lib
def run(): _lib.check(fe.L.vofe_triangulate_candidates(fe.ctx, dK.data_ptr(), dP.data_ptr(), dC.data_ptr(), dF.data_ptr(), 1, dT.data_ptr(), 12, dI.data_ptr(), 1000, float(np.deg2rad(1)), dO.data_ptr(), dX.data_ptr()))
for _ in range(5): run()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("triangulate kernel, 1000 candidates: %.1f us per launch" % (e0.elapsed_time(e1) * 1e3 / 50))
fe.close()
