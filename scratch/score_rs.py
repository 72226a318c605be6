"""Score-map launch time per strip height (VOFE_SCORE_RS) on 128 KITTI frames, a 4K pair and single frames (HIP events)."""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd
fe = FrontEnd(0); L = fe.L
for name, B in (("kitti", 128), ("stress", 2), ("kitti", 1), ("malaga", 1), ("stress", 1)):
    H, W, seed, n, win, lvl = synthetic.CONFIGS[name]
    a, b = synthetic.sequence(H, W, seed, 2)
    d = torch.from_numpy(np.stack([a, b] * max(1, B // 2))[:B]).cuda()
    score = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    ref = None
    for rs in ("16", "8", "4", "2", ""):
        if rs: os.environ["VOFE_SCORE_RS"] = rs
        else: os.environ.pop("VOFE_SCORE_RS", None)
        ts = []
        for _ in range(12):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.vofe_corner_score(fe.ctx, d.data_ptr(), B, H, W, W, H * W, 3, 3, 0, 0.04, score.data_ptr()))
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        ck = score.view(torch.int32).sum(dtype=torch.int64).item()
        if ref is None: ref = ck
        print("%-7s B=%3d RS=%-7s min %.1f us med %.1f us  (%.2f TB/s algorithmic)%s" % (
            name, B, rs or "default", min(ts), sorted(ts)[6], 5.0 * B * H * W / (min(ts) * 1e-6) / 1e12, "" if ck == ref else "  MISMATCH"), flush=True)
fe.close()
