"""Bandwidth of the streaming kernels (pyramid chain, Scharr, corner score, NMS) on the stress config
(3840x2160) and on a 64-image KITTI batch; HIP-event timed per API call."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from visual_odometry_amd import synthetic, _lib
from visual_odometry_amd.frontend import FrontEnd, Pyramid
fe = FrontEnd(0); L = fe.L
def timeit(fn, n=30):
    for _ in range(5): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for x, y in ev:
        x.record(); fn(); y.record()
    torch.cuda.synchronize()
    return sorted(x.elapsed_time(y) for x, y in ev)[n // 2] * 1e-3
def levels(h, w, win, ml):
    out = [(h, w)]
    for _ in range(ml):
        h, w = (h + 1) // 2, (w + 1) // 2
        out.append((h, w))
    return out
for name, H, W, B, win, ml in (("stress 3840x2160 x1", 2160, 3840, 1, (31, 31), 4), ("stress x8", 2160, 3840, 8, (31, 31), 4), ("kitti x64", 376, 1241, 64, (21, 21), 3), ("kitti x2", 376, 1241, 2, (21, 21), 3)):
    img = torch.randint(0, 256, (B, H, W), dtype=torch.uint8, device="cuda")
    lv = levels(H, W, win, ml)
    px = [h * w for h, w in lv]
    for deriv in (False, True):
        pyr = Pyramid(fe, B, H, W, win, ml, deriv)
        t = timeit(lambda: _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, B, img.data_ptr(), W, H * W)))
        chain = sum(px[k] + px[k + 1] for k in range(ml)) + 2 * px[0]          # pyrDown chain + level-0 copy
        byt = B * (chain + (5 * sum(px) if deriv else 0))
        print("%-22s pyramid%s: %8.1f us  %7.1f GB/s algorithmic (%.1f MB)" % (name, "+scharr" if deriv else "       ", t * 1e6, byt / t / 1e9, byt / 1e6))
        pyr.free()
    sc = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    t = timeit(lambda: _lib.check(L.vofe_corner_score(fe.ctx, img.data_ptr(), B, H, W, W, H * W, 3, 3, 0, 0.04, sc.data_ptr())))
    print("%-22s corner_score    : %8.1f us  %7.1f GB/s algorithmic" % (name, t * 1e6, B * px[0] * 5 / t / 1e9))
fe.close()
