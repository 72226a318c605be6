"""Batched detector (BASELINE configs[3]: 64 KITTI-shaped sequences, one vofe_gftt_batch call per frame index): time per
call and the per-kernel split (vofe_profile_begin / _end).  usage: python3 tools/gftt_batch_bench.py [batch] [maxCorners] [minDistance]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic, _lib                      # noqa: E402
from visual_odometry_amd.frontend import FrontEnd                    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
maxc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
md = float(sys.argv[3]) if len(sys.argv) > 3 else 10
fe = FrontEnd(0)
L = fe.L
H, W = 376, 1241
d = torch.from_numpy(np.stack([synthetic.sequence(H, W, 100 + s, 1)[0] for s in range(B)])).cuda()
for _ in range(5):
    out, cnt = fe.goodFeaturesToTrack_batch(d, maxc, 0.01, md)
torch.cuda.synchronize()
reps = 30
t0 = time.perf_counter()
for _ in range(reps):
    out, cnt = fe.goodFeaturesToTrack_batch(d, maxc, 0.01, md)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / reps * 1e6
_lib.check(L.vofe_profile_begin(fe.ctx))
for _ in range(4):
    fe.goodFeaturesToTrack_batch(d, maxc, 0.01, md)
cap = 512
names = C.create_string_buffer(cap * 48)
ms = (C.c_float * cap)()
n = C.c_int(0)
_lib.check(L.vofe_profile_end(fe.ctx, cap, names, ms, C.byref(n)))
acc = {}
for i in range(n.value):
    nm = names.raw[i * 48:(i + 1) * 48].split(b"\0", 1)[0].decode()
    acc.setdefault(nm, []).append(ms[i] * 1e3)
print("batch %d x %dx%d maxCorners %d minDistance %g: %.1f us per call = %.2f us per image, %d corners in image 0; kernels (us): %s" % (
    B, W, H, maxc, md, us, us / B, int(cnt[0]), "  ".join("%s %.1f x%.1f" % (k, np.mean(v), len(v) / 4) for k, v in acc.items())))
fe.close()
