"""Detector latency per call (device-resident image in, corners to the host) on the three detector workloads, and the
per-kernel split of one call (vofe_profile_begin / vofe_profile_end).  usage: python3 tools/gftt_bench.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import synthetic, _lib                      # noqa: E402
from visual_odometry_amd.frontend import FrontEnd                    # noqa: E402

fe = FrontEnd(0)
L = fe.L
for name, (h, w, seed, maxc, md) in {"kitti": (376, 1241, 0, 2000, 7), "kitti1000": (376, 1241, 0, 1000, 10), "parking": (480, 640, 3, 1000, 10),
                                     "malaga": (600, 800, 1, 3000, 10), "stress": (2160, 3840, 2, 20000, 10)}.items():
    frames = synthetic.sequence(h, w, seed, 4)
    d = [torch.from_numpy(f).cuda() for f in frames]
    for i in range(10):
        fe.goodFeaturesToTrack(d[i % 4], maxc, 0.01, md)
    torch.cuda.synchronize()
    reps = 200 if name != "stress" else 50
    t0 = time.perf_counter()
    for i in range(reps):
        c = fe.goodFeaturesToTrack(d[i % 4], maxc, 0.01, md)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    _lib.check(L.vofe_profile_begin(fe.ctx))
    for i in range(8):
        fe.goodFeaturesToTrack(d[i % 4], maxc, 0.01, md)
    cap = 512
    names = C.create_string_buffer(cap * 48)
    ms = (C.c_float * cap)()
    cnt = C.c_int(0)
    _lib.check(L.vofe_profile_end(fe.ctx, cap, names, ms, C.byref(cnt)))
    acc = {}
    for i in range(cnt.value):
        nm = names.raw[i * 48:(i + 1) * 48].split(b"\0", 1)[0].decode()
        acc.setdefault(nm, []).append(ms[i] * 1e3)
    print("%-10s %dx%d maxCorners %d minDistance %g: %.1f us per call, %d corners; kernels (us, event-bracketed): %s" % (
        name, w, h, maxc, md, us, c.shape[0], "  ".join("%s %.1f x%.1f" % (k, np.mean(v), len(v) / 8) for k, v in acc.items())))
fe.close()
