"""Bootstrap front end (reference initialization.py:41-70) with the reference's settings per dataset: host-to-host time of
BootstrapTracker.detect and .track (frame upload, level-0 pyramid + Scharr, forward LK, status / moved-far-enough
compaction, download of the pairs) against the same steps on the CPU oracle (1 and 16 threads).
usage: python3 tools/bootstrap_bench.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                                           # noqa: E402  (the timed CPU side only)
from visual_odometry_amd import synthetic, BOOTSTRAP_PRESETS           # noqa: E402
from visual_odometry_amd.bootstrap import BootstrapTracker             # noqa: E402
from visual_odometry_amd.frontend import FrontEnd                      # noqa: E402

SHAPES = {"kitti": (376, 1241, 2), "parking": (480, 640, 1), "malaga": (600, 800, 2)}
fe = FrontEnd(0)
for name, (h, w, stride) in SHAPES.items():
    p = dict(BOOTSTRAP_PRESETS[name])
    frames = synthetic.sequence(h, w, seed=31 + h, n_frames=(p["n_images"] - 1) * stride + 1)[::stride]
    bt = BootstrapTracker(h, w, frontend=fe, **p)
    t_det, t_trk, counts = [], [], []
    for rep in range(12):
        t0 = time.perf_counter()
        n0 = bt.detect(frames[0])
        t_det.append(time.perf_counter() - t0)
        for img in frames[1:]:
            t0 = time.perf_counter()
            go, gn = bt.track(img)
            t_trk.append(time.perf_counter() - t0)
            if rep == 0:
                counts.append(len(go))
    cpu = {}
    for nt in (1, 16):
        t0 = time.perf_counter()
        c = oracle.goodFeaturesToTrack(frames[0], p["maxCorners"], p["qualityLevel"], p["minDistance"])
        t1 = time.perf_counter()
        good = c
        for i in range(len(frames) - 1):
            nxt, st, _ = oracle.calcOpticalFlowPyrLK(frames[i], frames[i + 1], good, None, winSize=p["winSize"], maxLevel=0,
                                                     criteria=(3, 30, 0.01), nthreads=nt)
            good = nxt[st.flatten() == 1]
        cpu[nt] = ((t1 - t0) * 1e6, (time.perf_counter() - t1) / (len(frames) - 1) * 1e6)
    print("%-8s %dx%d win %s, %d images, %d corners, pairs after each image %s: detect %.0f us, track %.0f us per image "
          "(median, host to host); CPU oracle detect %.0f us, LK alone %.0f us per image on 1 thread, %.0f us on 16" % (
              name, w, h, p["winSize"], p["n_images"], n0, counts, np.median(t_det[2:]) * 1e6,
              np.median(t_trk[2 * (len(frames) - 1):]) * 1e6, cpu[1][0], cpu[1][1], cpu[16][1]), flush=True)
