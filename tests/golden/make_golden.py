"""Writes tests/golden/frontend_small.npz and variants_small.npz from the CPU oracle (NOT from cv2: cv2 is absent from the
image and the reference ships no fixtures).  Inputs are seeded synthetic frames;
rerun only when the oracle's definition changes on purpose:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from visual_odometry_amd import synthetic  # noqa: E402

a, b = synthetic.sequence(48, 64, seed=123, n_frames=2)
corners = oracle.goodFeaturesToTrack(a, 32, 0.01, 5)
pts = synthetic.pad_points(corners, 32, 48, 64, 123)
n, s, e = oracle.calcOpticalFlowPyrLK(a, b, pts, None, (11, 11), 2, (3, 30, 0.01), 0, 1e-4)
t, fs, fe = oracle.track_keypoints(pts.T, a, b, (11, 11), 2)
# cv2's optional blockSize / gradientSize (the reference passes neither): a few geometries of the generic score path
GEOM = [(5, 3), (3, 5), (7, 7), (2, 1)]
geom = {}
for bs, ks in GEOM:
    geom["score_b%d_k%d" % (bs, ks)] = oracle.corner_score(a, blockSize=bs, ksize=ks)
    geom["harris_b%d_k%d" % (bs, ks)] = oracle.corner_score(a, True, blockSize=bs, ksize=ks)
    geom["corners_b%d_k%d" % (bs, ks)] = oracle.goodFeaturesToTrack(a, 32, 0.01, 5, blockSize=bs, gradientSize=ks)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "frontend_small.npz"),
                    img0=a, img1=b, pyr1=oracle.pyr_down(a), scharr0=oracle.scharr(a), score=oracle.corner_score(a),
                    harris=oracle.corner_score(a, True), corners=corners, pts=pts, lk_next=n, lk_status=s, lk_err=e,
                    fb_next=t, fb_status=fs, fb_err=fe, **geom)
print("written", len(pts), "points,", corners.shape[0], "corners")

# The alternative arithmetics (oracle.variant, SURVEY Appendix A warning points) on a second small case: what each of them
# returns, so that a change of a variant's definition is caught like a change of the canonical one.
a2, b2 = synthetic.sequence(72, 96, seed=321, n_frames=2)
c2 = oracle.goodFeaturesToTrack(a2, 48, 0.01, 5)
p2 = synthetic.pad_points(c2, 48, 72, 96, 321)
out = dict(img0=a2, img1=b2, pts=p2)
for name, kw in oracle.LK_VARIANTS.items():
    with oracle.variant(**kw):
        t, fs, fe = oracle.track_keypoints(p2.T, a2, b2, (21, 21), 2)
    out["lk_%s_next" % name], out["lk_%s_status" % name], out["lk_%s_err" % name] = t, fs, fe
for name, kw in dict(oracle.SCORE_VARIANTS, **oracle.HARRIS_VARIANTS).items():
    with oracle.variant(**kw):
        out["score_%s" % name] = oracle.corner_score(a2, name.startswith("harris"))
        out["corners_%s" % name] = oracle.goodFeaturesToTrack(a2, 48, 0.01, 5, useHarrisDetector=name.startswith("harris"))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants_small.npz"), **out)
print("written", len(out), "variant arrays")
