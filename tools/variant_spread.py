"""How far do the plausible cv2 arithmetics lie from the canonical one?  (CPU only.)

cv2 is absent from the image, so parity with a cv2 binary cannot be pinned.  What CAN be measured is the
spread between the arithmetic orders a cv2 build may use at the points SURVEY Appendix A marks with a warning sign
(oracle/vofe_oracle.h, vofe_cpu_set_variant): f32 window sums in raster / 4-lane / 128-bit universal-intrinsics / 8-lane
order with and without FMA (A.5 steps 4, 6), FMA or scale-after in the Sobel smoothing pass, FMA in the min-eigenvalue
sqrt argument, running f64 box sums (A.1), the Harris evaluation orders, IPP's threshold rule (A.2 step 2).

For every workload (BASELINE configs[1], [2], [4] shapes, the Parking shape, the photographs in the image) the canonical
oracle and each variant run on the same inputs; the tool prints one JSON line per (workload, variant):
  LK:      max |position difference| over points tracked by both, status agreement (forward and after the FB check)
  corners: corner count both ways, first rank at which the lists differ (-1 = identical), size of the set difference
Usage: python tools/variant_spread.py [--quick] [--threads N] > profiles/r03_variant_spread.jsonl
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from visual_odometry_amd import synthetic  # noqa: E402

PHOTO_DIR = "/opt/conda/lib/python3.9/site-packages/skimage/data"


def photos():
    """Real photographs shipped with the image (load-if-present): (name, frame0, frame1)."""
    out = []
    try:
        from PIL import Image
    except ImportError:
        return out

    def load(name):
        p = os.path.join(PHOTO_DIR, name)
        if not os.path.exists(p):
            return None
        return np.ascontiguousarray(np.asarray(Image.open(p).convert("L"), np.uint8))
    left, right = load("motorcycle_left.png"), load("motorcycle_right.png")
    if left is not None and right is not None and left.shape == right.shape:
        out.append(("photo_motorcycle_stereo", left, right))
    for name in ("camera.png", "brick.png", "coins.png"):
        a = load(name)
        if a is not None:
            b = np.roll(a, (1, 2), (0, 1))         # a 2 x 1 px shift of the same photograph
            out.append(("photo_" + name[:-4], a, b))
    return out


def workloads(quick):
    w = []
    for name, frac in (("kitti", 1.0), ("malaga", 1.0), ("parking", 1.0), ("stress", 0.1 if quick else 1.0)):
        H, W, seed, n, win, lvl = synthetic.CONFIGS[name]
        n = max(200, int(n * frac))
        f0, f1 = synthetic.sequence(H, W, seed=seed, n_frames=2)
        md = 7 if name == "kitti" else 10
        w.append(dict(name=name, f0=f0, f1=f1, n=n, win=win, lvl=lvl, md=md, seed=seed))
    for name, a, b in photos():
        w.append(dict(name=name, f0=a, f1=b, n=1000, win=(21, 21), lvl=3, md=10, seed=5))
    return w


def lk_compare(ref, var):
    (t0, s0, e0), (t1, s1, e1) = ref, var
    both = (s0 == 1) & (s1 == 1)
    d = np.abs(t0 - t1).max(axis=0)
    return dict(n=int(len(s0)), status_agree=float(np.mean(s0 == s1)), tracked_canonical=int(s0.sum()),
                tracked_variant=int(s1.sum()), max_px=float(d[both].max()) if both.any() else 0.0,
                n_over_0p05=int((d[both] > 0.05).sum()), n_bit_different=int((d[both] > 0).sum()),
                max_err_diff=float(np.abs(e0 - e1)[both].max()) if both.any() else 0.0)


def corner_compare(c0, c1):
    a = np.zeros((0, 2), np.float32) if c0 is None else c0.reshape(-1, 2)
    b = np.zeros((0, 2), np.float32) if c1 is None else c1.reshape(-1, 2)
    m = min(len(a), len(b))
    neq = np.nonzero((a[:m] != b[:m]).any(axis=1))[0]
    first = int(neq[0]) if len(neq) else (-1 if len(a) == len(b) else m)
    sa, sb = set(map(tuple, a.tolist())), set(map(tuple, b.tolist()))
    return dict(n_canonical=len(a), n_variant=len(b), first_differing_rank=first, only_canonical=len(sa - sb),
                only_variant=len(sb - sa))


def run(quick=False, threads=None, out=sys.stdout):
    threads = threads or min(16, os.cpu_count() or 1)
    rows = []
    for wl in workloads(quick):
        f0, f1 = wl["f0"], wl["f1"]
        H, W = f0.shape
        corners = oracle.goodFeaturesToTrack(f0, wl["n"], 0.01, wl["md"], nthreads=threads)
        pts = synthetic.pad_points(corners, wl["n"], H, W, wl["seed"])
        kw = dict(winSize=wl["win"], maxLevel=wl["lvl"], nthreads=threads)
        ref = oracle.track_keypoints(pts.T, f0, f1, **kw)
        for vname, vkw in oracle.LK_VARIANTS.items():
            with oracle.variant(**vkw):
                var = oracle.track_keypoints(pts.T, f0, f1, **kw)
            row = dict(kind="lk", workload=wl["name"], shape=[H, W], win=list(wl["win"]), variant=vname, **lk_compare(ref, var))
            rows.append(row)
            print(json.dumps(row), file=out, flush=True)
        for harris, table in ((False, oracle.SCORE_VARIANTS), (True, dict(oracle.HARRIS_VARIANTS, **oracle.SCORE_VARIANTS))):
            c0 = oracle.goodFeaturesToTrack(f0, wl["n"], 0.01, wl["md"], useHarrisDetector=harris, nthreads=threads)
            s0 = oracle.corner_score(f0, harris, nthreads=threads)
            for vname, vkw in table.items():
                if harris and vname == "mineig_fma":
                    continue
                with oracle.variant(**vkw):
                    c1 = oracle.goodFeaturesToTrack(f0, wl["n"], 0.01, wl["md"], useHarrisDetector=harris, nthreads=threads)
                    s1 = oracle.corner_score(f0, harris, nthreads=threads)
                rel = float(np.abs(s1 - s0).max() / max(float(np.abs(s0).max()), 1e-30))
                row = dict(kind="corners", detector="harris" if harris else "shi_tomasi", workload=wl["name"], shape=[H, W],
                           variant=vname, score_px_different=int((s0 != s1).sum()), score_max_diff_rel_to_max=rel,
                           **corner_compare(c0, c1))
                rows.append(row)
                print(json.dumps(row), file=out, flush=True)
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="stress shape with a tenth of its points")
    ap.add_argument("--threads", type=int, default=None)
    a = ap.parse_args()
    run(a.quick, a.threads)
