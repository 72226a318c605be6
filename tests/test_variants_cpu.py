"""How far the plausible cv2 arithmetics lie from the canonical one the kernels follow (CPU only).

cv2 is absent from the image, so parity with a cv2 binary stays UNPINNED; what these tests pin is the
SPREAD between the arithmetic orders a cv2 build may use at the points SURVEY Appendix A marks as build dependent
(oracle/vofe_oracle.h, vofe_cpu_set_variant).  The reference's calls: continuous.py:145-151 (calcOpticalFlowPyrLK inside
_track_keypoints), continuous.py:380-385 and initialization.py:41-42 (goodFeaturesToTrack).  Bars = BASELINE.json
north_star: positions within 0.05 px, status flags >= 99.5 % identical, corner sets equal."""
import os
import sys

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_variant_zero_is_the_canonical_arithmetic_and_is_restored():
    z = np.load(os.path.join(GOLDEN, "variants_small.npz"))
    a, b, pts = z["img0"], z["img1"], z["pts"]
    ref = oracle.track_keypoints(pts.T, a, b, (21, 21), 2)
    with oracle.variant():
        same = oracle.track_keypoints(pts.T, a, b, (21, 21), 2)
    with oracle.variant(lk_sum=1, sobel=2):
        pass
    after = oracle.track_keypoints(pts.T, a, b, (21, 21), 2)
    for x, y, w in zip(ref, same, after):
        assert np.array_equal(x, y) and np.array_equal(x, w)
    with pytest.raises(ValueError):
        with oracle.variant(lk_sum=9):
            pass


def test_variant_golden_fixtures():
    """tests/golden/variants_small.npz (written by make_golden.py from this oracle): every variant's outputs on one
    small case, so that a variant's definition cannot drift unnoticed.  NOT cv2 outputs."""
    z = np.load(os.path.join(GOLDEN, "variants_small.npz"), allow_pickle=False)
    a, b, pts = z["img0"], z["img1"], z["pts"]
    for name, kw in oracle.LK_VARIANTS.items():
        with oracle.variant(**kw):
            t, s, e = oracle.track_keypoints(pts.T, a, b, (21, 21), 2)
        assert np.array_equal(t, z["lk_%s_next" % name]) and np.array_equal(s, z["lk_%s_status" % name]), name
        assert np.array_equal(e, z["lk_%s_err" % name]), name
    for name, kw in dict(oracle.SCORE_VARIANTS, **oracle.HARRIS_VARIANTS).items():
        harris = name.startswith("harris")
        with oracle.variant(**kw):
            assert np.array_equal(oracle.corner_score(a, harris), z["score_%s" % name]), name
            assert np.array_equal(oracle.goodFeaturesToTrack(a, 48, 0.01, 5, useHarrisDetector=harris),
                                  z["corners_%s" % name]), name


def test_all_sum_orders_agree_bit_for_bit_while_every_partial_sum_is_exact():
    """On an image whose gradients are so small that every window sum stays below 2**24, any f32 summation order is
    exact, so every variant must return the canonical tracks bit for bit: the variants differ from the canonical
    arithmetic in rounding only, not in what they sum."""
    rng = np.random.default_rng(3)
    img = (rng.integers(0, 4, (80, 100)) * 3 + 100).astype(np.uint8)            # |Ix|, |Iy| <= 16 * 9
    nxt = np.roll(img, 1, 1)
    pts = np.float32([[30.25, 30.5], [50.75, 40.125], [64.5, 20.25]])
    ref = oracle.calcOpticalFlowPyrLK(img, nxt, pts, None, (21, 21), 0, minEigThreshold=1e-7)
    assert ref[1].all()
    for name, kw in oracle.LK_VARIANTS.items():
        with oracle.variant(**kw):
            got = oracle.calcOpticalFlowPyrLK(img, nxt, pts, None, (21, 21), 0, minEigThreshold=1e-7)
        for x, y in zip(ref, got):
            assert np.array_equal(x, y), name


def test_spread_of_the_plausible_cv2_arithmetics_stays_inside_the_north_star_bars():
    """BASELINE configs[1] (KITTI 1241x376, 2000 points, 21x21), configs[2] (Malaga 800x600, 3000), configs[4]'s frame
    (3840x2160, 31x31, 5 levels; a tenth of its 20 000 points here, all of them in tools/variant_spread.py's full run,
    profiles/r03_variant_spread.jsonl), the Parking shape (640x480, 11x11) and the photographs shipped with the image.
    Canonical vs every variant.  Asserted: status flags >= 99.5 % identical, >= 99.5 % of the points tracked by both
    within 0.05 px (a Gauss-Newton walk that sits on a bistable point can land elsewhere under ANY change of rounding;
    seen once in ~10 000 points, on a photograph), corner lists equal as sets up to 0.5 % (a corner whose score is one
    ulp from the quality threshold, or from a neighbour's score inside minDistance, can change sides)."""
    import variant_spread
    rows = variant_spread.run(quick=True, out=open(os.devnull, "w"))
    lk = [r for r in rows if r["kind"] == "lk"]
    co = [r for r in rows if r["kind"] == "corners"]
    assert {r["workload"] for r in lk} >= {"kitti", "malaga", "parking", "stress"}
    assert {r["variant"] for r in lk} == set(oracle.LK_VARIANTS)
    for r in lk:
        assert r["status_agree"] >= 0.995, r
        assert r["tracked_canonical"] > 0.5 * r["n"] or r["workload"].startswith("photo"), r
        assert r["n_over_0p05"] <= 0.005 * r["tracked_canonical"], r
    worst = max(r["max_px"] for r in lk if not r["workload"].startswith("photo"))
    assert worst < 0.01, worst          # on the BASELINE shapes the spread is three orders below the 0.05 px bar
    for r in co:
        n = max(r["n_canonical"], 1)
        assert abs(r["n_canonical"] - r["n_variant"]) <= max(1, 0.005 * n), r
        assert r["only_canonical"] + r["only_variant"] <= max(2, 0.005 * n), r
        assert r["score_max_diff_rel_to_max"] < 2e-6, r
