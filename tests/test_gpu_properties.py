"""Oracle-free properties of the HIP path at BASELINE.json's full sizes (GPU against itself, or against what the
result must satisfy by definition).  The parity tests compare kernels with the CPU oracle; these do not use it at all, so
they hold whatever the oracle does:

  * pyrDown / Scharr / score map are local: a frame cropped by a multiple of 2^levels reproduces the full frame's planes
    away from the crop's border, bit for bit;
  * the tracker is invariant, bit for bit, under a constant intensity offset (J - I, the Scharr planes and every window sum
    are unchanged while nothing saturates), and equivariant under such a crop;
  * what goodFeaturesToTrack returns satisfies cv2's contract by itself: integer pixels off the frame's border, local 3x3
    maxima of the score map above qualityLevel * max, strongest first with the tie rule, no two closer than minDistance,
    and greedy-maximal (every rejected candidate has a stronger accepted corner within minDistance) -- checked with
    scipy's cKDTree on the KITTI, Malaga and 3840x2160 configurations;
  * forward-backward status: a tracked point's backward track ends within 1 px of where it started (continuous.py:155-158).

Reference calls: continuous.py:145-151, 380-385."""
import numpy as np
import pytest

from visual_odometry_amd import synthetic

pytestmark = pytest.mark.gpu


def _unsaturated(frames, lo=20, span=0.7):
    return [np.clip(f.astype(np.float32) * span + lo, 0, 255).astype(np.uint8) for f in frames]


def test_pyramid_scharr_and_score_map_are_local_under_a_crop(fe):
    import torch
    from visual_odometry_amd.frontend import Pyramid
    H, W, seed, _, win, lvl = synthetic.CONFIGS["kitti"]
    full = synthetic.sequence(H, W, seed, 1)[0]
    oy, ox = 16, 24                                    # multiples of 2^lvl: the levels stay aligned
    crop = np.ascontiguousarray(full[oy:, ox:])
    pf = Pyramid(fe, 1, H, W, win, lvl, True)
    pc = Pyramid(fe, 1, crop.shape[0], crop.shape[1], win, lvl, True)
    pf.build(torch.from_numpy(full).cuda(), 0)
    pc.build(torch.from_numpy(crop).cuda(), 0)
    reach = 0                                          # pixels of a level that the crop's border can influence
    for level in range(lvl + 1):
        lf, df = pf.download(0, level, deriv=True)
        lc, dc = pc.download(0, level, deriv=True)
        sy, sx = oy >> level, ox >> level
        m = reach + 1                                  # + the Scharr tap
        assert np.array_equal(lc[reach:, reach:], lf[sy + reach:, sx + reach:][: lc.shape[0] - reach, : lc.shape[1] - reach]), level
        assert np.array_equal(dc[m:-1, m:-1], df[sy + m:, sx + m:][: dc.shape[0] - m - 1, : dc.shape[1] - m - 1]), level
        reach = (reach + 2 + 1) // 2 + 1               # 5-tap pyrDown: ceil((reach + 2) / 2), one spare
    pf.free(); pc.free()
    for harris in (False, True):
        sf, sc = fe.corner_score(full, harris), fe.corner_score(crop, harris)
        assert np.array_equal(sc[2:, 2:], sf[oy + 2:, ox + 2:])
    H4, W4 = synthetic.CONFIGS["stress"][:2]
    big = synthetic.sequence(H4, W4, 2, 1)[0]
    sf, sc = fe.corner_score(big), fe.corner_score(np.ascontiguousarray(big[8:-40, 8:-24]))
    assert np.array_equal(sc[2:-2, 2:-2], sf[10:-42, 10:-26])
    for bs, ks in ((5, 3), (7, 7)):                    # the any-geometry kernel: reach = box half + Sobel half
        r = bs // 2 + ks // 2
        sf, sc = fe.corner_score(full, False, 0.04, bs, ks), fe.corner_score(crop, False, 0.04, bs, ks)
        assert np.array_equal(sc[r:, r:], sf[oy + r:, ox + r:]), (bs, ks)


def test_tracker_is_invariant_under_an_intensity_offset_and_equivariant_under_a_crop(fe):
    H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
    a, b = _unsaturated(synthetic.sequence(H, W, seed, 2))
    pts = synthetic.pad_points(fe.goodFeaturesToTrack(a, n, 0.01, 7), n, H, W, seed)
    ref = fe.track_keypoints(pts.T, a, b, win, lvl)
    assert ref[1].mean() > 0.9
    for c in (1, 40):
        got = fe.track_keypoints(pts.T, a + np.uint8(c), b + np.uint8(c), win, lvl)
        assert all(np.array_equal(x, y) for x, y in zip(got, ref)), c
    # crop by multiples of 2^lvl; points whose windows stay clear of both frames' borders on every level, where the
    # REFLECT_101 padding differs: identical tracks up to the float rounding of positions that differ by the offset
    oy, ox = 16, 32
    ac, bc = np.ascontiguousarray(a[oy:, ox:]), np.ascontiguousarray(b[oy:, ox:])
    keep = (pts[:, 0] > ox + 150) & (pts[:, 0] < W - 150) & (pts[:, 1] > oy + 135) & (pts[:, 1] < H - 135)
    assert keep.sum() > 100
    sub = pts[keep]
    t0, s0, e0 = fe.track_keypoints(sub.T, a, b, win, lvl)
    t1, s1, e1 = fe.track_keypoints((sub - np.float32([ox, oy])).T, ac, bc, win, lvl)
    # (positions that differ by the offset round differently in the last bit, which can move a walk that ends exactly at
    # the convergence threshold by one iteration: hence bounds instead of equality, an order below the 0.05 px bar)
    assert np.mean(s0 == s1) >= 0.995
    ok = (s0 == 1) & (s1 == 1)
    d = np.abs((t1 + np.float32([[ox], [oy]])) - t0)[:, ok].max(0)
    assert d.max() < 0.03, d.max()
    assert np.mean(d < 1e-3) > 0.97 and np.median(d) < 1e-4
    assert np.allclose(e0[ok], e1[ok], rtol=5e-2, atol=1e-2)


def test_forward_backward_status_means_the_backward_track_returns(fe):
    H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
    a, b = synthetic.sequence(H, W, seed, 2)
    pts = synthetic.pad_points(fe.goodFeaturesToTrack(a, n, 0.01, 7), n, H, W, seed)
    t, s, e = fe.track_keypoints(pts.T, a, b, win, lvl)
    f, fs, _ = fe.calcOpticalFlowPyrLK(a, b, pts, None, win, lvl, (3, 30, 0.01), 0, 1e-3)
    back, _, _ = fe.calcOpticalFlowPyrLK(b, a, f, None, win, lvl, (3, 30, 0.01), 0, 1e-3)
    assert np.array_equal(t.T, f)                                   # the fused launch's forward half is the forward call
    d = np.abs(pts - back).max(1)
    assert np.array_equal(s, fs.ravel() & (d < 1.0))               # continuous.py:155-158, backward status ignored
    assert s.sum() > 0.9 * n


@pytest.mark.parametrize("cfg,maxc,q,md", [("kitti", 2000, 0.01, 7), ("kitti", 1000, 0.01, 10), ("malaga", 3000, 0.01, 10),
                                           ("stress", 20000, 0.01, 10), ("kitti", 0, 0.05, 5.5), ("parking", 1000, 0.01, 10)])
def test_detector_output_satisfies_the_cv2_contract_by_itself(fe, cfg, maxc, q, md):
    from scipy.spatial import cKDTree
    H, W, seed = synthetic.CONFIGS[cfg][:3]
    img = synthetic.sequence(H, W, seed, 1)[0]
    c = fe.goodFeaturesToTrack(img, maxc, q, md)
    assert c is not None and c.dtype == np.float32 and c.shape[1:] == (1, 2)
    xy = c.reshape(-1, 2)
    assert np.array_equal(xy, np.round(xy))                          # integer pixels, no sub-pixel refinement
    x, y = xy[:, 0].astype(int), xy[:, 1].astype(int)
    assert x.min() >= 1 and y.min() >= 1 and x.max() <= W - 2 and y.max() <= H - 2
    if maxc > 0:
        assert len(xy) <= maxc
    S = fe.corner_score(img)
    thr = np.float32(np.float64(S.max()) * q)
    v = S[y, x]
    assert (v > thr).all()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            assert (v >= S[y + dy, x + dx]).all()                    # 3x3 local maxima (plateaus allowed)
    idx = y.astype(np.int64) * W + x
    assert len(np.unique(idx)) == len(idx)
    order_ok = (v[:-1] > v[1:]) | ((v[:-1] == v[1:]) & (idx[:-1] > idx[1:]))
    assert order_ok.all()                                            # strongest first, ties: larger address first
    tree = cKDTree(xy.astype(np.float64))
    if md >= 1:
        pairs = tree.query_pairs(md * (1 - 1e-12))                   # strictly closer than minDistance
        assert len(pairs) == 0
    # greedy-maximal: every candidate that was not returned is blocked by a STRONGER returned corner within minDistance --
    # or the list was cut at maxCorners, in which case only candidates ranked before the last returned corner are checked
    from scipy import ndimage
    T = np.where(S > thr, S, np.float32(0))
    loc = (T == ndimage.grey_dilation(T, size=(3, 3), mode="mirror")) & (T != 0)
    loc[0, :] = loc[-1, :] = False
    loc[:, 0] = loc[:, -1] = False
    taken = np.zeros_like(loc)
    taken[y, x] = True
    ys, xs = np.nonzero(loc & ~taken)
    if md >= 1 and len(ys):
        cv, cidx = S[ys, xs], ys.astype(np.int64) * W + xs
        if maxc > 0 and len(xy) == maxc:                             # cut: candidates ranked behind the last corner are free
            ahead = (cv > v[-1]) | ((cv == v[-1]) & (cidx > idx[-1]))
            ys, xs, cv, cidx = ys[ahead], xs[ahead], cv[ahead], cidx[ahead]
        near = tree.query_ball_point(np.stack([xs, ys], 1).astype(np.float64), md * (1 - 1e-12))
        for k in range(len(ys)):
            nb = np.asarray(near[k], dtype=int)
            assert len(nb), (xs[k], ys[k])
            stronger = (v[nb] > cv[k]) | ((v[nb] == cv[k]) & (idx[nb] > cidx[k]))
            assert stronger.any(), (xs[k], ys[k])
    elif md < 1 and (maxc == 0 or len(xy) < maxc):
        assert len(ys) == 0                                          # no suppression: every candidate is returned
