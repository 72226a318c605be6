"""CPU tests (no GPU): pin the C++ oracle with analytic known-answer tests,
with an independent numpy restatement, and with committed golden fixtures.  cv2 is absent from
the image and the reference has no tests, so parity with cv2 itself stays UNPINNED; these tests
pin the restatement against transcription slips and regressions."""
import os

import numpy as np
import pytest

import oracle
from oracle import np_twin as T
from visual_odometry_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------ analytic KATs
def test_constant_image():
    a = np.full((40, 50), 93, np.uint8)
    assert np.all(oracle.pyr_down(a) == 93)
    assert not oracle.scharr(a).any()
    assert not oracle.corner_score(a).any()
    assert oracle.goodFeaturesToTrack(a, 10, 0.01, 3) is None


def test_horizontal_ramp():
    c = 3
    a = np.tile((c * np.arange(60)).astype(np.uint8), (30, 1))
    d = oracle.scharr(a)
    assert np.all(d[:, 1:-1, 0] == 32 * c) and not d[..., 1].any()
    assert not d[:, 0, 0].any() and not d[:, -1, 0].any()          # REFLECT_101: t0[-1] == t0[1]
    p = oracle.pyr_down(a)
    assert np.array_equal(p[:, 1:-1], a[::2, ::2][:, 1:-1][:, :p.shape[1] - 2])
    s = oracle.corner_score(a)
    assert np.abs(s[:, 2:-2]).max() < 1e-9                          # rank-1 structure tensor -> lambda_min 0


def test_pyr_down_hand_computed():
    a = np.zeros((10, 10), np.uint8)
    a[4, 4] = 255
    k = np.array([1, 4, 6, 4, 1])
    # out(X,Y) = (sum_ij k_i k_j a[2Y-2+j, 2X-2+i] + 128) >> 8: single tap
    exp = np.zeros((5, 5), np.int64)
    for Y in range(5):
        for X in range(5):
            j, i = 4 - (2 * Y - 2), 4 - (2 * X - 2)
            if 0 <= j < 5 and 0 <= i < 5:
                exp[Y, X] = (k[j] * k[i] * 255 + 128) >> 8
    assert np.array_equal(oracle.pyr_down(a), exp.astype(np.uint8))
    # REFLECT_101 at the edge: a bright pixel in column 1 is seen twice by output column 0
    b = np.zeros((5, 5), np.uint8)
    b[2, 1] = 255
    assert oracle.pyr_down(b)[1, 0] == ((6 * (4 + 4) * 255 + 128) >> 8)


def test_bilinear_weights_sum():
    rng = np.random.default_rng(0)
    for fx, fy in rng.uniform(0, 1, (2000, 2)).astype(np.float32):
        w = T._weights(fx, fy, 0, 0)
        assert sum(w) == 16384


def negative_weight_points(n=4, x0=40.0, y0=40.0):
    """f32 points whose level-0 window origin (pt - 10) has bilinear weight w11 == -1 (21x21 window)."""
    rng = np.random.default_rng(5)
    out = []
    while len(out) < n:
        x = np.float32(x0 + rng.uniform(0, 1))
        y = np.float32(y0 + rng.uniform(0, 1e-4))
        fx, fy = x - np.float32(10), y - np.float32(10)
        if T._weights(fx, fy, int(np.floor(fx)), int(np.floor(fy)))[3] < 0:
            out.append((x, y))
    return np.float32(out)


def test_negative_fourth_weight_exists_and_is_handled():
    """16384 - w00 - w01 - w10 can be -1; the integer paths must treat it as signed."""
    pts = negative_weight_points()
    assert len(pts) >= 2
    a, b = synthetic.sequence(80, 90, 3, 2)
    n0, s0, e0 = oracle.calcOpticalFlowPyrLK(a, b, pts, None, (21, 21), 0)
    n1, s1, e1 = T.lk(a, b, pts, (21, 21), 0)
    assert np.array_equal(n0.reshape(-1, 2), n1) and np.array_equal(s0.ravel(), s1) and np.array_equal(e0.ravel(), e1)


def test_isolated_square_corners_and_tie_order():
    a = np.zeros((60, 80), np.uint8)
    a[20:36, 30:50] = 200
    c = oracle.goodFeaturesToTrack(a, 0, 0.5, 3).reshape(-1, 2)
    assert len(c) == 4
    assert {tuple(p) for p in c} <= {(x, y) for x in (29, 30, 49, 50) for y in (19, 20, 35, 36)}
    s = oracle.corner_score(a)
    v = [s[int(y), int(x)] for x, y in c]
    idx = [int(y) * 80 + int(x) for x, y in c]
    for i in range(3):      # descending score; exact ties broken by the larger linear index first
        assert v[i] > v[i + 1] or (v[i] == v[i + 1] and idx[i] > idx[i + 1])


def test_min_distance_strictness():
    a = synthetic.sequence(90, 120, 4, 1)[0]
    for md in (1, 2.5, 7, 10.4):
        c = oracle.goodFeaturesToTrack(a, 0, 0.01, md).reshape(-1, 2).astype(np.float64)
        d2 = ((c[:, None] - c[None]) ** 2).sum(-1) + np.eye(len(c)) * 1e9
        assert d2.min() >= md * md                     # strict '<' rejects, so == minDistance is kept
    c1 = oracle.goodFeaturesToTrack(a, 0, 0.01, 0.99)
    c0 = oracle.goodFeaturesToTrack(a, 0, 0.01, 0)
    assert np.array_equal(c0, c1)                      # minDistance < 1: no suppression at all


def test_lk_ground_truth_flow_and_failures():
    h, w = 200, 260
    f = synthetic.sequence(h, w, seed=2, n_frames=2, noise_sigma=0.0)
    c = oracle.goodFeaturesToTrack(f[0], 200, 0.01, 8).reshape(-1, 2)
    inner = (c[:, 0] > 30) & (c[:, 0] < w - 30) & (c[:, 1] > 30) & (c[:, 1] < h - 30)
    nxt, st, err = oracle.calcOpticalFlowPyrLK(f[0], f[1], c[inner], None, (21, 21), 3)
    truth = synthetic.true_flow(c[inner], h, w, 0, 1)
    ok = st.ravel() == 1
    assert ok.mean() > 0.95
    assert np.median(np.abs(nxt.reshape(-1, 2)[ok] - truth[ok])) < 0.05
    # window entirely outside the image -> status 0 and err 0 at level 0
    out = np.float32([[-40, 10], [w + 40, 10], [10, h + 40]])
    _, st, err = oracle.calcOpticalFlowPyrLK(f[0], f[1], out, None, (21, 21), 3)
    assert not st.any() and not err.any()
    flat = np.full((h, w), 50, np.uint8)
    _, st, _ = oracle.calcOpticalFlowPyrLK(flat, flat, np.float32([[100, 100]]), None, (21, 21), 3)
    assert st[0, 0] == 0                               # flat patch fails minEig / det


def test_pyramid_early_stop():
    assert len(oracle.pyramid_sizes(376, 1241, (21, 21), 3)) == 4
    assert oracle.pyramid_sizes(376, 1241, (21, 21), 3)[-1] == (47, 156)
    assert len(oracle.pyramid_sizes(64, 64, (21, 21), 3)) == 2          # 16 <= 21: stop after level 1
    assert len(oracle.pyramid_sizes(2160, 3840, (31, 31), 4)) == 5
    assert len(oracle.pyramid_sizes(20, 20, (21, 21), 3)) == 1


def test_track_keypoints_twin_semantics():
    a, b = synthetic.sequence(120, 160, 6, 2)
    c = oracle.goodFeaturesToTrack(a, 60, 0.01, 8).reshape(-1, 2)
    t, s, e = oracle.track_keypoints(c.T, a, b, (21, 21), 3)
    assert t.shape == (2, len(c)) and s.shape == (len(c),) and e.shape == (len(c),)
    f, fs, fe_ = oracle.calcOpticalFlowPyrLK(a, b, c, None, (21, 21), 3, (3, 30, 0.01), 0, 1e-3)
    bk, bs, _ = oracle.calcOpticalFlowPyrLK(b, a, f, None, (21, 21), 3, (3, 30, 0.01), 0, 1e-3)
    diff = np.abs(c - bk.reshape(-1, 2)).max(-1)
    assert np.array_equal(s, fs.ravel() & (diff < 1.0))               # back status ignored (continuous.py:158)
    assert np.array_equal(t, f.reshape(-1, 2).T) and np.array_equal(e, fe_.ravel())
    t, s, e = oracle.track_keypoints(np.empty((2, 0)), a, b)
    assert t.shape == (2, 0) and s.dtype == np.uint8 and e.dtype == np.float32


# ------------------------------------------------------------------ oracle vs numpy twin
@pytest.mark.parametrize("h,w,seed", [(57, 83, 1), (96, 130, 5), (40, 41, 9), (7, 9, 2), (1, 12, 3), (12, 1, 3), (2, 2, 1)])
def test_integer_stages_match_twin(h, w, seed):
    a = np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.uint8)
    assert np.array_equal(oracle.pyr_down(a), T.pyr_down(a))
    assert np.array_equal(oracle.scharr(a), T.scharr(a))


@pytest.mark.parametrize("h,w,seed", [(57, 83, 1), (96, 130, 5), (3, 3, 7), (5, 40, 2)])
@pytest.mark.parametrize("harris", [False, True])
def test_score_map_matches_twin(h, w, seed, harris):
    a = synthetic.sequence(h, w, seed, 1)[0]
    assert np.array_equal(oracle.corner_score(a, harris), T.corner_score(a, harris))
    assert np.array_equal(oracle.corner_score(a, harris, nthreads=4), oracle.corner_score(a, harris))


@pytest.mark.parametrize("maxc,q,md", [(50, 0.01, 7), (0, 0.05, 3), (20, 0.01, 0), (0, 0.01, 10.4), (500, 0.001, 1)])
def test_gftt_matches_twin(maxc, q, md):
    a = synthetic.sequence(96, 130, 5, 1)[0]
    assert np.array_equal(oracle.goodFeaturesToTrack(a, maxc, q, md), T.gftt(a, maxc, q, md))
    m = np.zeros_like(a)
    m[10:70, 20:100] = 1
    assert np.array_equal(oracle.goodFeaturesToTrack(a, maxc, q, md, mask=m), T.gftt(a, maxc, q, md, mask=m))


@pytest.mark.parametrize("win,lvl,flags", [((11, 11), 2, 0), ((21, 21), 3, 0), ((9, 15), 1, 0), ((21, 21), 2, 8), ((21, 21), 2, 4)])
def test_lk_matches_twin(win, lvl, flags):
    a, b = synthetic.sequence(96, 130, 5, 2)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 24, 0.01, 7), 30, 96, 130, 5)
    init = (pts + np.float32([0.7, -0.4])).astype(np.float32) if flags & 4 else None
    n0, s0, e0 = oracle.calcOpticalFlowPyrLK(a, b, pts, init, win, lvl, (3, 30, 0.01), flags, 1e-4)
    n1, s1, e1 = T.lk(a, b, pts, win, lvl, 30, 0.01, flags, 1e-4, init)
    assert np.array_equal(n0.reshape(-1, 2), n1)
    assert np.array_equal(s0.ravel(), s1)
    assert np.array_equal(e0.ravel(), e1)


def test_lk_wide_windows_match_twin():
    """Windows beyond 64 columns / rows (cv2 takes any winSize; the GPU runs them on its workgroup-per-keypoint fallback): the
    C++ restatement against the numpy twin, and its AVX2 loops against its scalar loops."""
    a, b = synthetic.sequence(150, 200, 8, 2)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 6, 0.01, 20), 8, 150, 200, 8)
    for win, lvl in (((71, 71), 1), ((33, 67), 0)):
        n0, s0, e0 = oracle.calcOpticalFlowPyrLK(a, b, pts, None, win, lvl, (3, 30, 0.01), 0, 1e-4)
        n1, s1, e1 = T.lk(a, b, pts, win, lvl, 30, 0.01, 0, 1e-4, None)
        assert np.array_equal(n0.reshape(-1, 2), n1) and np.array_equal(s0.ravel(), s1) and np.array_equal(e0.ravel(), e1)
        with oracle.simd():
            n2, s2, e2 = oracle.calcOpticalFlowPyrLK(a, b, pts, None, win, lvl, (3, 30, 0.01), 0, 1e-4)
        assert np.array_equal(n0, n2) and np.array_equal(s0, s2) and np.array_equal(e0, e2)


def _float_lk(prev_img, next_img, pts, win, max_level, max_count=30, eps=0.01, min_eig_thr=1e-4):
    """Pyramidal Lucas-Kanade in plain float64 -- the algorithm of Bouguet's report that OpenCV implements, WITHOUT OpenCV's
    fixed-point steps (14-bit integer bilinear weights, intensities kept with 5 fractional bits, derivatives rounded to
    int16, float32 sums): exact bilinear samples of the same u8 pyramid and the same integer Scharr planes, float64 sums and
    solve, the same level loop, window placement and stopping rules."""
    w, h = win
    Ps, Ns = T.pyramid(prev_img, win, max_level), T.pyramid(next_img, win, max_level)
    I = [T._pad101(p.astype(np.float64), ((h, h), (w, w))) for p in Ps]
    J = [T._pad101(p.astype(np.float64), ((h, h), (w, w))) for p in Ns]
    D = [np.pad(T.scharr(p).astype(np.float64), ((h, h), (w, w), (0, 0))) for p in Ps]
    hx, hy = (w - 1) * 0.5, (h - 1) * 0.5

    def sample(P, fx, fy):
        ix, iy = int(np.floor(fx)), int(np.floor(fy))
        a, b = fx - ix, fy - iy
        q = P[iy + h:iy + h + h + 1, ix + w:ix + w + w + 1]
        return (1 - a) * (1 - b) * q[:-1, :-1] + a * (1 - b) * q[:-1, 1:] + (1 - a) * b * q[1:, :-1] + a * b * q[1:, 1:]

    out, st = np.zeros((len(pts), 2)), np.ones(len(pts), np.uint8)
    top = len(Ps) - 1
    for i, (px, py) in enumerate(np.asarray(pts, np.float64)):
        ox = oy = 0.0
        for L in range(top, -1, -1):
            sc = 1.0 / (1 << L)
            ppx, ppy = px * sc - hx, py * sc - hy
            npx, npy = (px * sc, py * sc) if L == top else (ox * 2, oy * 2)
            ox, oy = npx, npy
            rows, cols = Ps[L].shape
            if not (-w <= np.floor(ppx) < cols and -h <= np.floor(ppy) < rows):
                if L == 0:
                    st[i] = 0
                continue
            Ip, Ix, Iy = sample(I[L], ppx, ppy), sample(D[L][..., 0], ppx, ppy), sample(D[L][..., 1], ppx, ppy)
            # OpenCV's units (its thresholds are stated in them): derivatives in raw Scharr units, differences times 32, sums 2^-20
            A11, A12, A22 = (Ix * Ix).sum() / (1 << 20), (Ix * Iy).sum() / (1 << 20), (Iy * Iy).sum() / (1 << 20)
            det = A11 * A22 - A12 * A12
            me = (A22 + A11 - np.sqrt((A11 - A22) ** 2 + 4 * A12 * A12)) / (2 * w * h)
            if me < min_eig_thr or det < 1.1920928955078125e-07:
                if L == 0:
                    st[i] = 0
                continue
            npx, npy = npx - hx, npy - hy
            pdx = pdy = 0.0
            for j in range(max_count):
                if not (-w <= np.floor(npx) < cols and -h <= np.floor(npy) < rows):
                    if L == 0:
                        st[i] = 0
                    break
                diff = (sample(J[L], npx, npy) - Ip) * 32.0
                b1, b2 = (diff * Ix).sum() / (1 << 20), (diff * Iy).sum() / (1 << 20)
                dx, dy = (A12 * b2 - A22 * b1) / det, (A12 * b1 - A11 * b2) / det
                npx, npy = npx + dx, npy + dy
                ox, oy = npx + hx, npy + hy
                if dx * dx + dy * dy <= eps * eps:
                    break
                if j > 0 and abs(dx + pdx) < 0.01 and abs(dy + pdy) < 0.01:
                    ox, oy = ox - dx * 0.5, oy - dy * 0.5
                    break
                pdx, pdy = dx, dy
        out[i] = (ox, oy)
    return out, st


@pytest.mark.parametrize("h,w,seed,win,lvl", [(240, 320, 5, (21, 21), 3), (200, 260, 9, (11, 11), 2), (240, 320, 6, (31, 31), 2)])
def test_lk_fixed_point_steps_stay_close_to_the_exact_float_algorithm(h, w, seed, win, lvl):
    """What only cv2 could pin is the tracker's fixed-point detail.  What CAN be bounded without cv2 is how far those steps
    (as restated) move a track away from the same algorithm in exact float arithmetic: they are roundings at the 2^-5
    intensity / 2^-14 weight level, so the two must agree to a few thousandths of a pixel wherever the walk is not bistable."""
    a, b = synthetic.sequence(h, w, seed, 2)
    c = oracle.goodFeaturesToTrack(a, 150, 0.01, 7).reshape(-1, 2)
    n0, s0, _ = oracle.calcOpticalFlowPyrLK(a, b, c, None, win, lvl, (3, 30, 0.01), 0, 1e-4)
    n1, s1 = _float_lk(a, b, c, win, lvl)
    s0 = s0.ravel()
    assert np.mean(s0 == s1) >= 0.99
    ok = (s0 == 1) & (s1 == 1)
    assert ok.sum() > 100
    d = np.abs(n0 - n1).max(1)[ok]
    # measured: median 4e-5 px, maximum 1.5e-4 px, identical status flags
    assert np.median(d) < 5e-4 and np.mean(d < 2e-3) >= 0.98 and d.max() < 0.05, (np.median(d), d.max())


def test_oracle_thread_count_invariance():
    a, b = synthetic.sequence(120, 160, 6, 2)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 150, 0.01, 5), 200, 120, 160, 6)
    r1 = oracle.calcOpticalFlowPyrLK(a, b, pts, None, (21, 21), 3, nthreads=1)
    r8 = oracle.calcOpticalFlowPyrLK(a, b, pts, None, (21, 21), 3, nthreads=8)
    assert all(np.array_equal(x, y) for x, y in zip(r1, r8))


@pytest.mark.parametrize("win,lvl,flags", [((5, 5), 2, 0), ((8, 8), 1, 0), ((11, 11), 2, 0), ((21, 21), 3, 0), ((9, 15), 1, 0),
                                           ((31, 17), 3, 0), ((21, 21), 2, 8), ((21, 21), 2, 4), ((33, 33), 0, 0)])
def test_vector_tracker_is_bit_identical_to_the_scalar_restatement(win, lvl, flags):
    """oracle.simd(): the window loops eight pixels at a time (AVX2), for bench.py's timed CPU baseline.  Same integer window
    sums, so the outputs must equal the scalar loops' bit for bit -- ragged window widths, points hanging over every image
    edge, min-eig errors, initial flow, and the multi-threaded pyramid build included."""
    a, b = synthetic.sequence(120, 164, 9, 2)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 150, 0.01, 5), 260, 120, 164, 9)
    edge = np.float32([[-3.5, 10], [0, 0], [163.2, 119.7], [170, 60], [80, -6.2], [81.5, 125.1], [1.25, 118.5], [162.5, 2.5], [400, 300], [-40, 50]])
    pts = np.concatenate([pts.reshape(-1, 2), edge]).astype(np.float32)
    init = (pts + np.float32([0.7, -0.4])).astype(np.float32) if flags & 4 else None
    r0 = oracle.calcOpticalFlowPyrLK(a, b, pts, init, win, lvl, (3, 30, 0.01), flags, 1e-4, nthreads=1)
    with oracle.simd():
        r1 = oracle.calcOpticalFlowPyrLK(a, b, pts, init, win, lvl, (3, 30, 0.01), flags, 1e-4, nthreads=1)
        r8 = oracle.calcOpticalFlowPyrLK(a, b, pts, init, win, lvl, (3, 30, 0.01), flags, 1e-4, nthreads=8)
    for x, y, z in zip(r0, r1, r8):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    assert r0[1].any() and not r0[1].all()


# ------------------------------------------------------------------ golden fixtures
def test_golden_fixtures():
    """Fixtures written by tests/golden/make_golden.py from the oracle (small inputs); they catch
    regressions of the oracle itself.  They are NOT cv2 outputs (cv2 is unavailable)."""
    z = np.load(os.path.join(GOLDEN, "frontend_small.npz"))
    a, b = z["img0"], z["img1"]
    assert np.array_equal(oracle.pyr_down(a), z["pyr1"])
    assert np.array_equal(oracle.scharr(a), z["scharr0"])
    assert np.array_equal(oracle.corner_score(a), z["score"])
    assert np.array_equal(oracle.corner_score(a, True), z["harris"])
    c = oracle.goodFeaturesToTrack(a, 32, 0.01, 5)
    assert np.array_equal(c, z["corners"])
    for bs, ks in ((5, 3), (3, 5), (7, 7), (2, 1)):
        assert np.array_equal(oracle.corner_score(a, blockSize=bs, ksize=ks), z["score_b%d_k%d" % (bs, ks)])
        assert np.array_equal(oracle.corner_score(a, True, blockSize=bs, ksize=ks), z["harris_b%d_k%d" % (bs, ks)])
        assert np.array_equal(oracle.goodFeaturesToTrack(a, 32, 0.01, 5, blockSize=bs, gradientSize=ks), z["corners_b%d_k%d" % (bs, ks)])
    n, s, e = oracle.calcOpticalFlowPyrLK(a, b, z["pts"], None, (11, 11), 2, (3, 30, 0.01), 0, 1e-4)
    assert np.array_equal(n, z["lk_next"]) and np.array_equal(s, z["lk_status"]) and np.array_equal(e, z["lk_err"])
    t, s, e = oracle.track_keypoints(z["pts"].T, a, b, (11, 11), 2)
    assert np.array_equal(t, z["fb_next"]) and np.array_equal(s, z["fb_status"])
    with oracle.simd():
        t, s, e = oracle.track_keypoints(z["pts"].T, a, b, (11, 11), 2, nthreads=4)
    assert np.array_equal(t, z["fb_next"]) and np.array_equal(s, z["fb_status"])


# ------------------------------------------------------------------ candidate bookkeeping
def dedup_cases():
    """Feature / existing point sets incl. distances exactly at, one ulp above and below the threshold."""
    rng = np.random.default_rng(11)
    cases = []
    for nf, ne in ((1, 1), (7, 3), (300, 900), (1000, 2500)):
        f = (rng.random((nf, 2)) * [1241, 376]).astype(np.float32)
        e = (rng.random((ne, 2)) * [1241, 376]).astype(np.float32)
        cases.append((f, e, 15))
    # exact 9-12-15 and 8-15-17 triangles: distance == threshold must be rejected
    f = np.float32([[100, 100], [100, 100], [50, 50], [50, 50]])
    e = np.float32([[109, 112], [400, 300]])
    cases.append((f, e, 15))
    cases.append((np.float32([[0, 0], [8, 15], [3, 4]]), np.float32([[8, 15], [0, 0]]), 17))
    # one float32 ulp around the boundary, and a fractional threshold
    base = np.float32([[200, 200]])
    near = np.float32([[209, np.nextafter(np.float32(212), np.float32(300))], [209, np.nextafter(np.float32(212), np.float32(0))], [209, 212]])
    for row in near:
        cases.append((base, row.reshape(1, 2), 15))
    cases.append((f, e, 14.999999999999998))
    cases.append((f, e, 15.000000000000002))
    cases.append((f, e, 0.0))
    cases.append((f, e, -1.0))
    cases.append((np.float32([[1, 1], [np.nan, 2], [np.inf, 0]]), np.float32([[100, 100], [5, 5]]), 3))
    return cases


def test_new_feature_mask_restatement_is_pinned_by_scipy_cdist():
    """This row is NOT parity-unpinned: scipy is importable here, so the numpy restatement of
    continuous.py:408-413 is checked against scipy.spatial.distance.cdist itself."""
    from scipy.spatial.distance import cdist
    for f, e, thr in dedup_cases():
        with np.errstate(invalid="ignore"):
            ref = np.min(cdist(f, e), axis=1) > thr
            got = oracle.new_feature_mask(f, e, thr)
        assert np.array_equal(got, ref), (f.shape, e.shape, thr)


def candidate_cases():
    """(tracked (2,N) f32, status, Fi (2,N) f32|f64, Ti (D,N), dmin, dmax, dtri) incl. travel distances exactly at and
    within a few ulps of every threshold, NaN/inf coordinates, empty and all-lost sets."""
    rng = np.random.default_rng(23)
    cases = []
    for n, tdim, fdt in ((1, 12, np.float32), (63, 12, np.float64), (1024, 16, np.float32), (1025, 12, np.float64),
                         (5000, 12, np.float32), (3000, 12, np.float64)):
        Fi = (rng.random((2, n)) * [[1241], [376]]).astype(fdt)
        ang = rng.random(n) * 2 * np.pi
        r = np.where(rng.random(n) < 0.5, rng.random(n) * 130, rng.choice([2.0, 15.0, 100.0], n) * (1 + rng.integers(-4, 5, n) * 6e-8))
        Ci = (Fi.astype(np.float64) + r * np.vstack([np.cos(ang), np.sin(ang)])).astype(np.float32)
        st = (rng.random(n) < 0.8).astype(np.uint8)
        Ti = rng.normal(0, 3, (tdim, n))
        cases.append((Ci, st, Fi, Ti, 2, 100, 15))
    # integer Pythagorean displacements: distance == threshold exactly (rejected by the strict comparisons)
    disp = np.array([[0, 2], [2, 0], [60, 80], [9, 12], [12, 9], [3, 4], [0, 0], [28, 96], [0, 100], [1, 2], [80, 60.5]]).T
    for fdt in (np.float32, np.float64):
        Fi = np.full((2, disp.shape[1]), 300.0, fdt)
        Ci = (Fi + disp).astype(np.float32)
        cases.append((Ci, np.ones(disp.shape[1], np.uint8), Fi, np.arange(12.0 * disp.shape[1]).reshape(12, -1), 2, 100, 15))
        # ulp sweeps along an axis around each threshold
        xs = []
        for t in (2.0, 15.0, 100.0, 0.1, 2.5):
            x = np.float32(t)
            lo = x
            for _ in range(4):
                lo = np.nextafter(lo, np.float32(0))
            for _ in range(9):
                xs.append(lo)
                lo = np.nextafter(lo, np.float32(1e9))
        xs = np.float32(xs)
        Fi = np.zeros((2, xs.size), fdt)
        Ci = np.vstack([xs, np.zeros_like(xs)]).astype(np.float32)
        one = np.ones(xs.size, np.uint8)
        Ti = rng.normal(0, 1, (12, xs.size))
        cases.append((Ci, one, Fi, Ti, 2, 100, 15))
        cases.append((Ci, one, Fi, Ti, 0.1, 2.5, 2))
        cases.append((Ci, one, Fi, Ti, 2.5, 100.0, 0.1))
        cases.append((Ci, one, Fi, Ti, -1, np.inf, 0))
        cases.append((Ci, one, Fi, Ti, 0, 1e30, np.nan))
        # NaN / inf coordinates and statuses other than 0/1
        Ci = np.float32([[np.nan, 5, np.inf, 10, 20, 30], [0, np.nan, 0, 10, 20, 30]])
        Fi = np.array([[0, 0, 0, 0, np.inf, 0], [0, 0, 0, 0, 0, 0]], fdt)
        cases.append((Ci, np.uint8([1, 1, 1, 1, 1, 2]), Fi, rng.normal(0, 1, (12, 6)), 2, 100, 15))
        cases.append((np.empty((2, 0), np.float32), np.empty(0, np.uint8), np.empty((2, 0), fdt), np.empty((12, 0)), 2, 100, 15))
        cases.append((Ci, np.zeros(6, np.uint8), Fi, rng.normal(0, 1, (12, 6)), 2, 100, 15))
    return cases


def test_filter_candidates_restatement_is_the_reference_expression():
    """continuous.py:391-405/446-448 are plain numpy, so the oracle for this row IS the reference's arithmetic; this
    checks the bookkeeping around it: shapes, dtypes, order, and a hand-worked example."""
    Ci = np.float32([[10, 13, 70, 19], [10, 14, 90, 22]])
    Fi = np.float32([[10, 10, 10, 10], [10, 10, 10, 10]])
    Ti = np.arange(48.0).reshape(12, 4)
    c, f, t, tri = oracle.filter_candidates(Ci, np.uint8([1, 1, 1, 1]), Fi, Ti)
    # distances 0, 5, 100, 15 -> kept: 5 and 15; triangulation flag: 15 is not > 15
    assert np.array_equal(c, Ci[:, [1, 3]]) and np.array_equal(f, Fi[:, [1, 3]]) and np.array_equal(t, Ti[:, [1, 3]])
    assert tri.tolist() == [False, False] and c.dtype == np.float32 and t.dtype == np.float64
    with np.errstate(invalid="ignore", over="ignore"):
        for Ci, st, Fi, Ti, a, b, d in candidate_cases():
            c, f, t, tri = oracle.filter_candidates(Ci, st, Fi, Ti, a, b, d)
            assert c.shape[0] == 2 and f.shape == c.shape and t.shape == (Ti.shape[0], c.shape[1]) and tri.shape == (c.shape[1],)
            assert f.dtype == Fi.dtype and c.shape[1] <= int((st == 1).sum())


# ------------------------------------------------------------------ third-party pins of the linear stages
# cv2 is absent, so the LK/GFTT restatement as a whole stays "parity unpinned"; the stages below are plain linear
# filters with a published definition, and scipy.ndimage (an independent implementation, in the image) evaluates
# that definition with 'mirror' = REFLECT_101 borders.  The oracle must agree with it.
def _texture(h, w, seed):
    from visual_odometry_amd import synthetic
    return synthetic.sequence(h, w, seed=seed, n_frames=1)[0]


@pytest.mark.parametrize("h,w", [(64, 64), (97, 131), (33, 250), (5, 7), (376, 1241)])
def test_pyr_down_and_scharr_pinned_by_scipy_ndimage(h, w):
    from scipy import ndimage
    img = _texture(h, w, h * 7 + w)
    g = np.array([1, 4, 6, 4, 1], np.int64)
    full = ndimage.correlate(img.astype(np.int64), np.outer(g, g), mode="mirror")
    assert np.array_equal(oracle.pyr_down(img), ((full + 128) >> 8)[::2, ::2].astype(np.uint8))
    sm, df = np.array([3, 10, 3], np.int64), np.array([-1, 0, 1], np.int64)
    ix = ndimage.correlate(img.astype(np.int64), np.outer(sm, df), mode="mirror")
    iy = ndimage.correlate(img.astype(np.int64), np.outer(df, sm), mode="mirror")
    d = oracle.scharr(img)
    assert np.array_equal(d[..., 0], ix) and np.array_equal(d[..., 1], iy)


@pytest.mark.parametrize("harris", [False, True])
def test_score_map_pinned_by_scipy_ndimage(harris):
    """Sobel 3x3 / (4*3*255), products, 3x3 unnormalised box sum, min-eigenvalue or Harris response, evaluated by
    scipy.ndimage in float64: the float32 oracle map agrees to float32 rounding of its intermediate products."""
    from scipy import ndimage
    img = _texture(120, 160, 5)
    a = img.astype(np.float64)
    sm, df = np.array([1, 2, 1.0]), np.array([-1, 0, 1.0])
    dx = ndimage.correlate(a, np.outer(sm, df), mode="mirror") / 3060.0
    dy = ndimage.correlate(a, np.outer(df, sm), mode="mirror") / 3060.0
    box = lambda m: ndimage.correlate(m, np.ones((3, 3)), mode="mirror")
    xx, xy, yy = box(dx * dx), box(dx * dy), box(dy * dy)
    if harris:
        ref = xx * yy - xy * xy - 0.04 * (xx + yy) ** 2
    else:
        ref = (xx * 0.5 + yy * 0.5) - np.sqrt((xx * 0.5 - yy * 0.5) ** 2 + xy * xy)
    got = oracle.corner_score(img, harris=harris).astype(np.float64)
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-6 * scale + 1e-12


# ------------------------------------------------------------------ any blockSize / gradientSize (cv2's optional arguments)
GEOM = [(1, 3), (2, 3), (3, 1), (3, 5), (5, 3), (5, 5), (7, 7), (4, 7), (15, 5), (31, 1)]


def test_sobel_taps_are_opencvs_kernels():
    """cv::getSobelKernels' published kernels (its documentation and the classic Sobel tables)."""
    assert list(T.sobel_taps(1, 0)) == [1] and list(T.sobel_taps(1, 1)) == [-1, 0, 1]
    assert list(T.sobel_taps(3, 0)) == [1, 2, 1] and list(T.sobel_taps(3, 1)) == [-1, 0, 1]
    assert list(T.sobel_taps(5, 0)) == [1, 4, 6, 4, 1] and list(T.sobel_taps(5, 1)) == [-1, -2, 0, 2, 1]
    assert list(T.sobel_taps(7, 0)) == [1, 6, 15, 20, 15, 6, 1] and list(T.sobel_taps(7, 1)) == [-1, -4, -5, 0, 5, 4, 1]


@pytest.mark.parametrize("harris", [False, True])
def test_generic_score_path_reproduces_the_dedicated_one_for_block3_aperture3(harris):
    for h, w, seed in [(57, 83, 1), (3, 3, 7), (5, 40, 2), (1, 9, 4), (120, 160, 5)]:
        a = synthetic.sequence(h, w, seed, 1)[0]
        assert np.array_equal(oracle.corner_score(a, harris, force_generic=True), oracle.corner_score(a, harris))
        assert np.array_equal(T.corner_score_generic(a, 3, 3, harris), T.corner_score(a, harris))


@pytest.mark.parametrize("block,ksize", GEOM)
@pytest.mark.parametrize("harris", [False, True])
def test_generic_score_map_matches_twin(block, ksize, harris):
    """C++ loops vs whole-array numpy, bit for bit, incl. images smaller than the box / the Sobel kernel (REFLECT_101 bounces
    more than once there, as cv::borderInterpolate does)."""
    for h, w, seed in [(57, 83, 1), (40, 41, 9), (3, 3, 7), (5, 40, 2), (1, 1, 0), (2, 7, 3), (9, 2, 3)]:
        a = np.random.default_rng(seed).integers(0, 256, (h, w)).astype(np.uint8) if h * w < 50 else synthetic.sequence(h, w, seed, 1)[0]
        got = oracle.corner_score(a, harris, blockSize=block, ksize=ksize)
        assert np.array_equal(got, T.corner_score_generic(a, block, ksize, harris)), (h, w)
        assert np.array_equal(got, oracle.corner_score(a, harris, blockSize=block, ksize=ksize, nthreads=4))


@pytest.mark.parametrize("block,ksize", [(5, 3), (3, 5), (7, 7), (2, 1), (9, 5)])
@pytest.mark.parametrize("harris", [False, True])
def test_generic_score_map_pinned_by_scipy_ndimage(block, ksize, harris):
    """The published definition (Sobel taps of the aperture, 1 / (2^(aperture-1) * blockSize * 255), blockSize x blockSize
    unnormalised box sum anchored at blockSize / 2) evaluated by scipy.ndimage in float64."""
    from scipy import ndimage
    img = _texture(120, 160, 5)
    a = img.astype(np.float64)
    sm, df = T.sobel_taps(ksize, 0).astype(np.float64), T.sobel_taps(ksize, 1).astype(np.float64)
    scale = 1.0 / ((1 << (ksize - 1)) * block * 255.0)
    dx = ndimage.correlate1d(ndimage.correlate1d(a, df, axis=1, mode="mirror"), sm, axis=0, mode="mirror") * scale
    dy = ndimage.correlate1d(ndimage.correlate1d(a, sm, axis=1, mode="mirror"), df, axis=0, mode="mirror") * scale
    # scipy centres an even-length kernel at n // 2 as well: origin 0 is OpenCV's anchor blockSize / 2
    box = lambda m: ndimage.correlate1d(ndimage.correlate1d(m, np.ones(block), axis=1, mode="mirror"), np.ones(block), axis=0, mode="mirror")
    xx, xy, yy = box(dx * dx), box(dx * dy), box(dy * dy)
    if harris:
        ref = xx * yy - xy * xy - 0.04 * (xx + yy) ** 2
    else:
        ref = (xx * 0.5 + yy * 0.5) - np.sqrt((xx * 0.5 - yy * 0.5) ** 2 + xy * xy)
    got = oracle.corner_score(img, harris=harris, blockSize=block, ksize=ksize).astype(np.float64)
    assert np.abs(got - ref).max() <= 4e-6 * np.abs(ref).max() + 1e-12


@pytest.mark.parametrize("block,ksize", [(5, 3), (7, 5), (2, 1), (3, 7)])
def test_gftt_with_block_and_gradient_size_matches_twin(block, ksize):
    a = synthetic.sequence(96, 130, 5, 1)[0]
    for maxc, q, md in [(50, 0.01, 7), (0, 0.05, 3)]:
        got = oracle.goodFeaturesToTrack(a, maxc, q, md, blockSize=block, gradientSize=ksize)
        assert got is not None and np.array_equal(got, T.gftt(a, maxc, q, md, block=block, ksize=ksize))
    with pytest.raises(ValueError):
        oracle.goodFeaturesToTrack(a, 10, 0.01, 5, blockSize=0)
    with pytest.raises(ValueError):
        oracle.goodFeaturesToTrack(a, 10, 0.01, 5, gradientSize=4)


@pytest.mark.parametrize("h,w,seed,maxc,md", [(376, 1241, 0, 2000, 7), (480, 640, 3, 1000, 10), (600, 800, 1, 1000, 10)])
def test_corner_lists_from_scipys_float64_score_map_equal_the_oracles(h, w, seed, maxc, md):
    """The whole detector on the reference's frame shapes with NOTHING of the float32 restatement in the score map: Sobel / box
    sums / minimum eigenvalue evaluated by scipy.ndimage in float64 from the published definition, the selection rule (threshold,
    3x3 maxima, order, greedy minDistance) by the numpy twin.  The corner LISTS -- sets and order -- must equal the oracle's:
    the float32 operation order, which is what cannot be pinned without cv2, does not decide a single corner on these frames."""
    from scipy import ndimage
    img = _texture(h, w, seed)
    a = img.astype(np.float64)
    sm, df = np.array([1, 2, 1.0]), np.array([-1, 0, 1.0])
    dx = ndimage.correlate(a, np.outer(sm, df), mode="mirror") / 3060.0
    dy = ndimage.correlate(a, np.outer(df, sm), mode="mirror") / 3060.0
    box = lambda m: ndimage.correlate(m, np.ones((3, 3)), mode="mirror")
    xx, xy, yy = box(dx * dx), box(dx * dy), box(dy * dy)
    ref = (xx * 0.5 + yy * 0.5) - np.sqrt((xx * 0.5 - yy * 0.5) ** 2 + xy * xy)
    c64 = T.gftt(img, maxc, 0.01, md, eig=ref)
    c32 = oracle.goodFeaturesToTrack(img, maxc, 0.01, md, nthreads=4)
    assert c64.shape == c32.shape and np.array_equal(c64, c32)


def test_gftt_candidate_rule_pinned_by_scipy_ndimage():
    """goodFeaturesToTrack with minDistance 0 and no corner limit returns every candidate: pixels above
    qualityLevel * max that equal their 3x3 maximum (grey dilation), image border excluded, strongest first with
    ties in reverse raster order.  The candidate set and order are rebuilt with scipy.ndimage from the oracle's own
    score map."""
    from scipy import ndimage
    for img in (_texture(90, 120, 3), np.kron((np.indices((8, 10)).sum(0) % 2).astype(np.uint8) * 200 + 20, np.ones((9, 9), np.uint8))):
        S = oracle.corner_score(img)
        thr = np.float32(np.float64(S.max()) * 0.01)
        T = np.where(S > thr, S, np.float32(0))
        loc = (T == ndimage.grey_dilation(T, size=(3, 3), mode="mirror")) & (T != 0)
        loc[0, :] = loc[-1, :] = False
        loc[:, 0] = loc[:, -1] = False
        ys, xs = np.nonzero(loc)
        order = np.lexsort((-(ys * img.shape[1] + xs), -S[ys, xs].astype(np.float64)))
        ref = np.stack([xs[order], ys[order]], 1).astype(np.float32)
        got = oracle.goodFeaturesToTrack(img, 0, 0.01, 0).reshape(-1, 2)
        assert got.shape == ref.shape and np.array_equal(got, ref)
        assert len(ref) > 50


# ------------------------------------------------------------------ batched triangulation
def triangulation_scene(n, seed, fdt=np.float32, noise=0.3):
    """A two-view scene: world points in front of both cameras, first view at a random earlier pose per candidate,
    pixel noise on both observations, some points behind the camera / with a tiny baseline."""
    rng = np.random.default_rng(seed)
    K = np.array([[718.856, 0, 607.19], [0, 718.856, 185.2], [0, 0, 1.0]])

    def pose_at(tz, yaw, tx):
        c, s = np.cos(yaw), np.sin(yaw)
        T = np.eye(4)
        T[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
        T[:3, 3] = [tx, 0.02 * tz, tz]
        return T                                         # camera-to-world, as the reference stores poses

    cur = pose_at(6.0, 0.05, 0.3)
    X = np.vstack([rng.uniform(-15, 15, n), rng.uniform(-3, 3, n), rng.uniform(10, 60, n)])
    Ci, Fi, Ti = np.empty((2, n), np.float32), np.empty((2, n), fdt), np.empty((12, n))
    for i in range(n):
        first = pose_at(rng.uniform(0, 5.9), rng.uniform(-0.05, 0.05), rng.uniform(-0.5, 0.5))
        if i % 11 == 0:
            first = pose_at(5.999, 0.05, 0.3)            # almost no baseline: fails the angle test
        for T, out in ((first, Fi), (cur, Ci)):
            pc = np.linalg.inv(T) @ np.append(X[:, i], 1.0)
            uv = (K @ pc[:3])[:2] / pc[2] + rng.normal(0, noise, 2)
            out[:, i] = uv
        Ti[:, i] = first[:3, :].reshape(-1)
    if n > 7:
        Ci[:, 7] = Fi[:, 7][::-1] * 0.5                  # inconsistent observation: triangulates behind or far off
    flags = rng.random(n) < 0.85
    return K, cur[:3, :], Ci, Fi, Ti, flags


def test_triangulation_restatement_recovers_known_points():
    """The restated loop (continuous.py:450-485) on a noise-free scene returns the world points themselves; the
    stand-in for cv2.triangulatePoints is checked against an independent least-squares DLT solve."""
    K, pose, Ci, Fi, Ti, flags = triangulation_scene(60, 4, np.float64, noise=0.0)
    Ci64 = Ci.astype(np.float64)
    P2 = K @ oracle._invert_transformation(pose)
    for i in (1, 2, 3, 5, 13, 20, 31, 47):               # consistent observations with a real baseline
        P1 = K @ oracle._invert_transformation(Ti[:, i].reshape(3, 4))
        X4 = oracle._triangulate_points(P1, P2, Fi[:, i], Ci64[:, i])
        Xh = (X4[:3] / X4[3]).ravel()
        for P, uv in ((P1, Fi[:, i]), (P2, Ci64[:, i])):
            pr = P @ np.append(Xh, 1.0)
            assert np.allclose(pr[:2] / pr[2], uv, atol=2e-2)            # Ci was rounded to float32 pixels
    keep, pts = oracle.triangulate_candidates(K, pose, Ci, Fi, Ti, flags)
    assert pts.shape == (3, int(keep.sum())) and keep.sum() > 10 and not keep[::11].any() and not (keep & ~flags).any()
    none, empty = oracle.triangulate_candidates(K, pose, Ci, Fi, Ti, np.zeros(60, bool))
    assert not none.any() and empty.shape == (3, 0)


def test_bootstrap_loop_restatement_follows_known_motion_and_its_rules():
    """oracle.bootstrap_correspondences (initialization.py:41-70): on the synthetic sequence every surviving pair links a
    corner of image 0 to where the known warp puts it in the last image; the moved-far-enough rule and the mask are applied
    after every image (no cv2 here: the loop is pinned by the ground truth and by its own numpy rules, parity unpinned)."""
    h, w, n_img = 200, 320, 5
    frames = synthetic.sequence(h, w, seed=9, n_frames=n_img)
    params = dict(maxCorners=300, qualityLevel=0.01, minDistance=10, winSize=(11, 11), dist_threshold_move=1.5)
    calls = []

    def mask(good_new, good_old):
        calls.append(len(good_new))
        assert good_new.shape == good_old.shape and good_new.dtype == np.float32 and good_new.shape[1:] == (1, 2)
        d = np.sqrt(np.sum((good_new - good_old) ** 2, axis=2)).flatten()
        assert (d >= np.float32(1.5)).all()                   # the rule ran before the mask
        m = np.ones((len(good_new), 1), np.uint8)
        m[::5] = 0
        return m
    good_old, good_new, steps = oracle.bootstrap_correspondences(frames, params, mask)
    corners = oracle.goodFeaturesToTrack(frames[0], 300, 0.01, 10).reshape(-1, 2)
    assert len(calls) == n_img - 1 and len(steps) == n_img - 1 and calls == [len(s[0]) for s in steps]
    assert all(calls[i + 1] <= calls[i] - (calls[i] + 4) // 5 for i in range(len(calls) - 1))
    assert 20 < len(good_old) == len(good_new) < len(corners)
    # good_old are corners of image 0, in their original order
    idx = [int(np.flatnonzero((corners == p).all(axis=1))[0]) for p in good_old.reshape(-1, 2)]
    assert idx == sorted(idx)
    t = n_img - 1
    cx, cy = (w - 1) * 0.5, (h - 1) * 0.5
    z = 1.0 + 0.002 * t
    px = (good_old[:, 0, 0] - cx - 1.5 * t) * z + cx
    py = (good_old[:, 0, 1] - cy - 0.4 * t) * z + cy
    e = np.hypot(good_new[:, 0, 0] - px, good_new[:, 0, 1] - py)
    assert np.mean(e < 0.25) > 0.9, (np.mean(e < 0.25), np.median(e))
