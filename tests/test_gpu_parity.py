"""GPU parity tests: the HIP path (through the C ABI, libvofe.so) against the CPU oracle on the
same seeded inputs.  Bars (BASELINE.json north_star): integer stages bit-exact; tracked positions
within 0.05 px; status flags >= 99.5 % identical; corner sets identical.  Because both sides use
the same canonical operation order the float stages are additionally expected to be bit-exact,
which is asserted where it has been observed to hold."""
import numpy as np
import pytest

import oracle
from visual_odometry_amd import synthetic

pytestmark = pytest.mark.gpu

POS_TOL = 0.05          # px, BASELINE.json north_star
STATUS_MIN = 0.995      # fraction of identical status flags


def _seq(h, w, seed, n=2):
    return synthetic.sequence(h, w, seed=seed, n_frames=n)


# ------------------------------------------------------------------ committed golden data (not the live oracle)
def test_hip_path_against_the_committed_golden_fixture(fe):
    """tests/golden/frontend_small.npz holds inputs AND expected outputs (written once by make_golden.py from the oracle;
    cv2 is absent).  Every other GPU test compares the kernels with the oracle as built today; this one compares
    them with committed data, so kernels and oracle cannot drift together unnoticed on the GPU box.  No oracle call here."""
    import os
    import torch
    from visual_odometry_amd.frontend import Pyramid
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_small.npz"), allow_pickle=False)
    a, b, pts = z["img0"], z["img1"], z["pts"]
    pyr = Pyramid(fe, 1, a.shape[0], a.shape[1], (11, 11), 1, True)
    pyr.build(torch.from_numpy(a).cuda(), 0)
    lvl0, der0 = pyr.download(0, 0, deriv=True)
    assert np.array_equal(lvl0, a) and np.array_equal(der0, z["scharr0"])
    assert np.array_equal(pyr.download(0, 1), z["pyr1"])
    pyr.free()
    assert np.array_equal(fe.corner_score(a), z["score"])
    assert np.array_equal(fe.corner_score(a, True), z["harris"])
    assert np.array_equal(fe.goodFeaturesToTrack(a, 32, 0.01, 5), z["corners"])
    for bs, ks in ((5, 3), (3, 5), (7, 7), (2, 1)):                   # cv2's optional blockSize / gradientSize
        assert np.array_equal(fe.corner_score(a, False, 0.04, bs, ks), z["score_b%d_k%d" % (bs, ks)]), (bs, ks)
        assert np.array_equal(fe.corner_score(a, True, 0.04, bs, ks), z["harris_b%d_k%d" % (bs, ks)]), (bs, ks)
        assert np.array_equal(fe.goodFeaturesToTrack(a, 32, 0.01, 5, blockSize=bs, gradientSize=ks), z["corners_b%d_k%d" % (bs, ks)])
    n, s, e = fe.calcOpticalFlowPyrLK(a, b, pts, None, (11, 11), 2, (3, 30, 0.01), 0, 1e-4)
    assert np.array_equal(s, z["lk_status"])
    ok = s.ravel() == 1
    assert np.abs(n - z["lk_next"]).reshape(-1, 2)[ok].max() <= POS_TOL
    assert np.array_equal(n, z["lk_next"]) and np.array_equal(e[ok], z["lk_err"][ok])
    t, fs, fe_ = fe.track_keypoints(pts.T, a, b, (11, 11), 2)
    assert np.array_equal(fs, z["fb_status"]) and np.array_equal(t, z["fb_next"])
    assert np.array_equal(fe_[fs == 1], z["fb_err"][fs == 1])
    # the canonical tracks of the second fixture (variants_small.npz holds the alternative arithmetics' outputs: the
    # kernels must stay within the north-star bars of every one of them, as the CPU spread test shows for the oracle)
    v = np.load(os.path.join(os.path.dirname(__file__), "golden", "variants_small.npz"), allow_pickle=False)
    t, fs, _ = fe.track_keypoints(v["pts"].T, v["img0"], v["img1"], (21, 21), 2)
    for name in oracle.LK_VARIANTS:
        vs, vt = v["lk_%s_status" % name], v["lk_%s_next" % name]
        assert np.mean(vs == fs) >= STATUS_MIN, name
        both = (vs == 1) & (fs == 1)
        assert np.abs(vt - t)[:, both].max() <= POS_TOL, name
    for name in oracle.SCORE_VARIANTS:
        c = fe.goodFeaturesToTrack(v["img0"], 48, 0.01, 5)
        assert set(map(tuple, c.reshape(-1, 2).tolist())) == set(map(tuple, v["corners_%s" % name].reshape(-1, 2).tolist())), name


# ------------------------------------------------------------------ pyramid + Scharr (bit-exact)
@pytest.mark.parametrize("h,w,win,levels", [(376, 1241, (21, 21), 3), (97, 131, (11, 11), 3),
                                            (64, 64, (21, 21), 3), (480, 640, (11, 11), 3),
                                            (33, 250, (5, 7), 2), (600, 800, (41, 41), 0), (23, 49, (21, 21), 0),
                                            (200, 333, (31, 17), 2)])
@pytest.mark.parametrize("path", ["fused", "levels", "fused-generic", "levels-generic", "fused-separate", "levels-separate", "levels-unstepped",
                                  "levels-untailed", "levels-tail8"])
def test_pyramid_and_scharr_bit_exact(fe, knob, h, w, win, levels, path):
    """Both build paths (one fused launch for small frames, per-level kernels for large ones), each with the
    divergence-free copy / border kernels where the levels are wide enough and with the any-width ones forced; the fused
    launch with the border bands written by the owners of their source pixels (default where every level is large enough
    for one reflection) and with the bands left to the border kernel (VOFE_PYR_BORDERS=separate); likewise the per-level
    pyrDown kernel, which writes the bands of the level it produces unless told to leave them to the border launch; the
    default per-level path is the level pipeline (launch k = pyrDown to level k + a share of the Scharr blocks below it, the
    last two levels from LDS in the pipeline's tail launch), "untailed" the pipeline with a pyrDown launch for every level,
    "tail8" the tail kernel with its smaller tile, "unstepped" the plain sequence of the per-level kernels."""
    import torch
    knob("VOFE_PYRAMID", path.split("-")[0])
    if path.endswith("generic"):
        knob("VOFE_PAD_COPY", "generic")
    if path.endswith("separate"):
        knob("VOFE_PYR_BORDERS", "separate")
    if path.endswith("unstepped"):
        knob("VOFE_PYR_STEP", "0")                 # pyrDown launches with their own border bands, Scharr of every level afterwards
    if path.endswith("untailed"):
        knob("VOFE_PYR_TAIL", "0")
    if path.endswith("tail8"):
        knob("VOFE_PYR_TAIL_TH", "8")
    from visual_odometry_amd.frontend import Pyramid
    imgs = _seq(h, w, seed=h + w, n=2)
    pyr = Pyramid(fe, 2, h, w, win, levels, True)
    pyr.build(torch.from_numpy(np.stack(imgs)).cuda(), 0)
    sizes = oracle.pyramid_sizes(h, w, win, levels)
    assert pyr.levels == len(sizes)
    for slot in range(2):
        for lvl in range(pyr.levels):
            ref = oracle.pyramid_level(imgs[slot], lvl)
            assert pyr.level_size(lvl) == sizes[lvl] == ref.shape
            got, der = pyr.download(slot, lvl, deriv=True)
            assert np.array_equal(got, ref), "level %d differs" % lvl
            assert np.array_equal(der, oracle.scharr(ref)), "Scharr of level %d differs" % lvl
            pad = pyr.download_padded(slot, lvl)
            exp = np.pad(ref, ((win[1], win[1]), (win[0], win[0])), mode="reflect")
            assert np.array_equal(pad, exp), "REFLECT_101 border of level %d differs" % lvl
    pyr.free()


@pytest.mark.parametrize("kernel", ["default", "generic"])
def test_level0_copy_kernels_around_the_width_limit(fe, knob, kernel):
    """The level-0 copy has a divergence-free kernel for frames of at least 16*ceil(win/16)+17 columns and an
    any-width one below that (VOFE_PAD_COPY=generic forces the latter): both, on widths around the limit, around
    multiples of 16, and with windows of 1 to 3 mirrored groups per side."""
    import torch
    from visual_odometry_amd.frontend import Pyramid
    knob("VOFE_PYRAMID", "levels")
    if kernel == "generic":
        knob("VOFE_PAD_COPY", "generic")
    rng = np.random.default_rng(5)
    for win in ((5, 5), (21, 21), (16, 9), (33, 12), (40, 40)):
        for w in (41, 48, 49, 50, 63, 64, 65, 80, 81, 95, 96, 97, 200):
            if w <= win[0]:
                continue
            h = int(rng.integers(win[1] + 1, 70))
            img = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
            pyr = Pyramid(fe, 2, h, w, win, 0, False)
            pyr.build(torch.from_numpy(img).cuda(), 0)
            for slot in range(2):
                pad = pyr.download_padded(slot, 0)
                exp = np.pad(img[slot], ((win[1], win[1]), (win[0], win[0])), mode="reflect")
                assert np.array_equal(pad, exp), (win, w, h, slot)
            pyr.free()


def test_pyramid_large_frame_default_path(fe):
    """Above 2 Mpx per build the per-level kernels are chosen by default (6 levels also exceed the fused plan)."""
    import torch
    from visual_odometry_amd.frontend import Pyramid
    h, w = 1300, 2500
    img = _seq(h, w, seed=2, n=1)[0]
    for levels in (3, 6):
        pyr = Pyramid(fe, 1, h, w, (21, 21), levels, True)
        pyr.build(torch.from_numpy(img).cuda(), 0)
        for lvl in range(pyr.levels):
            ref = oracle.pyramid_level(img, lvl)
            got, der = pyr.download(0, lvl, deriv=True)
            assert np.array_equal(got, ref) and np.array_equal(der, oracle.scharr(ref))
        pyr.free()


@pytest.mark.parametrize("path", ["fused", "levels"])
def test_region_of_interest_views_pitch_and_misaligned_base(fe, knob, path):
    """Callers hand over views: a crop of a larger device image has pitch > width and a base address with no
    alignment at all.  Pyramid, score map, detector and tracker must read exactly the crop."""
    import torch
    from visual_odometry_amd.frontend import Pyramid
    knob("VOFE_PYRAMID", path)
    big = _seq(300, 517, seed=9, n=2)
    d_big = torch.from_numpy(np.stack(big)).cuda()
    for (y0, x0, h, w) in ((5, 3, 240, 401), (1, 1, 128, 130), (17, 64, 200, 333)):
        crops = [np.ascontiguousarray(b[y0:y0 + h, x0:x0 + w]) for b in big]
        view = d_big[:, y0:y0 + h, x0:x0 + w]                     # strides (300*517, 517, 1), odd base offset
        assert not view.is_contiguous()
        pyr = Pyramid(fe, 2, h, w, (21, 21), 3, True)
        pyr.build(view, 0)
        for slot in range(2):
            for lvl in range(pyr.levels):
                ref = oracle.pyramid_level(crops[slot], lvl)
                got, der = pyr.download(slot, lvl, deriv=True)
                assert np.array_equal(got, ref) and np.array_equal(der, oracle.scharr(ref)), (y0, x0, slot, lvl)
                pad = pyr.download_padded(slot, lvl)
                assert np.array_equal(pad, np.pad(ref, ((21, 21), (21, 21)), mode="reflect")), (y0, x0, slot, lvl)
        pyr.free()
        assert np.array_equal(fe.corner_score(view[0]), oracle.corner_score(crops[0]))
        ref = oracle.goodFeaturesToTrack(crops[0], 300, 0.01, 6)
        got = fe.goodFeaturesToTrack(view[0], 300, 0.01, 6)
        assert np.array_equal(got, ref)
        rt, rs, re = oracle.track_keypoints(ref.reshape(-1, 2).T, crops[0], crops[1])
        gt, gs, ge = fe.track_keypoints(ref.reshape(-1, 2).T, view[0], view[1])
        assert np.array_equal(gs, rs) and np.array_equal(gt, rt) and np.array_equal(ge, re)


def test_pyramid_rebuild_other_slot(fe):
    import torch
    from visual_odometry_amd.frontend import Pyramid
    a, b, c = _seq(120, 200, seed=9, n=3)
    pyr = Pyramid(fe, 2, 120, 200, (21, 21), 3, True)
    pyr.build(torch.from_numpy(np.stack([a, b])).cuda(), 0)
    pyr.build(torch.from_numpy(c).cuda(), 1)
    assert np.array_equal(pyr.download(0, 1), oracle.pyramid_level(a, 1))
    assert np.array_equal(pyr.download(1, 2), oracle.pyramid_level(c, 2))
    pyr.free()


# ------------------------------------------------------------------ LK
def _lk_compare(fe, a, b, pts, win, max_level, min_eig=1e-4, flags=0, init=None, criteria=(3, 30, 0.01)):
    n1, s1, e1 = fe.calcOpticalFlowPyrLK(a, b, pts, init, win, max_level, criteria, flags, min_eig)
    n0, s0, e0 = oracle.calcOpticalFlowPyrLK(a, b, pts, init, win, max_level, criteria, flags, min_eig, nthreads=8)
    s0, s1 = s0.ravel(), s1.ravel()
    assert np.mean(s0 == s1) >= STATUS_MIN
    ok = (s0 == 1) & (s1 == 1)
    d = np.abs(n1.reshape(-1, 2) - n0.reshape(-1, 2))
    assert d[ok].max(initial=0) <= POS_TOL
    assert np.allclose(e1.ravel()[ok], e0.ravel()[ok], rtol=1e-5, atol=1e-6)
    exact = np.array_equal(n1, n0) and np.array_equal(s0, s1) and np.array_equal(e1, e0)
    return exact, n1, s1


@pytest.mark.parametrize("h,w,seed,win,lvl,npts", [(376, 1241, 0, (21, 21), 3, 2000), (480, 640, 3, (11, 11), 3, 800),
                                                   (600, 800, 1, (41, 41), 0, 400), (240, 320, 5, (31, 31), 4, 500),
                                                   (200, 300, 6, (15, 9), 2, 300), (200, 300, 7, (7, 25), 2, 300)])
def test_lk_matches_oracle(fe, h, w, seed, win, lvl, npts):
    a, b = _seq(h, w, seed)
    c = oracle.goodFeaturesToTrack(a, npts, 0.01, 7)
    pts = synthetic.pad_points(c, npts, h, w, seed)
    exact, nxt, st = _lk_compare(fe, a, b, pts, win, lvl)
    assert exact, "HIP LK is expected to be bit-identical to the oracle"
    assert st.sum() > 0.5 * npts


def test_lk_21x21_generic_kernel_path(fe, knob):
    """21x21 windows normally run the compile-time specialisation; the run-time-geometry kernel must agree."""
    knob("VOFE_LK", "generic")
    a, b = _seq(240, 320, 12)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 400, 0.01, 7), 400, 240, 320, 12)
    exact, _, st = _lk_compare(fe, a, b, pts, (21, 21), 3)
    assert exact and st.sum() > 200
    t1 = fe.track_keypoints(pts.T, a, b)
    t0 = oracle.track_keypoints(pts.T, a, b, nthreads=8)
    assert all(np.array_equal(x, y) for x, y in zip(t1, t0))


@pytest.mark.parametrize("win", [11, 31])
@pytest.mark.parametrize("kernel", ["default", "x1", "lp", "generic"])
def test_lk_ragged_compile_time_windows(fe, knob, win, kernel):
    """11x11 (the reference's Parking window) and 31x31 run compile-time-geometry kernels whose last row group is
    ragged (3+3+3+2 and 16+15 rows); the run-time-geometry kernel (VOFE_LK=generic) must agree with them and the
    oracle on tracking, flags, wild points, flat frames and extreme contrast."""
    if kernel != "default":
        knob("VOFE_LK", kernel)       # (default: the size rule -- level-parallel for 11x11 at this point count; "lp" has no 31x31 form)
    a, b = _seq(240, 320, 21)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 401, 0.01, 7), 401, 240, 320, 21)
    for lvl in (3, 0):
        exact, _, st = _lk_compare(fe, a, b, pts, (win, win), lvl)
        assert exact and st.sum() > 200
    init = (pts + np.float32([1.0, 0.5])).astype(np.float32)
    assert _lk_compare(fe, a, b, pts, (win, win), 2, flags=4, init=init)[0]
    assert _lk_compare(fe, a, b, pts, (win, win), 2, flags=8, min_eig=1e-3)[0]
    t1 = fe.track_keypoints(pts.T, a, b, winSize=(win, win))
    t0 = oracle.track_keypoints(pts.T, a, b, winSize=(win, win), nthreads=8)
    assert all(np.array_equal(x, y) for x, y in zip(t1, t0))
    wild = np.float32([[-30, 5], [5, -30], [400, 50], [50, 400], [-5.5, -5.5], [319.9, 239.9], [0, 0],
                       [80, 60], [np.nan, 3], [1e12, 1e12], [-1e20, 4], [np.inf, 0], [100, 100]])
    flat = np.full_like(a, 77)
    for (x, y) in ((a, b), (flat, flat), (a, flat)):
        n1, s1, e1 = fe.calcOpticalFlowPyrLK(x, y, wild, None, (win, win), 2)
        n0, s0, e0 = oracle.calcOpticalFlowPyrLK(x, y, wild, None, (win, win), 2)
        assert np.array_equal(s0, s1) and np.array_equal(n0, n1, equal_nan=True) and np.array_equal(e0, e1, equal_nan=True)
    rng = np.random.default_rng(18)
    hc = np.kron((rng.random((60, 80)) > 0.5).astype(np.uint8) * 255, np.ones((4, 4), np.uint8))
    hd = np.roll(hc, (1, 2), axis=(0, 1))
    p2 = (rng.random((301, 2)) * [319, 239]).astype(np.float32)
    n0, s0, e0 = oracle.calcOpticalFlowPyrLK(hc, hd, p2, None, (win, win), 2, (3, 30, 0.01), 0, 1e-4, nthreads=8)
    n1, s1, e1 = fe.calcOpticalFlowPyrLK(hc, hd, p2, None, (win, win), 2, (3, 30, 0.01), 0, 1e-4)
    assert np.array_equal(s0, s1) and np.array_equal(n0, n1) and np.array_equal(e0, e1)


@pytest.mark.parametrize("win", [(21, 21), (11, 11), (9, 15), (31, 17), (5, 5)])
def test_lk_wide_window_fallback_forced_on_small_windows(fe, knob, win):
    """VOFE_LK=wide sends every launch through the workgroup-per-keypoint fallback (lk_wide.hip): on windows the
    wavefront kernels also handle it must return the oracle's results bit for bit -- tracking, flags, the FB twin, a ragged
    batch of sequences, wild points, flat frames, extreme contrast."""
    import torch
    from visual_odometry_amd import _lib
    from visual_odometry_amd.frontend import Pyramid
    knob("VOFE_LK", "wide")
    a, b = _seq(240, 320, 21)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 301, 0.01, 7), 301, 240, 320, 21)
    for lvl in (3, 0):
        exact, _, st = _lk_compare(fe, a, b, pts, win, lvl)
        assert exact and st.sum() > 150
    init = (pts + np.float32([1.0, 0.5])).astype(np.float32)
    assert _lk_compare(fe, a, b, pts, win, 2, flags=4, init=init)[0]
    assert _lk_compare(fe, a, b, pts, win, 2, flags=8, min_eig=1e-3)[0]
    assert _lk_compare(fe, a, b, pts, win, 2, criteria=(1, 5, 0.01))[0]
    assert _lk_compare(fe, a, b, pts, win, 2, criteria=(2, 30, 0.5))[0]
    t1 = fe.track_keypoints(pts.T, a, b, winSize=win)
    t0 = oracle.track_keypoints(pts.T, a, b, winSize=win, nthreads=8)
    assert all(np.array_equal(x, y) for x, y in zip(t1, t0))
    wild = np.float32([[-30, 5], [5, -30], [400, 50], [50, 400], [-5.5, -5.5], [319.9, 239.9], [0, 0],
                       [80, 60], [np.nan, 3], [1e12, 1e12], [-1e20, 4], [np.inf, 0], [100, 100]])
    flat = np.full_like(a, 77)
    for (x, y) in ((a, b), (flat, flat), (a, flat)):
        n1, s1, e1 = fe.calcOpticalFlowPyrLK(x, y, wild, None, win, 2)
        n0, s0, e0 = oracle.calcOpticalFlowPyrLK(x, y, wild, None, win, 2)
        assert np.array_equal(s0, s1) and np.array_equal(n0, n1, equal_nan=True) and np.array_equal(e0, e1, equal_nan=True)
    rng = np.random.default_rng(18)
    hc = np.kron((rng.random((60, 80)) > 0.5).astype(np.uint8) * 255, np.ones((4, 4), np.uint8))
    hd = np.roll(hc, (1, 2), axis=(0, 1))
    p2 = (rng.random((101, 2)) * [319, 239]).astype(np.float32)
    n0, s0, e0 = oracle.calcOpticalFlowPyrLK(hc, hd, p2, None, win, 2, (3, 30, 0.01), 0, 1e-4, nthreads=8)
    n1, s1, e1 = fe.calcOpticalFlowPyrLK(hc, hd, p2, None, win, 2, (3, 30, 0.01), 0, 1e-4)
    assert np.array_equal(s0, s1) and np.array_equal(n0, n1) and np.array_equal(e0, e1)
    # a ragged batch: three sequences in one launch through the image index, one index out of range
    seqs = [_seq(120, 160, 40 + k) for k in range(3)]
    imgs = torch.from_numpy(np.stack([q[0] for q in seqs] + [q[1] for q in seqs])).cuda()
    pyr = Pyramid(fe, 6, 120, 160, win, 2, True)
    pyr.build(imgs, 0)
    ps = [synthetic.pad_points(oracle.goodFeaturesToTrack(q[0], 40 + 7 * k, 0.01, 5), 40 + 7 * k, 120, 160, k) for k, q in enumerate(seqs)]
    allp = np.concatenate(ps)
    idx = np.concatenate([np.full(len(q), k, np.int32) for k, q in enumerate(ps)])
    idx[5] = 7                                                   # outside the batch: the point is reported lost
    d_p, d_i = torch.from_numpy(allp).cuda(), torch.from_numpy(idx).cuda()
    n = len(allp)
    nxt = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    st = torch.empty(n, dtype=torch.uint8, device="cuda")
    er = torch.empty(n, dtype=torch.float32, device="cuda")
    _lib.check(fe.L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, 3, d_p.data_ptr(), d_i.data_ptr(), 3, nxt.data_ptr(), st.data_ptr(),
                                  er.data_ptr(), n, 3, 30, 0.01, 1e-3, 1.0))
    fe.synchronize()
    nxt, st, er = nxt.cpu().numpy(), st.cpu().numpy(), er.cpu().numpy()
    o = 0
    for k, q in enumerate(seqs):
        t0, s0, e0 = oracle.track_keypoints(ps[k].T, q[0], q[1], win, 2)
        sl = slice(o, o + len(ps[k]))
        keep = idx[sl] == k
        assert np.array_equal(nxt[sl][keep], t0.T[keep]) and np.array_equal(st[sl][keep], s0[keep])
        assert np.array_equal(er[sl][keep & (s0 == 1)], e0[keep & (s0 == 1)])
        o += len(ps[k])
    assert st[5] == 0 and er[5] == 0
    pyr.free()


@pytest.mark.parametrize("win,lvl", [((65, 65), 2), ((101, 101), 1), ((71, 33), 2), ((33, 71), 1), ((3, 200), 0), ((104, 104), 0)])
def test_lk_windows_beyond_the_wavefront_kernels(fe, win, lvl):
    """cv2.calcOpticalFlowPyrLK takes any winSize; windows wider than 64 columns (or taller than 64 rows per lane group) run the
    workgroup-per-keypoint fallback.  Against the oracle, bit for bit, incl. the FB twin and points whose window hangs over
    the frame's edge."""
    a, b = _seq(300, 400, 33)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 120, 0.01, 9), 128, 300, 400, 33)
    exact, _, st = _lk_compare(fe, a, b, pts, win, lvl)
    assert exact and st.sum() > 60
    assert _lk_compare(fe, a, b, pts, win, lvl, flags=8, min_eig=1e-3)[0]
    t1 = fe.track_keypoints(pts.T, a, b, winSize=win, maxLevel=lvl)
    t0 = oracle.track_keypoints(pts.T, a, b, winSize=win, maxLevel=lvl, nthreads=8)
    assert all(np.array_equal(x, y) for x, y in zip(t1, t0))
    edge = np.float32([[0, 0], [399, 299], [-40, 20], [20, -40], [430, 150], [200, 340], [1.5, 298.5], [np.nan, 3], [1e12, 5]])
    n1, s1, e1 = fe.calcOpticalFlowPyrLK(a, b, edge, None, win, lvl)
    n0, s0, e0 = oracle.calcOpticalFlowPyrLK(a, b, edge, None, win, lvl)
    assert np.array_equal(s0, s1) and np.array_equal(n0, n1, equal_nan=True) and np.array_equal(e0, e1, equal_nan=True)


def test_lk_window_too_large_for_the_fallback_is_refused(fe):
    from visual_odometry_amd import error
    a, b = _seq(300, 400, 33)
    with pytest.raises(error):
        fe.calcOpticalFlowPyrLK(a, b, np.float32([[100, 100]]), None, (121, 121), 0)


@pytest.mark.parametrize("ppw", ["x1", "lp", "x2", "x2p", "x4"])
def test_lk_21x21_kernel_families(fe, knob, ppw):
    """The 21x21 kernel families, each forced on the same small inputs: one keypoint per wavefront (x1), one wavefront
    per pyramid level (lp: what small launches take by the size rule), several keypoints per wavefront (x2 / x4: what
    large launches take): counts not divisible by the packing, every level count, wild points, flags, extreme
    contrast, FB twin."""
    knob("VOFE_LK", ppw)
    a, b = _seq(240, 320, 12)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 401, 0.01, 7), 401, 240, 320, 12)
    for lvl in (3, 2, 1, 0):
        exact, _, st = _lk_compare(fe, a, b, pts, (21, 21), lvl)
        assert exact and st.sum() > 200
    init = (pts + np.float32([1.0, 0.5])).astype(np.float32)
    assert _lk_compare(fe, a, b, pts, (21, 21), 3, flags=4, init=init)[0]
    assert _lk_compare(fe, a, b, pts, (21, 21), 3, flags=8, min_eig=1e-3)[0]
    assert _lk_compare(fe, a, b, pts[:1], (21, 21), 2)[0]
    t1 = fe.track_keypoints(pts.T, a, b)
    t0 = oracle.track_keypoints(pts.T, a, b, nthreads=8)
    assert all(np.array_equal(x, y) for x, y in zip(t1, t0))
    wild = np.float32([[-30, 5], [5, -30], [400, 50], [50, 400], [-11.5, -11.5], [319.9, 239.9], [0, 0],
                       [80, 60], [np.nan, 3], [1e12, 1e12], [-1e20, 4], [np.inf, 0], [100, 100]])
    flat = np.full_like(a, 77)
    for (x, y) in ((a, b), (flat, flat), (a, flat)):
        n1, s1, e1 = fe.calcOpticalFlowPyrLK(x, y, wild, None, (21, 21), 2)
        n0, s0, e0 = oracle.calcOpticalFlowPyrLK(x, y, wild, None, (21, 21), 2)
        assert np.array_equal(s0, s1) and np.array_equal(n0, n1, equal_nan=True) and np.array_equal(e0, e1, equal_nan=True)
    rng = np.random.default_rng(17)
    hc = np.kron((rng.random((60, 80)) > 0.5).astype(np.uint8) * 255, np.ones((4, 4), np.uint8))
    hd = np.roll(hc, (1, 2), axis=(0, 1))
    p2 = (rng.random((301, 2)) * [319, 239]).astype(np.float32)
    n0, s0, e0 = oracle.calcOpticalFlowPyrLK(hc, hd, p2, None, (21, 21), 2, (3, 30, 0.01), 0, 1e-4, nthreads=8)
    n1, s1, e1 = fe.calcOpticalFlowPyrLK(hc, hd, p2, None, (21, 21), 2, (3, 30, 0.01), 0, 1e-4)
    assert np.array_equal(s0, s1) and np.array_equal(n0, n1) and np.array_equal(e0, e1)


def test_lk_flags_initial_flow_and_min_eig(fe):
    a, b = _seq(240, 320, 12)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 300, 0.01, 7), 300, 240, 320, 12)
    init = (pts + np.float32([1.0, 0.5])).astype(np.float32)
    exact, _, _ = _lk_compare(fe, a, b, pts, (21, 21), 3, flags=4, init=init)
    assert exact
    exact, _, _ = _lk_compare(fe, a, b, pts, (21, 21), 3, flags=8, min_eig=1e-3)
    assert exact
    exact, _, _ = _lk_compare(fe, a, b, pts, (21, 21), 3, criteria=(1, 5, 0.01))
    assert exact
    exact, _, _ = _lk_compare(fe, a, b, pts, (21, 21), 3, criteria=(2, 30, 0.3))
    assert exact


def test_lk_out_of_bounds_flat_and_wild_points(fe):
    a, b = _seq(120, 160, 21)
    flat = np.full((120, 160), 77, np.uint8)
    pts = np.float32([[-30, 5], [5, -30], [400, 50], [50, 400], [-11.5, -11.5], [159.9, 119.9], [0, 0],
                      [80, 60], [np.nan, 3], [1e12, 1e12], [-1e20, 4], [np.inf, 0]])
    for (x, y) in ((a, b), (flat, flat), (a, flat)):
        n1, s1, e1 = fe.calcOpticalFlowPyrLK(x, y, pts, None, (21, 21), 2)
        n0, s0, e0 = oracle.calcOpticalFlowPyrLK(x, y, pts, None, (21, 21), 2)
        assert np.array_equal(s0, s1)
        assert np.array_equal(n0, n1, equal_nan=True)
        assert np.array_equal(e0, e1, equal_nan=True)
    n1, s1, _ = fe.calcOpticalFlowPyrLK(flat, flat, pts[6:8], None, (21, 21), 2)
    assert s1.sum() == 0                     # flat patch fails the minEig test


def test_lk_negative_fourth_bilinear_weight(fe):
    """16384 - w00 - w01 - w10 == -1 for ~1e-5 of sub-pixel positions: must be handled as signed."""
    from test_oracle_cpu import negative_weight_points
    a, b = _seq(120, 160, 3)
    pts = np.concatenate([negative_weight_points(8, 40.0, 40.0), negative_weight_points(8, 90.0, 70.0)])
    for lvl in (0, 2):
        exact, _, st = _lk_compare(fe, a, b, pts, (21, 21), lvl)
        assert exact and st.sum() > 0
    exact, _, _ = _lk_compare(fe, b, a, pts, (21, 21), 0)
    assert exact


def test_lk_extreme_contrast_takes_the_wide_sum_path(fe):
    """0/255 block noise: per-lane partial sums exceed 2^25, so the window sums leave the int32 fast
    path and must still equal the oracle's exact int64 sums (large windows make it certain)."""
    rng = np.random.default_rng(17)
    blocks = (rng.random((60, 80)) > 0.5).astype(np.uint8) * 255
    a = np.kron(blocks, np.ones((4, 4), np.uint8))
    b = np.roll(a, (1, 2), axis=(0, 1))
    b[::7, ::5] ^= 255
    pts = (rng.random((300, 2)) * [319, 239]).astype(np.float32)
    for win, lvl in (((21, 21), 2), ((41, 41), 1), ((64, 64), 0)):
        n0, s0, e0 = oracle.calcOpticalFlowPyrLK(a, b, pts, None, win, lvl, (3, 30, 0.01), 0, 1e-4, nthreads=8)
        n1, s1, e1 = fe.calcOpticalFlowPyrLK(a, b, pts, None, win, lvl, (3, 30, 0.01), 0, 1e-4)
        assert np.array_equal(s0, s1) and np.array_equal(n0, n1) and np.array_equal(e0, e1)
        assert s1.sum() > 0


def test_lk_counters(fe):
    import torch
    from visual_odometry_amd import _lib
    a, b = _seq(120, 160, 3)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 100, 0.01, 7), 100, 120, 160, 3)
    cnt = torch.zeros(3, dtype=torch.int64, device="cuda")
    _lib.check(fe.L.vofe_lk_set_counters(fe.ctx, cnt.data_ptr()))
    fe.calcOpticalFlowPyrLK(a, b, pts, None, (21, 21), 2)
    _lib.check(fe.L.vofe_lk_set_counters(fe.ctx, None))
    fe.calcOpticalFlowPyrLK(a, b, pts, None, (21, 21), 2)
    c = cnt.cpu().numpy()
    assert c[0] == 100 and 100 <= c[1] <= 300 and c[1] <= c[2] <= 30 * c[1]


def test_lk_recovers_known_flow(fe):
    h, w = 376, 1241
    f = synthetic.sequence(h, w, seed=0, n_frames=2, noise_sigma=0.0)
    c = fe.goodFeaturesToTrack(f[0], 500, 0.01, 10).reshape(-1, 2)
    inner = (c[:, 0] > 40) & (c[:, 0] < w - 40) & (c[:, 1] > 40) & (c[:, 1] < h - 40)
    nxt, st, _ = fe.calcOpticalFlowPyrLK(f[0], f[1], c[inner], None, (21, 21), 3)
    truth = synthetic.true_flow(c[inner], h, w, 0, 1)
    ok = st.ravel() == 1
    assert ok.mean() > 0.95
    assert np.median(np.abs(nxt.reshape(-1, 2)[ok] - truth[ok])) < 0.05


def test_track_keypoints_twin(fe):
    h, w = 376, 1241
    a, b = _seq(h, w, 0)
    pts = synthetic.pad_points(oracle.goodFeaturesToTrack(a, 2000, 0.01, 7), 2000, h, w, 0)
    t1, s1, e1 = fe.track_keypoints(pts.T, a, b, (21, 21), 3, (3, 30, 0.01), 1e-3)
    t0, s0, e0 = oracle.track_keypoints(pts.T, a, b, (21, 21), 3, (3, 30, 0.01), 1e-3, nthreads=8)
    assert t1.shape == (2, 2000) and s1.shape == (2000,) and e1.shape == (2000,)
    assert t1.dtype == np.float32 and s1.dtype == np.uint8 and e1.dtype == np.float32
    assert np.mean(s0 == s1) >= STATUS_MIN
    ok = (s0 == 1) & (s1 == 1)
    assert np.abs(t1 - t0)[:, ok].max() <= POS_TOL
    assert np.array_equal(t1, t0) and np.array_equal(s1, s0) and np.array_equal(e1, e0)
    # empty input (reference continuous.py:131-132)
    t, s, e = fe.track_keypoints(np.empty((2, 0)), a, b)
    assert t.shape == (2, 0) and s.shape == (0,) and s.dtype == np.uint8 and e.dtype == np.float32


def test_lk_bad_arguments_raise(fe):
    from visual_odometry_amd import error
    a, b = _seq(64, 64, 1)
    pts = np.float32([[10, 10]])
    with pytest.raises(error):
        fe.calcOpticalFlowPyrLK(a, b, pts, None, (2, 2), 3)
    with pytest.raises(error):
        fe.calcOpticalFlowPyrLK(a, b, pts, None, (21, 21), -1)
    with pytest.raises(error):
        fe.calcOpticalFlowPyrLK(a, b[:32], pts, None, (21, 21), 1)
    with pytest.raises(error):
        fe.calcOpticalFlowPyrLK(a, b, pts.astype(np.float64), None, (21, 21), 1)
    assert fe.calcOpticalFlowPyrLK(a, b, np.zeros((0, 1, 2), np.float32)) == (None, None, None)


# ------------------------------------------------------------------ corners
@pytest.mark.parametrize("h,w,seed", [(376, 1241, 0), (600, 800, 1), (97, 131, 4), (3, 3, 2), (1, 50, 2), (40, 2, 2)])
@pytest.mark.parametrize("harris", [False, True])
def test_corner_score_bit_exact(fe, h, w, seed, harris):
    a = _seq(h, w, seed, 1)[0]
    got = fe.corner_score(a, harris, 0.04)
    ref = oracle.corner_score(a, harris, 0.04, nthreads=8)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("path", ["strip", "tile"])
def test_corner_score_kernel_paths_and_edges(fe, knob, path):
    """Register-strip kernel (w >= 8, h >= 4) and LDS-tile kernel on widths/heights around every
    internal boundary: 4-column threads, 256-column waves, 4/8-row strips, last-column position."""
    knob("VOFE_CORNER_SCORE", path)
    rng = np.random.default_rng(3)
    for (h, w) in ((2, 8), (3, 9), (4, 8), (4, 13), (5, 10), (6, 300), (7, 11), (9, 12), (17, 255), (16, 256), (33, 257), (31, 258), (4, 259),
                   (40, 515), (130, 64), (129, 1030)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        for harris in (False, True):
            assert np.array_equal(fe.corner_score(a, harris, 0.04), oracle.corner_score(a, harris, 0.04)), (h, w, harris)
    for (h, w) in ((1100, 1300), (2170, 3000)):            # enough waves for the 8-row / 16-row strips
        big = _seq(h, w, 5, 1)[0]
        assert np.array_equal(fe.corner_score(big, False, 0.04), oracle.corner_score(big, False, 0.04, nthreads=8))
    a = _seq(129, 1030, 6, 1)[0]
    for (r0, r1, c0, c1) in ((10, 100, 500, 1029), (0, 129, 1020, 1030), (0, 40, 0, 300)):
        mask = np.zeros((129, 1030), np.uint8); mask[r0:r1, c0:c1] = 1
        assert np.array_equal(fe.goodFeaturesToTrack(a, 100, 0.01, 5, mask=mask),
                              oracle.goodFeaturesToTrack(a, 100, 0.01, 5, mask=mask))


SCORE_GEOM = [(1, 3), (2, 3), (3, 1), (3, 5), (5, 3), (7, 3), (5, 5), (7, 7), (4, 7), (15, 5), (31, 1), (31, 7), (16, 3)]


@pytest.mark.parametrize("block,ksize", SCORE_GEOM)
def test_corner_score_any_block_size_and_aperture_bit_exact(fe, block, ksize):
    """cv2.goodFeaturesToTrack's optional blockSize / gradientSize (cornerMinEigenVal / cornerHarris with any block and a
    Sobel aperture of 1, 3, 5 or 7): the run-time-geometry kernel against the oracle's generic restatement, incl. frames
    smaller than the box or the Sobel kernel (REFLECT_101 bounces more than once), one-pixel frames, tile edges."""
    rng = np.random.default_rng(block * 10 + ksize)
    for (h, w) in ((97, 131), (1, 1), (1, 50), (40, 2), (3, 3), (16, 64), (17, 65), (15, 33), (33, 16), (20, 17), (200, 333)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8) if h * w < 3000 else _seq(h, w, h + w, 1)[0]
        for harris in (False, True):
            got = fe.corner_score(a, harris, 0.04, block, ksize)
            ref = oracle.corner_score(a, harris, 0.04, blockSize=block, ksize=ksize)
            assert np.array_equal(got, ref), (h, w, harris)
    big = _seq(376, 1241, 0, 1)[0]
    assert np.array_equal(fe.corner_score(big, False, 0.04, block, ksize),
                          oracle.corner_score(big, False, 0.04, nthreads=8, blockSize=block, ksize=ksize))


def test_generic_score_kernel_forced_for_the_default_geometry(fe, knob):
    """VOFE_CORNER_SCORE=generic sends block 3 / aperture 3 through the run-time-geometry kernel: same maps and corners as the
    dedicated kernels and the dedicated restatement, bit for bit."""
    knob("VOFE_CORNER_SCORE", "generic")
    rng = np.random.default_rng(5)
    for (h, w) in ((2, 8), (3, 9), (17, 255), (33, 257), (129, 1030), (376, 1241)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8) if h < 100 else _seq(h, w, 6, 1)[0]
        for harris in (False, True):
            assert np.array_equal(fe.corner_score(a, harris, 0.04), oracle.corner_score(a, harris, 0.04, nthreads=8)), (h, w, harris)
    a = _seq(376, 1241, 0, 1)[0]
    assert np.array_equal(fe.goodFeaturesToTrack(a, 1000, 0.01, 10), oracle.goodFeaturesToTrack(a, 1000, 0.01, 10, nthreads=8))
    # the geometries with unrolled instances (5/3, 7/3, 3/5, 5/5) take the run-time instance under the switch: same maps
    b = _seq(97, 131, 3, 1)[0]
    for bs, ks in ((5, 3), (7, 3), (3, 5), (5, 5)):
        for harris in (False, True):
            assert np.array_equal(fe.corner_score(b, harris, 0.04, bs, ks), oracle.corner_score(b, harris, 0.04, blockSize=bs, ksize=ks)), (bs, ks)


@pytest.mark.parametrize("block,ksize", [(5, 3), (7, 3), (3, 5), (7, 7), (2, 1), (11, 5)])
def test_gftt_with_block_size_and_gradient_size(fe, block, ksize):
    """The detector with cv2's optional arguments (OpenCV's own LK example passes blockSize=7): identical corner lists, also
    with a mask, Harris, through the batched and the no-allocation entry points."""
    import ctypes as C
    import torch
    from visual_odometry_amd import _lib
    for (h, w, seed, maxc, q, md) in ((376, 1241, 0, 1000, 0.01, 10), (480, 640, 3, 500, 0.01, 7), (200, 300, 8, 0, 0.05, 5.5)):
        a = _seq(h, w, seed, 1)[0]
        ref = oracle.goodFeaturesToTrack(a, maxc, q, md, blockSize=block, gradientSize=ksize, nthreads=8)
        got = fe.goodFeaturesToTrack(a, maxc, q, md, blockSize=block, gradientSize=ksize)
        assert ref is not None and np.array_equal(got, ref), (h, w)
    a = _seq(200, 300, 8, 1)[0]
    mask = np.zeros_like(a); mask[50:150, 100:250] = 255
    for harris in (False, True):
        ref = oracle.goodFeaturesToTrack(a, 200, 0.01, 8, mask=mask, blockSize=block, useHarrisDetector=harris, gradientSize=ksize)
        got = fe.goodFeaturesToTrack(a, 200, 0.01, 8, mask=mask, blockSize=block, useHarrisDetector=harris, gradientSize=ksize)
        assert np.array_equal(got, ref), harris
    # raw C ABI: batched and no-allocation entry points with the same geometry
    L = fe.L
    frames = np.stack(_seq(120, 160, 21, 3))
    d = torch.from_numpy(frames).cuda()
    out = torch.empty((3, 100, 2), dtype=torch.float32, device="cuda")
    cnt = (C.c_int * 3)()
    _lib.check(L.vofe_gftt_batch(fe.ctx, d.data_ptr(), 3, 120, 160, 160, 120 * 160, 100, 0.01, 6.0, block, ksize, 0, 0.04,
                                 out.data_ptr(), 100, cnt))
    fe.synchronize()
    for i in range(3):
        ref = oracle.goodFeaturesToTrack(frames[i], 100, 0.01, 6.0, blockSize=block, gradientSize=ksize)
        assert cnt[i] == ref.shape[0] and np.array_equal(out[i, :cnt[i]].cpu().numpy(), ref.reshape(-1, 2)), i
    one = C.c_int(0)
    o1 = torch.empty((100, 2), dtype=torch.float32, device="cuda")
    _lib.check(L.vofe_gftt_noalloc(fe.ctx, d.data_ptr(), 120, 160, 160, 100, 0.01, 6.0, None, block, ksize, 0, 0.04,
                                   o1.data_ptr(), 100, C.byref(one)))
    ref = oracle.goodFeaturesToTrack(frames[0], 100, 0.01, 6.0, blockSize=block, gradientSize=ksize)
    assert one.value == ref.shape[0] and np.array_equal(o1[:one.value].cpu().numpy(), ref.reshape(-1, 2))


def test_next_points_take_the_layout_of_prev_points_as_in_cv2(fe):
    """cv2 creates nextPts with prevPts' own layout: (N,1,2) in -> (N,1,2) out (what the reference passes,
    continuous.py:134-135), (N,2) in -> (N,2) out, (1,N,2) in -> (1,N,2) out; status and err are (N,1) always."""
    a, b = _seq(64, 80, 2)
    c = fe.goodFeaturesToTrack(a, 20, 0.01, 5)
    n = c.shape[0]
    ref = fe.calcOpticalFlowPyrLK(a, b, c, None, (11, 11), 1)
    assert ref[0].shape == (n, 1, 2) and ref[1].shape == (n, 1) and ref[2].shape == (n, 1)
    for shape in ((n, 2), (1, n, 2)):
        got = fe.calcOpticalFlowPyrLK(a, b, c.reshape(shape), None, (11, 11), 1)
        assert got[0].shape == shape and got[1].shape == (n, 1) and got[2].shape == (n, 1)
        assert np.array_equal(got[0].reshape(-1, 2), ref[0].reshape(-1, 2)) and np.array_equal(got[1], ref[1])
        assert oracle.calcOpticalFlowPyrLK(a, b, c.reshape(shape), None, (11, 11), 1)[0].shape == shape


def test_cv2_output_placeholder_keywords_are_accepted(fe):
    """cv2's bindings take optional output placeholders (corners=, status=, err=); callers that name them keep working."""
    a, b = _seq(64, 80, 2)
    c = fe.goodFeaturesToTrack(a, maxCorners=20, qualityLevel=0.01, minDistance=5, corners=None, mask=None, blockSize=3)
    assert np.array_equal(c, oracle.goodFeaturesToTrack(a, 20, 0.01, 5))
    n, s, e = fe.calcOpticalFlowPyrLK(a, b, c, None, status=None, err=None, winSize=(11, 11), maxLevel=1)
    n2, s2, e2 = oracle.calcOpticalFlowPyrLK(a, b, c, None, (11, 11), 1)
    assert np.array_equal(n, n2) and np.array_equal(s, s2)


def test_block_size_and_gradient_size_argument_errors(fe):
    """cv2 raises for blockSize < 1 and for an even aperture; apertures above 7 and blocks above 31 are refused here with a
    message that says so (cv2 would run them)."""
    from visual_odometry_amd import error
    a = _seq(64, 64, 1, 1)[0]
    for kw in (dict(blockSize=0), dict(blockSize=-3), dict(gradientSize=4), dict(gradientSize=0), dict(gradientSize=9), dict(blockSize=32)):
        with pytest.raises(error):
            fe.goodFeaturesToTrack(a, 10, 0.01, 5, **kw)
    with pytest.raises(error):
        fe.corner_score(a, False, 0.04, 3, 2)


@pytest.mark.parametrize("h,w,seed,maxc,q,md", [(376, 1241, 0, 1000, 0.01, 10), (376, 1241, 0, 2000, 0.01, 7),
                                                (600, 800, 1, 3000, 0.01, 10), (600, 800, 1, 400, 0.01, 10),
                                                (480, 640, 3, 1000, 0.01, 10), (200, 300, 8, 0, 0.05, 5.5),
                                                (200, 300, 8, 50, 0.3, 0), (200, 300, 8, 100, 0.01, 0.5),
                                                (200, 300, 8, 0, 0.001, 1), (200, 300, 8, 10000, 0.001, 2.5)])
def test_gftt_identical_corner_sets(fe, h, w, seed, maxc, q, md):
    a = _seq(h, w, seed, 1)[0]
    got = fe.goodFeaturesToTrack(a, maxc, q, md)
    ref = oracle.goodFeaturesToTrack(a, maxc, q, md, nthreads=8)
    assert (got is None) == (ref is None)
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.array_equal(got, ref)            # identical set AND identical order


def test_gftt_harris_mask_and_degenerate(fe):
    a = _seq(200, 300, 8, 1)[0]
    got = fe.goodFeaturesToTrack(a, 200, 0.01, 8, useHarrisDetector=True, k=0.04)
    assert np.array_equal(got, oracle.goodFeaturesToTrack(a, 200, 0.01, 8, useHarrisDetector=True, k=0.04))
    mask = np.zeros_like(a)
    mask[50:150, 100:250] = 255
    got = fe.goodFeaturesToTrack(a, 200, 0.01, 8, mask=mask)
    ref = oracle.goodFeaturesToTrack(a, 200, 0.01, 8, mask=mask)
    assert np.array_equal(got, ref)
    assert (got[:, 0, 0] >= 100).all() and (got[:, 0, 1] < 150).all()
    assert fe.goodFeaturesToTrack(a, 200, 0.01, 8, mask=np.zeros_like(a)) is None
    assert fe.goodFeaturesToTrack(np.full((50, 60), 9, np.uint8), 100, 0.01, 5) is None      # constant image
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (60, 1))
    assert (fe.goodFeaturesToTrack(ramp, 10, 0.01, 5) is None) == (oracle.goodFeaturesToTrack(ramp, 10, 0.01, 5) is None)


def test_gftt_plateaus_and_ties(fe):
    """Isolated squares give exactly tied scores: order must be 'larger linear index first'."""
    a = np.zeros((120, 160), np.uint8)
    for (y, x) in ((20, 20), (20, 100), (80, 30), (70, 120)):
        a[y:y + 12, x:x + 12] = 200
    for md in (0, 1, 3, 20):
        got = fe.goodFeaturesToTrack(a, 0, 0.01, md)
        ref = oracle.goodFeaturesToTrack(a, 0, 0.01, md)
        assert np.array_equal(got, ref)
    chk = (np.indices((96, 128)).sum(0) // 8 % 2 * 255).astype(np.uint8)     # periodic: massive ties
    for md in (0, 4, 10):
        assert np.array_equal(fe.goodFeaturesToTrack(chk, 0, 0.01, md), oracle.goodFeaturesToTrack(chk, 0, 0.01, md))


def test_gftt_large_sort(fe):
    """More accepted corners than one LDS sort chunk (8192): exercises the global bitonic steps."""
    a = _seq(700, 900, 31, 1)[0]
    got = fe.goodFeaturesToTrack(a, 0, 0.0005, 1)
    ref = oracle.goodFeaturesToTrack(a, 0, 0.0005, 1, nthreads=8)
    assert ref.shape[0] > 8192
    assert np.array_equal(got, ref)
    got = fe.goodFeaturesToTrack(a, 20000, 0.0005, 0)
    ref = oracle.goodFeaturesToTrack(a, 20000, 0.0005, 0, nthreads=8)
    assert ref.shape[0] == 20000 and np.array_equal(got, ref)


def test_gftt_bitonic_network_forced(fe, knob):
    """The rank sort serves up to 32768 corners; the bitonic network behind it must stay exact too."""
    knob("VOFE_GFTT_SORT", "bitonic")
    a = _seq(700, 900, 31, 1)[0]
    for args in ((0, 0.0005, 1), (20000, 0.0005, 0), (500, 0.01, 6)):
        assert np.array_equal(fe.goodFeaturesToTrack(a, *args), oracle.goodFeaturesToTrack(a, *args, nthreads=8))


def test_gftt_rank_prefix_growth_and_large_windows(fe):
    """maxCorners just below / above what the image holds (the selection widens its rank prefix until
    enough corners are accepted or the list ends), and minDistance beyond the register-window limit."""
    a = _seq(600, 800, 1, 1)[0]
    full = oracle.goodFeaturesToTrack(a, 0, 0.01, 10, nthreads=8).shape[0]
    for maxc in (full - 1, full, full + 1, full // 2, 3 * full, 7):
        assert np.array_equal(fe.goodFeaturesToTrack(a, maxc, 0.01, 10), oracle.goodFeaturesToTrack(a, maxc, 0.01, 10, nthreads=8))
    for md in (9.99, 10.5, 15, 33.3, 70):
        for maxc in (0, 40):
            assert np.array_equal(fe.goodFeaturesToTrack(a, maxc, 0.01, md), oracle.goodFeaturesToTrack(a, maxc, 0.01, md, nthreads=8))


def test_gftt_batch_mixed_prefix_needs(fe):
    """One image of the batch runs out of corners while the others have plenty."""
    import torch
    imgs = np.stack([_seq(240, 320, 50 + i, 1)[0] for i in range(3)])
    imgs[1, :, :] = 10
    imgs[1, 100:140, 100:160] = _seq(40, 60, 77, 1)[0]
    out, cnt = fe.goodFeaturesToTrack_batch(torch.from_numpy(imgs).cuda(), 150, 0.01, 6)
    out = out.cpu().numpy()
    for i in range(3):
        ref = oracle.goodFeaturesToTrack(imgs[i], 150, 0.01, 6)
        assert cnt[i] == ref.shape[0]
        assert np.array_equal(out[i, :cnt[i]], ref.reshape(-1, 2))
    assert cnt[1] < 150 <= cnt[0]


def test_gftt_batch(fe):
    import torch
    imgs = np.stack([_seq(200, 300, 40 + i, 1)[0] for i in range(5)])
    out, cnt = fe.goodFeaturesToTrack_batch(torch.from_numpy(imgs).cuda(), 300, 0.01, 9)
    out = out.cpu().numpy()
    for i in range(5):
        ref = oracle.goodFeaturesToTrack(imgs[i], 300, 0.01, 9)
        assert cnt[i] == ref.shape[0]
        assert np.array_equal(out[i, :cnt[i]], ref.reshape(-1, 2))


@pytest.mark.parametrize("chain", ["merged", "split", "rank"])
def test_gftt_launch_chains_agree(fe, knob, chain):
    """The detector's last stage in its three forms -- default (one block for short rank prefixes, the order-preserving
    multi-block compaction for long ones), VOFE_GFTT_CHAIN=split (round 1's gather / rank_count / emit always) and =rank
    (one block for short prefixes, round 1's chain for long ones) -- on the reference's settings, on whole lists, without
    suppression, masked, on tied scores, on lists longer than one block holds, and batched: identical to the oracle."""
    import torch
    knob("VOFE_GFTT_CHAIN", chain)
    a = _seq(376, 1241, 0, 1)[0]
    for maxc, q, md in ((2000, 0.01, 7), (1000, 0.01, 10), (0, 0.01, 10), (300, 0.01, 0), (5000, 0.001, 3), (0, 0.05, 25.5)):
        assert np.array_equal(fe.goodFeaturesToTrack(a, maxc, q, md), oracle.goodFeaturesToTrack(a, maxc, q, md, nthreads=8)), (maxc, q, md)
    chk = (np.indices((96, 128)).sum(0) // 8 % 2 * 255).astype(np.uint8)     # every candidate in one bucket: one long run
    for md in (0, 4):
        assert np.array_equal(fe.goodFeaturesToTrack(chk, 0, 0.01, md), oracle.goodFeaturesToTrack(chk, 0, 0.01, md))
    mask = np.zeros_like(a); mask[40:300, 100:1100] = 1
    assert np.array_equal(fe.goodFeaturesToTrack(a, 500, 0.01, 8, mask=mask), oracle.goodFeaturesToTrack(a, 500, 0.01, 8, mask=mask))
    # long lists (more candidates than the one-block stage holds): whole list with and without suppression, an odd pixel
    # count, and a prefix that has to be widened (few corners asked for first, so the next call's list guess is short)
    big = _seq(701, 903, 31, 1)[0]
    for maxc, q, md in ((0, 0.0005, 2), (0, 0.0005, 0), (20000, 0.0005, 3), (40000, 0.0002, 1.5), (6000, 0.0005, 6)):
        ref = oracle.goodFeaturesToTrack(big, maxc, q, md, nthreads=8)
        assert np.array_equal(fe.goodFeaturesToTrack(big, maxc, q, md), ref), (maxc, q, md, ref.shape)
    imgs = np.stack([_seq(200, 300, 40 + i, 1)[0] for i in range(4)])
    out, cnt = fe.goodFeaturesToTrack_batch(torch.from_numpy(imgs).cuda(), 300, 0.01, 9)
    out = out.cpu().numpy()
    for i in range(4):
        ref = oracle.goodFeaturesToTrack(imgs[i], 300, 0.01, 9)
        assert cnt[i] == ref.shape[0] and np.array_equal(out[i, :cnt[i]], ref.reshape(-1, 2))


def test_gftt_last_stage_falls_back_when_the_prefix_outgrows_one_block(fe):
    """The one-block last stage is sized from the previous call's candidate count when the whole list is taken; a frame
    with far more candidates than the last one must still come out right (the kernel reports back, the split chain runs)."""
    small = _seq(60, 80, 3, 1)[0]
    big = _seq(700, 900, 31, 1)[0]
    for _ in range(2):
        assert np.array_equal(fe.goodFeaturesToTrack(small, 0, 0.01, 3), oracle.goodFeaturesToTrack(small, 0, 0.01, 3))
        ref = oracle.goodFeaturesToTrack(big, 0, 0.0005, 2, nthreads=8)
        assert ref.shape[0] > 12288
        assert np.array_equal(fe.goodFeaturesToTrack(big, 0, 0.0005, 2), ref)


def test_gftt_bad_arguments_raise(fe):
    from visual_odometry_amd import error
    a = _seq(64, 64, 1, 1)[0]
    for args in ((10, 0.0, 5), (10, 0.01, -1), (-1, 0.01, 5)):
        with pytest.raises(error):
            fe.goodFeaturesToTrack(a, *args)
    with pytest.raises(error):
        fe.goodFeaturesToTrack(a.astype(np.float32), 10, 0.01, 5)


# ------------------------------------------------------------------ frame-level tracker
def test_sequence_tracker_matches_per_call_path(fe):
    """One pyramid per frame + both point sets in one launch == ContinuousVO._track_keypoints per set."""
    from visual_odometry_amd import SequenceTracker
    H, W = 188, 310
    frames = _seq(H, W, 21, 4)
    trk = SequenceTracker(H, W, winSize=(21, 21), maxLevel=3, frontend=fe)
    trk.reset(frames[0])
    rng = np.random.default_rng(5)
    land = (rng.random((2, 150)) * [[W - 1], [H - 1]]).astype(np.float32)
    cand = (rng.random((2, 70)) * [[W - 1], [H - 1]]).astype(np.float32)
    for t in range(1, 4):
        (lp, ls, le), (cp, cs, ce), (ep, es, ee) = trk.track(frames[t], land, cand, np.empty((2, 0)))
        for pts, got in ((land, (lp, ls, le)), (cand, (cp, cs, ce))):
            ref = oracle.track_keypoints(pts, frames[t - 1], frames[t], nthreads=8)
            assert np.array_equal(got[1], ref[1])
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2])
        assert ep.shape == (2, 0) and es.shape == (0,) and ee.shape == (0,)
        det = trk.detect(200, 0.01, 7)
        assert np.array_equal(det, oracle.goodFeaturesToTrack(frames[t], 200, 0.01, 7))
        land, cand = lp[:, ls == 1], cp[:, cs == 1]        # carry the survivors, as process_frame does
    trk.close()
    # ingest overlap: prefetching the next frame changes nothing in the results
    trk = SequenceTracker(H, W, winSize=(21, 21), maxLevel=3, frontend=fe)
    trk.prefetch(frames[0])
    trk.reset(frames[0])
    pts0 = (rng.random((2, 120)) * [[W - 1], [H - 1]]).astype(np.float32)
    for t in range(1, 4):
        trk.prefetch(frames[t])
        (p1, s1, e1), = trk.track(frames[t], pts0)
        ref = oracle.track_keypoints(pts0, frames[t - 1], frames[t], nthreads=8)
        assert np.array_equal(p1, ref[0]) and np.array_equal(s1, ref[1]) and np.array_equal(e1, ref[2])
    trk.prefetch(frames[1])                                 # a prefetched frame that is then not the one passed
    (p1, s1, e1), = trk.track(frames[0], pts0)
    assert np.array_equal(s1, oracle.track_keypoints(pts0, frames[3], frames[0], nthreads=8)[1])
    trk.close()
    with pytest.raises(Exception):
        SequenceTracker(H, W, frontend=fe).track(frames[0], land)


def test_sequence_tracker_overlapped_track_and_detect(fe):
    """track_and_detect runs the detector on a second context/stream while the tracker works: same triples and the same
    corner list as track() followed by detect(), frame after frame (also after a prefetch)."""
    from visual_odometry_amd import SequenceTracker
    frames = _seq(376, 1241, 21, 5)
    pts = oracle.goodFeaturesToTrack(frames[0], 800, 0.01, 7).reshape(-1, 2).T
    a = SequenceTracker(376, 1241, frontend=fe)
    b = SequenceTracker(376, 1241, frontend=fe)
    a.reset(frames[0]); b.reset(frames[0])
    cur = pts
    for t in range(1, 5):
        if t == 3:
            b.prefetch(frames[t])
        ref = a.track(frames[t], cur, cur[:, :50], np.empty((2, 0)))
        ref_c = a.detect(500, 0.01, 10)
        got, got_c = b.track_and_detect(frames[t], (cur, cur[:, :50], np.empty((2, 0))), 500, 0.01, 10)
        assert len(got) == 3 and np.array_equal(got_c, ref_c) and np.array_equal(ref_c, oracle.goodFeaturesToTrack(frames[t], 500, 0.01, 10))
        for (x0, s0, e0), (x1, s1, e1) in zip(ref, got):
            assert np.array_equal(x0, x1) and np.array_equal(s0, s1) and np.array_equal(e0, e1)
        cur = ref[0][0][:, ref[0][1] == 1]
    a.close(); b.close()


def test_multi_sequence_tracker_equals_independent_trackers(fe):
    """configs[3] at the user level: B sequences advanced by one pyramid build + one launch per frame index give what
    B independent SequenceTrackers give (ragged point counts, one sequence without points, batched detection)."""
    from visual_odometry_amd import SequenceTracker, MultiSequenceTracker
    B, h, w = 5, 200, 320
    seqs = [_seq(h, w, 700 + s, 4) for s in range(B)]
    multi = MultiSequenceTracker(B, h, w, frontend=fe)
    singles = [SequenceTracker(h, w, frontend=fe) for _ in range(B)]
    multi.reset(np.stack([q[0] for q in seqs]))
    for s, t in enumerate(singles):
        t.reset(seqs[s][0])
    det = multi.detect(150, 0.01, 8)
    pts = []
    for s in range(B):
        ref = singles[s].detect(150, 0.01, 8)
        assert (det[s] is None) == (ref is None) and (ref is None or np.array_equal(det[s], ref))
        pts.append(ref.reshape(-1, 2).T[:, : 150 - 37 * s] if s != 3 else np.empty((2, 0)))
    for f in range(1, 4):
        got = multi.track(np.stack([q[f] for q in seqs]), pts)
        nxt = []
        for s in range(B):
            (rt, rs, re), = singles[s].track(seqs[s][f], pts[s])
            gt, gs, ge = got[s]
            assert gt.shape == rt.shape and np.array_equal(gt, rt) and np.array_equal(gs, rs) and np.array_equal(ge, re), (f, s)
            nxt.append(rt[:, rs == 1] if rt.size else rt)
        pts = nxt
    assert sum(p.shape[1] for p in pts) > 200
    multi.close()
    for t in singles:
        t.close()


# ------------------------------------------------------------------ BASELINE.json full-size configurations
@pytest.mark.parametrize("cfg,md", [("malaga", 10), ("stress", 10)])
def test_full_size_config_front_end_frame(fe, cfg, md):
    """configs[2] (Malaga-shaped detection + tracking) and configs[4] (3840x2160, 20k points, 31x31, 5 levels)
    at full size: detection identical to the oracle, forward-backward tracking bit-identical, and the
    size-independent properties (sorted scores, minDistance respected, FB-consistent survivors)."""
    H, W, seed, n, win, lvl = synthetic.CONFIGS[cfg]
    a, b = synthetic.sequence(H, W, seed, 2)
    got = fe.goodFeaturesToTrack(a, n, 0.01, md)
    ref = oracle.goodFeaturesToTrack(a, n, 0.01, md, nthreads=16)
    assert np.array_equal(got, ref)
    xy = got.reshape(-1, 2)
    sc = fe.corner_score(a)[xy[:, 1].astype(int), xy[:, 0].astype(int)]
    assert (np.diff(sc) <= 0).all()                                   # strongest first
    from scipy.spatial import cKDTree
    assert len(cKDTree(xy).query_pairs(md - 1e-6)) == 0               # no two corners closer than minDistance
    pts = synthetic.pad_points(got, n, H, W, seed).T.copy()
    t1, s1, e1 = fe.track_keypoints(pts, a, b, win, lvl)
    t0, s0, e0 = oracle.track_keypoints(pts, a, b, win, lvl, nthreads=16)
    assert np.array_equal(s1, s0) and np.array_equal(t1, t0) and np.array_equal(e1, e0)
    assert s1.mean() > 0.8
    # survivors are forward-backward consistent by construction: tracking them back lands within 1 px
    keep = s1 == 1
    back, sb, _ = fe.calcOpticalFlowPyrLK(b, a, t1.T[keep].reshape(-1, 1, 2).astype(np.float32), None, win, lvl, (3, 30, 0.01), 0, 1e-3)
    d = np.linalg.norm(back.reshape(-1, 2) - pts.T[keep], axis=1)
    assert (d[sb.ravel() == 1] < 1.0).all()


@pytest.mark.parametrize("kernel", ["x1", "x2", "x2p", "x4"])
def test_ragged_batch_of_sequences_in_one_launch(fe, knob, kernel):
    """configs[3] batching: B pairs in one pyramid object, points of all pairs in ONE launch with a per-point
    image index (ragged: different counts per pair, one pair with none) == the per-pair results."""
    import torch
    from visual_odometry_amd import _lib
    from visual_odometry_amd.frontend import Pyramid
    knob("VOFE_LK", kernel)
    H, W, B = 120, 200, 4
    pairs = [_seq(H, W, 60 + s, 2) for s in range(B)]
    counts = [90, 0, 37, 150]
    rng = np.random.default_rng(1)
    pts = [(rng.random((c, 2)) * [W - 1, H - 1]).astype(np.float32) for c in counts]
    imgs = torch.from_numpy(np.stack([p[0] for p in pairs] + [p[1] for p in pairs])).cuda()
    pyr = Pyramid(fe, 2 * B, H, W, (21, 21), 3, True)
    pyr.build(imgs, 0)
    allp = torch.from_numpy(np.concatenate(pts)).cuda()
    idx = torch.from_numpy(np.repeat(np.arange(B, dtype=np.int32), counts)).cuda()
    n = int(sum(counts))
    nxt = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    st = torch.empty(n, dtype=torch.uint8, device="cuda")
    err = torch.empty(n, dtype=torch.float32, device="cuda")
    _lib.check(fe.L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, allp.data_ptr(), idx.data_ptr(), B, nxt.data_ptr(), st.data_ptr(),
                                  err.data_ptr(), n, 3, 30, 0.01, 1e-3, 1.0))
    nxt, st, err = nxt.cpu().numpy(), st.cpu().numpy(), err.cpu().numpy()
    o = 0
    for s in range(B):
        if counts[s] == 0:
            continue
        t0, s0, e0 = oracle.track_keypoints(pts[s].T, pairs[s][0], pairs[s][1], nthreads=4)
        k = counts[s]
        assert np.array_equal(nxt[o:o + k].T, t0) and np.array_equal(st[o:o + k], s0) and np.array_equal(err[o:o + k], e0)
        o += k
    pyr.free()


def test_bench_launch_shape_default_dispatch_64_pairs(fe):
    """The launch bench.py times, checked at the shape it is timed at: 64 KITTI-shaped pairs (sequences 100 + s),
    2000 points each, ONE vofe_pyramid_build of 128 slots and ONE vofe_track_fb of 128 000 points with a per-point
    image index -- so the size rule (not VOFE_LK) picks lk_track_kernel_x2<2> and the 32-bit slot offsets span a
    ~290 MB derivative plane set.  Every pair must equal oracle.track_keypoints bit for bit."""
    import os
    import torch
    from visual_odometry_amd import _lib
    from visual_odometry_amd.frontend import Pyramid
    assert not os.environ.get("VOFE_LK")
    H, W, _, n, win, lvl = synthetic.CONFIGS["kitti"]
    B = 64
    pairs = [synthetic.sequence(H, W, seed=100 + s, n_frames=2) for s in range(B)]
    d_prev = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    d_next = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    det, cnt = fe.goodFeaturesToTrack_batch(d_prev, n, 0.01, 7)
    det = det.cpu().numpy()
    pts = [synthetic.pad_points(det[s, :cnt[s]], n, H, W, 100 + s) for s in range(B)]
    pyr = Pyramid(fe, 2 * B, H, W, win, lvl, True)
    pyr.build(torch.cat([d_prev, d_next]), 0)
    allp = torch.from_numpy(np.concatenate(pts)).cuda()
    idx = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(n).contiguous()
    N = n * B
    nxt = torch.empty((N, 2), dtype=torch.float32, device="cuda")
    st = torch.empty(N, dtype=torch.uint8, device="cuda")
    err = torch.empty(N, dtype=torch.float32, device="cuda")
    _lib.check(fe.L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, allp.data_ptr(), idx.data_ptr(), B, nxt.data_ptr(),
                                  st.data_ptr(), err.data_ptr(), N, 3, 30, 0.01, 1e-3, 1.0))
    nxt, st, err = nxt.cpu().numpy(), st.cpu().numpy(), err.cpu().numpy()
    tracked = 0
    for s in range(B):
        t0, s0, e0 = oracle.track_keypoints(pts[s].T, pairs[s][0], pairs[s][1], win, lvl, (3, 30, 0.01), 1e-3, 1.0, nthreads=16)
        o = s * n
        assert np.mean(st[o:o + n] == s0) >= STATUS_MIN, s
        ok = (st[o:o + n] == 1) & (s0 == 1)
        assert np.abs(nxt[o:o + n].T - t0)[:, ok].max() <= POS_TOL, s
        assert np.array_equal(st[o:o + n], s0) and np.array_equal(nxt[o:o + n].T, t0) and np.array_equal(err[o:o + n], e0), s
        tracked += int(s0.sum())
    assert tracked > 0.9 * N
    # an image index outside [0, n_images) loses the point instead of addressing memory outside the batch, and the
    # slot range the indices can reach is checked on the host
    bad = idx.clone()
    bad[5] = B
    bad[N - 3] = -1
    bad[7001] = 1 << 30
    d_st2 = torch.empty(N, dtype=torch.uint8, device="cuda")
    nxt2 = torch.empty((N, 2), dtype=torch.float32, device="cuda")
    err2 = torch.empty(N, dtype=torch.float32, device="cuda")
    _lib.check(fe.L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, allp.data_ptr(), bad.data_ptr(), B, nxt2.data_ptr(),
                                  d_st2.data_ptr(), err2.data_ptr(), N, 3, 30, 0.01, 1e-3, 1.0))
    st2 = d_st2.cpu().numpy()
    keep = np.ones(N, bool)
    keep[[5, N - 3, 7001]] = False
    assert not st2[~keep].any() and np.array_equal(st2[keep], st[keep]) and np.array_equal(nxt2.cpu().numpy()[keep], nxt[keep])
    assert (err2.cpu().numpy()[~keep] == 0).all()
    assert fe.L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, allp.data_ptr(), idx.data_ptr(), B + 1, nxt2.data_ptr(),
                              d_st2.data_ptr(), err2.data_ptr(), N, 3, 30, 0.01, 1e-3, 1.0) == -1
    assert fe.L.vofe_track_fb(fe.ctx, pyr._h, B + 1, pyr._h, 0, allp.data_ptr(), idx.data_ptr(), B, nxt2.data_ptr(),
                              d_st2.data_ptr(), err2.data_ptr(), N, 3, 30, 0.01, 1e-3, 1.0) == -1
    pyr.free()


def test_config3_64_kitti_sequences_full_shape(fe):
    """BASELINE.json configs[3] as SURVEY 8d defines it, at full shape: 64 independent KITTI-shaped 1241x376 sequences
    (seeds 100 + s), K = 1000 landmarks + M = 1000 candidates per sequence, advanced over 3 frame steps by ONE pyramid
    build + ONE tracker launch per frame index (MultiSequenceTracker) plus batched detection -- every sequence, every
    frame, both point sets and the corner lists against the oracle, bit for bit."""
    from visual_odometry_amd import MultiSequenceTracker
    H, W, _, _, win, lvl = synthetic.CONFIGS["kitti"]
    B, K, M, F = 64, 1000, 1000, 4
    seqs = [synthetic.sequence(H, W, seed=100 + s, n_frames=F) for s in range(B)]
    multi = MultiSequenceTracker(B, H, W, winSize=win, maxLevel=lvl, frontend=fe)
    multi.reset(np.stack([q[0] for q in seqs]))
    det = multi.detect(K + M, 0.01, 7)
    land, cand = [], []
    for s in range(B):
        ref = oracle.goodFeaturesToTrack(seqs[s][0], K + M, 0.01, 7, nthreads=16)
        assert np.array_equal(det[s], ref), s
        p = synthetic.pad_points(ref, K + M, H, W, 100 + s)
        land.append(p[0::2].T.copy())            # strongest corners alternate between the two sets
        cand.append(p[1::2].T.copy())
    total = 0
    for f in range(1, F):
        sets = [np.concatenate([land[s], cand[s]], axis=1) for s in range(B)]
        got = multi.track(np.stack([q[f] for q in seqs]), sets)
        corners = multi.detect(1000, 0.01, 10)                        # the reference's per-frame detection (constants.py:31-33)
        for s in range(B):
            kl = land[s].shape[1]
            gt, gs, ge = got[s]
            for name, pts, sl in (("landmarks", land[s], slice(0, kl)), ("candidates", cand[s], slice(kl, None))):
                rt, rs, re = oracle.track_keypoints(pts, seqs[s][f - 1], seqs[s][f], win, lvl, (3, 30, 0.01), 1e-3, 1.0, nthreads=16)
                assert np.array_equal(gs[sl], rs) and np.array_equal(gt[:, sl], rt) and np.array_equal(ge[sl], re), (f, s, name)
            assert np.array_equal(corners[s], oracle.goodFeaturesToTrack(seqs[s][f], 1000, 0.01, 10, nthreads=16)), (f, s)
            # carry the survivors, as process_frame does (continuous.py:90-92, 391-396)
            land[s] = gt[:, :kl][:, gs[:kl] == 1]
            cand[s] = gt[:, kl:][:, gs[kl:] == 1]
            total += land[s].shape[1] + cand[s].shape[1]
    assert total > 0.8 * B * (K + M) * (F - 1)
    multi.close()


def test_gftt_host_capacity_below_the_corner_count(fe):
    """vofe_gftt_host delivers min(count, capacity) corners to the host buffer and never writes past it: capacity below
    maxCorners, and maxCorners = 0 with a small capacity (guard region behind the buffer)."""
    import ctypes as C
    import torch
    a = _seq(240, 320, 12, 1)[0]
    d = torch.from_numpy(a).cuda()
    for maxc, cap in ((300, 40), (0, 25), (300, 1), (50, 400)):
        ref = oracle.goodFeaturesToTrack(a, maxc, 0.01, 6).reshape(-1, 2)
        host = np.full((cap + 64, 2), -3.0, np.float32)
        dev = torch.full((cap, 2), -3.0, dtype=torch.float32, device="cuda")
        n = C.c_int(-1)
        rc = fe.L.vofe_gftt_host(fe.ctx, d.data_ptr(), 240, 320, 320, maxc, 0.01, 6.0, None, 3, 3, 0, 0.04, dev.data_ptr(),
                                 host.ctypes.data, cap, C.byref(n))
        assert rc == 0 and n.value == ref.shape[0], (maxc, cap, n.value)
        m = min(cap, ref.shape[0])
        assert np.array_equal(host[:m], ref[:m]) and (host[m:] == -3.0).all(), (maxc, cap)
        assert np.array_equal(dev.cpu().numpy()[:m], ref[:m])


# ------------------------------------------------------------------ candidate bookkeeping
def test_new_feature_mask_matches_scipy_cdist_rule(fe):
    from scipy.spatial.distance import cdist
    from test_oracle_cpu import dedup_cases
    for f, e, thr in dedup_cases():
        with np.errstate(invalid="ignore"):
            ref = np.min(cdist(f, e), axis=1) > thr
            assert np.array_equal(oracle.new_feature_mask(f, e, thr), ref)
        got = fe.new_feature_mask(f, e, thr)
        assert got.dtype == bool and np.array_equal(got, ref), (f.shape, e.shape, thr)
    # the reference's use: detections of the current frame against candidates + landmarks
    a = _seq(376, 1241, 0, 1)[0]
    det = oracle.goodFeaturesToTrack(a, 1000, 0.01, 10).reshape(-1, 2)
    rng = np.random.default_rng(2)
    existing = np.vstack([det[::3] + rng.normal(0, 6, det[::3].shape).astype(np.float32), (rng.random((400, 2)) * [1241, 376]).astype(np.float32)])
    new = fe.filter_new_features(det, existing, 15)
    ref = det[np.min(cdist(det, existing), axis=1) > 15].T
    assert new.shape == ref.shape and np.array_equal(new, ref) and 0 < new.shape[1] < det.shape[0]
    assert fe.new_feature_mask(np.empty((0, 2), np.float32), existing).shape == (0,)
    with pytest.raises(ValueError):
        fe.new_feature_mask(det, np.empty((0, 2), np.float32))
    for bad_thr in (np.inf, np.nan):
        assert not fe.new_feature_mask(det[:10], existing, bad_thr).any()


def test_filter_candidates_matches_reference_numpy_expressions(fe):
    """vofe_candidate_filter vs continuous.py:391-405/446-448 evaluated by numpy itself: same kept set, order, values
    and triangulation flags, bit for bit, for float32 and float64 first observations."""
    from test_oracle_cpu import candidate_cases
    kept = 0
    for Ci, st, Fi, Ti, a, b, d in candidate_cases():
        with np.errstate(invalid="ignore", over="ignore"):
            rc, rf, rt, rtri = oracle.filter_candidates(Ci, st, Fi, Ti, a, b, d)
        c, f, t, tri = fe.filter_candidates(Ci, st, Fi, Ti, a, b, d)
        tag = (Ci.shape, Fi.dtype, a, b, d)
        assert c.shape == rc.shape and c.dtype == np.float32 and f.dtype == Fi.dtype and t.dtype == Ti.dtype, tag
        assert np.array_equal(c, rc, equal_nan=True) and np.array_equal(f, rf, equal_nan=True), tag
        assert np.array_equal(t, rt) and tri.dtype == bool and np.array_equal(tri, rtri), tag
        kept += c.shape[1]
    assert kept > 3000
    import visual_odometry_amd as voa
    with pytest.raises(voa.error):
        fe.filter_candidates(np.zeros((2, 3)), np.ones(3, np.uint8), np.zeros((2, 3), np.float32), np.zeros((12, 3)))
    with pytest.raises(voa.error):
        fe.filter_candidates(np.zeros((2, 3), np.float32), np.ones(3, np.uint8), np.zeros((2, 3), np.float32), np.zeros((17, 3)))


def test_filter_candidates_device_resident_after_tracking(fe):
    """The reference's use: candidates tracked by vofe_track_fb, then compacted in place on the device; the kept
    columns equal what the reference's numpy lines keep from the host copies."""
    import torch
    a, b = _seq(376, 1241, 0, 2)
    first = oracle.goodFeaturesToTrack(a, 1500, 0.01, 7).reshape(-1, 2)
    rng = np.random.default_rng(5)
    prev = (first + rng.normal(0, 8, first.shape)).astype(np.float32)       # candidates have travelled since first seen
    trk, st, _ = fe.track_keypoints(prev.T, a, b)
    Ti = rng.normal(0, 1, (12, prev.shape[0]))
    rc, rf, rt, rtri = oracle.filter_candidates(trk, st, first.T.copy(), Ti)
    d_c = torch.from_numpy(np.ascontiguousarray(trk.T)).cuda()
    d_f = torch.from_numpy(first.copy()).cuda()
    d_t = torch.from_numpy(np.ascontiguousarray(Ti.T)).cuda()
    d_s = torch.from_numpy(st.copy()).cuda()
    d_tri = torch.empty(prev.shape[0], dtype=torch.uint8, device="cuda")
    m = fe.filter_candidates_device(d_c, d_s, d_f, d_t, d_tri)
    assert m == rc.shape[1] and 100 < m < prev.shape[0]
    assert np.array_equal(d_c[:m].cpu().numpy().T, rc) and np.array_equal(d_f[:m].cpu().numpy().T, rf)
    assert np.array_equal(d_t[:m].cpu().numpy().T, rt) and np.array_equal(d_tri[:m].cpu().numpy().astype(bool), rtri)
    assert rtri.any() and not rtri.all()
    # no pose matrix: positions only
    d_c = torch.from_numpy(np.ascontiguousarray(trk.T)).cuda()
    d_f = torch.from_numpy(first.copy()).cuda()
    assert fe.filter_candidates_device(d_c, d_s, d_f, None, d_tri) == m
    assert np.array_equal(d_c[:m].cpu().numpy().T, rc)


def test_triangulate_candidates_against_restated_loop(fe):
    """vofe_triangulate_candidates vs the reference's loop with a LAPACK SVD in place of cv2.triangulatePoints
    (cv2 is absent: this row is tolerance-pinned, not bit-pinned).  3D points agree to SVD rounding error; the
    keep/drop decisions are identical for every candidate whose decision does not hang on that rounding."""
    from test_oracle_cpu import triangulation_scene
    total = 0
    for n, seed, fdt in ((1, 1, np.float32), (200, 2, np.float32), (1500, 3, np.float64), (64, 5, np.float32)):
        K, pose, Ci, Fi, Ti, flags = triangulation_scene(n, seed, fdt)
        margins = []
        with np.errstate(all="ignore"):
            rkeep, rpts = oracle.triangulate_candidates(K, pose, Ci, Fi, Ti, flags, margins=margins)
        keep, pts = fe.triangulate_candidates(K, pose, Ci, Fi, Ti, flags)
        assert keep.dtype == bool and keep.shape == rkeep.shape and pts.dtype == np.float64 and pts.shape[0] == 3
        shaky = np.zeros(n, bool)
        for idx, depth_margin, angle_margin in margins:
            shaky[idx] = not (abs(depth_margin) > 1e-3 and abs(angle_margin) > 1e-5)
        assert np.array_equal(keep[~shaky], rkeep[~shaky]) and shaky.sum() <= max(2, n // 50)
        assert not (keep & ~flags).any()
        both = keep & rkeep
        a = pts[:, np.cumsum(keep)[both] - 1]
        b = rpts[:, np.cumsum(rkeep)[both] - 1]
        tol = 2e-3 if fdt == np.float32 else 1e-7           # float32 points4D: 1 ulp of a homogeneous coordinate
        assert np.allclose(a, b, rtol=tol, atol=tol * 10), np.abs(a - b).max()
        total += int(both.sum())
    assert total > 300
    none, empty = fe.triangulate_candidates(K, pose, Ci, Fi, Ti, np.zeros(Ci.shape[1], bool))
    assert not none.any() and empty.shape == (3, 0)


def test_handle_candidates_follows_the_reference_method_over_a_sequence(fe):
    """FrontEnd.handle_candidates (every stage on the device) against the line-by-line restatement of
    ContinuousVO._handle_candidates (continuous.py:375-495) over six frames: both get the same previous state each
    frame; landmark / candidate arrays must come out identical, triangulated points to SVD rounding."""
    frames = _seq(376, 1241, 3, 7)
    K = np.array([[718.856, 0, 607.19], [0, 718.856, 185.2], [0, 0, 1.0]])

    def pose_at(t):
        T = np.eye(4)
        T[:3, 3] = [0.02 * t, 0.0, 0.6 * t]
        return T

    first = oracle.goodFeaturesToTrack(frames[0], 600, 0.01, 10).reshape(-1, 2)
    Ci = first.T.copy()                                           # (2,M) float32
    Fi = first.T.astype(np.float64)                               # the initial state is float64 in the reference
    Ti = np.tile(pose_at(0)[:3, :].reshape(-1, 1), (1, Ci.shape[1]))
    Pi = np.empty((2, 0), np.float32)
    Xi = np.empty((3, 0))
    promoted = grown = 0
    for t in range(1, 7):
        pose = pose_at(t)
        args = (K, frames[t], frames[t - 1], Ci, Fi, Ti, Pi, Xi, pose)
        kw = dict(max_corners=600, quality_level=0.01, min_distance=10, threshold_alpha_deg=0.05)
        with np.errstate(all="ignore"):
            rPi, rXi, rCi, rFi, rTi = oracle.handle_candidates(*args, **kw)
        gPi, gXi, gCi, gFi, gTi = fe.handle_candidates(*args, **kw)
        for name, g, r in (("Pi", gPi, rPi), ("Ci", gCi, rCi), ("Fi", gFi, rFi), ("Ti", gTi, rTi)):
            assert g.shape == r.shape and g.dtype == r.dtype and np.array_equal(g, r), (t, name, g.shape, r.shape)
        assert gXi.shape == rXi.shape and gXi.dtype == rXi.dtype and np.allclose(gXi, rXi, rtol=1e-6, atol=1e-6), t
        promoted += rPi.shape[1] - Pi.shape[1]
        grown += int(rCi.shape[1] > 0)
        Pi, Xi, Ci, Fi, Ti = rPi, rXi, rCi, rFi, rTi
    assert promoted > 0 and grown == 6 and Ci.shape[1] > 100


def test_device_resident_frame_state_follows_the_reference_frame_loop(fe):
    """DeviceFrameState (landmarks and candidates resident in HBM, two device phases per frame around the CPU pose
    estimation) against the reference's per-frame sequence restated with the oracle: landmark tracking
    (continuous.py:86-92), a stand-in for the pose estimation's inlier selection, _handle_candidates (:375-495).  Every
    frame both sides start from their OWN previous state, so differences would accumulate; all arrays must stay equal."""
    from visual_odometry_amd import DeviceFrameState
    H, W = 376, 1241
    frames = _seq(H, W, 3, 7)
    K = np.array([[718.856, 0, 607.19], [0, 718.856, 185.2], [0, 0, 1.0]])

    def pose_at(t):
        T = np.eye(4)
        T[:3, 3] = [0.02 * t, 0.0, 0.6 * t]
        return T

    first = oracle.goodFeaturesToTrack(frames[0], 900, 0.01, 10).reshape(-1, 2)
    rng = np.random.default_rng(4)
    Pi = first[::3].T.copy()                                     # a third of the detections start as landmarks
    Xi = rng.normal(0, 5, (3, Pi.shape[1])) + [[0], [0], [20]]
    cand = np.delete(first, np.arange(0, first.shape[0], 3), axis=0)
    Ci = cand.T.copy()
    Fi = cand.T.astype(np.float64)
    Ti = np.tile(pose_at(0)[:3, :].reshape(-1, 1), (1, Ci.shape[1]))
    kw = dict(max_corners=600, quality_level=0.01, min_distance=10, threshold_alpha_deg=0.05)
    st = DeviceFrameState(H, W, K, frontend=fe, **kw)
    st.reset(frames[0], Pi, Xi, Ci, Fi, Ti)
    for name, g, r in zip(("Pi", "Xi", "Ci", "Fi", "Ti"), st.download(), (Pi, Xi, Ci, Fi, Ti)):
        assert g.shape == r.shape and np.array_equal(g, r), name
    promoted = 0
    for t in range(1, 7):
        pose = pose_at(t)
        # reference side (oracle for the cv2 calls)
        rt, rs, _ = oracle.track_keypoints(Pi, frames[t - 1], frames[t], nthreads=8)
        gt, gs = st.track(frames[t])
        assert gt.dtype == np.float32 and np.array_equal(gt, rt) and np.array_equal(gs, rs), t
        Xw = st.landmarks_world                                  # between the phases: the world points PnP pairs with `gt`
        assert Xw.shape == Xi.shape and np.allclose(Xw, Xi, rtol=1e-6, atol=1e-6), t
        keep = rs == 1
        keep[::7] = False                                        # stand-in for the RANSAC inlier mask (continuous.py:95-101)
        Pi, Xi = rt[:, keep], Xi[:, keep]
        with np.errstate(all="ignore"):
            rPi, rXi, Ci, Fi, Ti = oracle.handle_candidates(K, frames[t], frames[t - 1], Ci, Fi, Ti, Pi, Xi, pose, **kw)
        promoted += st.update(pose, keep)
        gPi, gXi, gCi, gFi, gTi = st.download()
        for name, g, r in (("Pi", gPi, rPi), ("Ci", gCi, Ci), ("Fi", gFi, Fi), ("Ti", gTi, Ti)):
            assert g.shape == r.shape and np.array_equal(g, r), (t, name, g.shape, r.shape)
        assert gPi.dtype == np.float32 and gFi.dtype == np.float64
        assert gXi.shape == rXi.shape and np.allclose(gXi, rXi, rtol=1e-6, atol=1e-6), t
        Pi, Xi = rPi, rXi
    assert promoted > 0
    assert Ci.shape[1] > 100 and Pi.shape[1] > 100
    with pytest.raises(Exception):
        st.update(pose_at(7), np.ones(st.K, bool))               # update() without track()
    st.close()


def test_native_frame_state_edges_float32_staging_buffer_empty_and_overflow(fe):
    """vofe_framestate through the C ABI beyond the main sequence test: float32 first observations (numpy then evaluates
    the distance band in float32, continuous.py:398-401), frames written straight into the object's pinned staging
    buffer, a start without landmarks, the reference's ValueError when detections meet an empty state, the capacity
    error, and the call-order checks."""
    from visual_odometry_amd import DeviceFrameState
    H, W = 240, 320
    frames = _seq(H, W, 11, 5)
    K = np.array([[300.0, 0, 160], [0, 300.0, 120], [0, 0, 1.0]])

    def pose_at(t):
        T = np.eye(4)
        T[:3, 3] = [0.05 * t, 0.0, 0.8 * t]
        return T

    kw = dict(max_corners=300, quality_level=0.01, min_distance=8, threshold_alpha_deg=0.02, threshold_triangulation=3)
    lk = dict(winSize=(15, 15), maxLevel=2)
    first = oracle.goodFeaturesToTrack(frames[0], 300, 0.01, 8).reshape(-1, 2)
    Ci, Fi = first.T.copy(), first.T.copy()                      # float32 first observations
    Ti = np.tile(pose_at(0)[:3, :].reshape(-1, 1), (1, Ci.shape[1]))
    Pi, Xi = np.empty((2, 0), np.float32), np.empty((3, 0))
    st = DeviceFrameState(H, W, K, first_dtype=np.float32, cap_landmarks=512, cap_candidates=1024, **lk, **kw)
    with pytest.raises(Exception):
        st.track(frames[1])                                      # track() before reset()
    st.reset(frames[0], Pi, Xi, Ci, Fi, Ti)
    assert st.K == 0 and st.M == Ci.shape[1]
    promoted = 0
    for t in range(1, 5):
        st.frame_buffer[...] = frames[t]                         # "the decoder" writes into pinned memory
        gt, gs = st.track(st.frame_buffer)
        rt, rs, _ = oracle.track_keypoints(Pi, frames[t - 1], frames[t], **lk)
        assert gt.shape == (2, Pi.shape[1]) and np.array_equal(gt, rt) and np.array_equal(gs, rs), t
        with pytest.raises(Exception):
            st.track(frames[t])                                  # twice without update()
        with pytest.raises(Exception):
            st.download()                                        # between the phases
        keep = rs == 1
        Pi, Xi = rt[:, keep].astype(np.float32), Xi[:, keep]
        with np.errstate(all="ignore"):
            rPi, rXi, Ci, Fi, Ti = oracle.handle_candidates(K, frames[t], frames[t - 1], Ci, Fi, Ti, Pi, Xi, pose_at(t), lk_params=lk, **kw)
        # queued update: on even frames nothing collects it before the next track() does (which first copies, uploads
        # and builds the pyramid of its frame); on odd frames download() collects it
        assert st.update(pose_at(t), keep, wait=False) is None
        Pi, Xi = rPi, rXi
        if t % 2 == 0:
            continue
        gPi, gXi, gCi, gFi, gTi = st.download()
        promoted += st.wait()                                    # already collected: the count of that update
        assert gFi.dtype == np.float32 and Fi.dtype == np.float32
        for name, g, r in (("Pi", gPi, rPi), ("Ci", gCi, Ci), ("Fi", gFi, Fi), ("Ti", gTi, Ti)):
            assert g.shape == r.shape and np.array_equal(g, r), (t, name, g.shape, r.shape)
        assert gXi.shape == rXi.shape and np.allclose(gXi, rXi, rtol=2e-3, atol=2e-2), t      # float32 points4D
    assert promoted > 0 and Pi.shape[1] >= promoted and st.K == Pi.shape[1]
    # an empty state meeting detections: the reference's np.min over an empty axis raises ValueError
    st.reset(frames[0], np.empty((2, 0)), np.empty((3, 0)))
    gt, gs = st.track(frames[1])
    assert gt.shape == (2, 0) and gs.shape == (0,)
    with pytest.raises(ValueError):
        st.update(pose_at(1), np.empty(0, bool))
    st.close()
    # capacity: more new candidates than the arrays hold
    import visual_odometry_amd
    assert first.shape[0] > 100
    small = DeviceFrameState(H, W, K, cap_landmarks=64, cap_candidates=64, **lk, **kw)
    with pytest.raises(visual_odometry_amd.error):
        small.reset(frames[0], first[:4].T, np.zeros((3, 4)), Ci[:, :65], Fi[:, :65], Ti[:, :65])      # over capacity at the start
    small.reset(frames[0], first[:4].T, np.zeros((3, 4)))
    _, s = small.track(frames[1])
    with pytest.raises(visual_odometry_amd.error) as ei:
        small.update(pose_at(1), s == 1)                         # ~300 detections into 64 candidate slots
    assert ei.value.code == -4
    with pytest.raises(Exception):
        small.track(frames[1])                                   # the state has to be reset after an overflow
    small.close()


def test_framestate_c_abi_raw_calls_pitched_frames_bad_arguments_two_objects(fe):
    """vofe_framestate_* called straight through ctypes, as a C host would: frames handed over with a row pitch larger
    than the width, argument errors (NULL pointers, pitch below the width, bad parameters) answered with VOFE_E_ARG and a
    message instead of a fault, and two objects on one GPU advanced in turn (each owns its streams and buffers) giving
    what one object gives alone."""
    import ctypes as C
    from visual_odometry_amd import _lib
    L = _lib.load()
    H, W = 200, 300
    frames = _seq(H, W, 23, 4)
    pitched = [np.zeros((H, W + 52), np.uint8) for _ in frames]
    for dst, src in zip(pitched, frames):
        dst[:, :W] = src
        dst[:, W:] = 255                                          # must never be read as pixels
    p = _lib.FrameStateParams()
    p.height, p.width, p.win_w, p.win_h, p.max_level = H, W, 21, 21, 3
    p.crit_type, p.max_count, p.eps, p.min_eig_thr, p.fb_thresh = 3, 30, 0.01, 1e-3, 1.0
    p.max_corners, p.quality_level, p.min_distance = 200, 0.01, 10.0
    p.thr_new_keypoints, p.thr_triangulation, p.thr_candidates_min, p.thr_candidates_max = 15.0, 3.0, 0.5, 100.0
    p.thr_alpha_rad = float(np.deg2rad(0.02))
    p.cap_landmarks, p.cap_candidates, p.first_is_f64 = 1024, 1024, 1
    p.K[:] = [250.0, 0, 150, 0, 250.0, 100, 0, 0, 1.0]
    h = C.c_void_p()
    bad = _lib.FrameStateParams.from_buffer_copy(p)
    bad.max_corners = 0
    assert L.vofe_framestate_create(0, C.byref(bad), C.byref(h)) == -1 and not h.value
    assert L.vofe_framestate_create(0, None, C.byref(h)) == -1
    assert L.vofe_framestate_create(99, C.byref(p), C.byref(h)) != 0 and not h.value
    assert b"device" in L.vofe_last_error()
    objs = []
    for _ in range(3):
        assert L.vofe_framestate_create(0, C.byref(p), C.byref(h)) == 0, L.vofe_last_error()
        objs.append(C.c_void_p(h.value))
    first = oracle.goodFeaturesToTrack(frames[0], 200, 0.01, 10).reshape(-1, 2)
    P0 = np.ascontiguousarray(first[:60])
    X0 = np.ascontiguousarray(np.random.default_rng(2).normal(0, 4, (60, 3)) + [0, 0, 15])
    C0 = np.ascontiguousarray(first[60:])
    F0 = C0.astype(np.float64)
    pose0 = np.eye(4)[:3].reshape(-1)
    T0 = np.ascontiguousarray(np.tile(pose0, (C0.shape[0], 1)))

    def reset(o, img):
        return L.vofe_framestate_reset(o, img.ctypes.data, img.strides[0], P0.ctypes.data, X0.ctypes.data, P0.shape[0], C0.ctypes.data,
                                       F0.ctypes.data, T0.ctypes.data, C0.shape[0])

    xy, st, k, nf, kept = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int(), C.c_int()

    def track(o, img):
        rc = L.vofe_framestate_track(o, img.ctypes.data, img.strides[0], C.byref(xy), C.byref(st), C.byref(k), C.byref(nf), C.byref(kept))
        assert rc == 0, L.vofe_last_error()
        n = k.value
        return (np.ctypeslib.as_array(C.cast(xy, C.POINTER(C.c_float)), shape=(n, 2)).copy(),
                np.ctypeslib.as_array(C.cast(st, C.POINTER(C.c_uint8)), shape=(n,)).copy(), nf.value, kept.value)

    def state(o):
        a, b = C.c_int(), C.c_int()
        assert L.vofe_framestate_counts(o, C.byref(a), C.byref(b)) == 0
        Pq, Cq = np.empty((a.value, 2), np.float32), np.empty((b.value, 2), np.float32)
        Fq, Tq = np.empty((b.value, 2), np.float64), np.empty((b.value, 12), np.float64)
        assert L.vofe_framestate_download(o, Pq.ctypes.data, Cq.ctypes.data, Fq.ctypes.data, Tq.ctypes.data) == 0
        return Pq, Cq, Fq, Tq

    a, b, solo = objs
    # argument errors: no fault, no state change
    assert L.vofe_framestate_track(a, pitched[1].ctypes.data, pitched[1].strides[0], C.byref(xy), C.byref(st), C.byref(k), None, None) == -1   # before reset
    assert L.vofe_framestate_reset(a, None, W, P0.ctypes.data, X0.ctypes.data, 60, None, None, None, 0) == -1
    assert L.vofe_framestate_reset(a, frames[0].ctypes.data, W - 1, P0.ctypes.data, X0.ctypes.data, 60, None, None, None, 0) == -1
    assert L.vofe_framestate_reset(a, frames[0].ctypes.data, W, None, None, 60, None, None, None, 0) == -1
    assert L.vofe_framestate_reset(a, frames[0].ctypes.data, W, P0.ctypes.data, X0.ctypes.data, 5000, None, None, None, 0) == -1
    assert b"capacit" in L.vofe_last_error()
    for o, img in ((a, pitched[0]), (b, frames[0]), (solo, frames[0])):
        assert reset(o, img) == 0, L.vofe_last_error()
    assert L.vofe_framestate_track(a, pitched[1].ctypes.data, W - 4, C.byref(xy), C.byref(st), C.byref(k), None, None) == -1
    assert L.vofe_framestate_update(a, pose0.ctypes.data, None, None) == -1                       # before track
    for t in range(1, 4):
        pose = np.eye(4)[:3]
        pose[:, 3] = [0.03 * t, 0, 0.5 * t]
        pose = np.ascontiguousarray(pose.reshape(-1))
        # a: pitched frames; b: contiguous frames, its calls interleaved with a's; solo: alone, afterwards
        ta = track(a, pitched[t])
        tb = track(b, frames[t])
        assert np.array_equal(ta[0], tb[0]) and np.array_equal(ta[1], tb[1]) and ta[2:] == tb[2:]
        assert L.vofe_framestate_update(b, None, ta[1].ctypes.data, None) == -1                  # NULL pose
        keep = np.ascontiguousarray((ta[1] == 1).astype(np.uint8))
        prom_a, prom_b = C.c_int(), C.c_int()
        assert L.vofe_framestate_update_async(a, pose.ctypes.data, keep.ctypes.data) == 0
        assert L.vofe_framestate_update(b, pose.ctypes.data, keep.ctypes.data, C.byref(prom_b)) == 0
        assert L.vofe_framestate_wait(a, C.byref(prom_a)) == 0 and prom_a.value == prom_b.value
        ts = track(solo, frames[t])
        assert np.array_equal(ts[0], ta[0]) and np.array_equal(ts[1], ta[1])
        assert L.vofe_framestate_update(solo, pose.ctypes.data, keep.ctypes.data, None) == 0
        sa, sb, ss = state(a), state(b), state(solo)
        for x, y, z in zip(sa, sb, ss):
            assert np.array_equal(x, y) and np.array_equal(x, z), t
    assert state(a)[0].shape[0] > 0 and state(a)[1].shape[0] > 50
    for o in objs:
        assert L.vofe_framestate_destroy(o) == 0
    assert L.vofe_framestate_destroy(None) == 0


# ------------------------------------------------------------------ bootstrap front end (initialization.py:41-70)
def _pseudo_ransac_mask(good_new, good_old):
    """Deterministic stand-in for the inlier mask of cv2.findEssentialMat (CPU geometry, out of scope): depends on the
    pair values only, so both sides see the same mask as long as their pairs agree."""
    key = np.floor(good_new[:, 0, 0] * 4) + 3 * np.floor(good_old[:, 0, 1])
    return (key.astype(np.int64) % 9 != 0).astype(np.uint8).reshape(-1, 1)


@pytest.mark.parametrize("dataset,h,w,stride,move", [("kitti", 376, 1241, 2, None), ("parking", 480, 640, 1, None),
                                                     ("malaga", 600, 800, 2, None), ("kitti", 376, 1241, 2, 4.37),
                                                     ("parking", 200, 320, 1, 1.56)])
def test_bootstrap_tracker_follows_the_reference_loop(fe, dataset, h, w, stride, move):
    """Initialization.get_features_and_k up to the essential matrix, with the reference's own bootstrap settings per
    dataset (single-level forward LK, 11x11 / 41x41 windows, minEigThreshold 1e-4, 3 / 7 / 5 images): the device-resident
    twin returns the oracle's pairs after every image, bit for bit; plus thresholds that split the set."""
    from visual_odometry_amd import BOOTSTRAP_PRESETS
    from visual_odometry_amd.bootstrap import BootstrapTracker, bootstrap_correspondences
    params = dict(BOOTSTRAP_PRESETS[dataset])
    if move is not None:
        params["dist_threshold_move"] = move
    n_img = params["n_images"]
    frames = _seq(h, w, 31 + h, (n_img - 1) * stride + 1)[::stride]
    ro, rn, steps = oracle.bootstrap_correspondences(frames, params, _pseudo_ransac_mask)
    bt = BootstrapTracker(h, w, frontend=fe, **params)
    n0 = bt.detect(frames[0])
    assert n0 == oracle.goodFeaturesToTrack(frames[0], params["maxCorners"], 0.01, 10).shape[0]
    for i, img in enumerate(frames[1:]):
        go, gn = bt.track(img)
        assert go.dtype == np.float32 and go.shape == steps[i][0].shape and go.shape[1:] == (1, 2), (i, go.shape, steps[i][0].shape)
        assert np.array_equal(go, steps[i][0]) and np.array_equal(gn, steps[i][1]), "image %d" % (i + 1)
        bt.keep(_pseudo_ransac_mask(gn, go))
    go, gn = bt.pairs()
    assert np.array_equal(go, ro) and np.array_equal(gn, rn)
    assert 10 < len(ro) < n0                                 # the rules removed some pairs and left some
    if move is not None:
        d = np.sqrt(np.sum((steps[0][1] - steps[0][0]) ** 2, axis=2)).flatten()
        assert d.min() >= np.float32(move) and len(steps[0][0]) < n0
    # the one-call form
    fo, fn = bootstrap_correspondences(frames, params, _pseudo_ransac_mask, frontend=fe)
    assert np.array_equal(fo, ro) and np.array_equal(fn, rn)


def test_pair_filter_c_abi_numpy_decisions_nan_in_place(fe):
    """vofe_pair_filter raw: float32 distance decisions of numpy at thresholds that sit on / next to representable
    distances, NaN and infinite coordinates (a NaN distance is not below the threshold: kept), in place and out of place,
    the mask form (threshold 0), n = 0 and argument errors."""
    import ctypes as C
    import torch
    L, ctx = fe.L, fe.ctx
    rng = np.random.default_rng(77)
    for n, thr in ((0, 2.0), (1, 0.0), (63, 5.0), (1024, 2.0), (1025, 3.0000002), (5000, 2.9999998), (3000, -1.0), (3000, float("inf")),
                   (2000, float("nan"))):
        old = np.round(rng.uniform(0, 600, (n, 2))).astype(np.float32)
        step = rng.integers(-4, 5, (n, 2)).astype(np.float32)                 # many distances are exactly 3, 5, sqrt(8) ...
        new = (old + step + (rng.random((n, 2)) < 0.2) * rng.normal(0, 1e-6, (n, 2))).astype(np.float32)
        if n > 10:
            new[3] = np.nan; new[5, 0] = np.inf; old[7, 1] = np.nan; new[9] = old[9]
        flag = rng.integers(0, 3, n).astype(np.uint8)                         # 0, 1, 2: only == 1 passes
        keep = flag == 1
        with np.errstate(invalid="ignore"):
            dist = np.sqrt(np.sum((new[keep].reshape(-1, 1, 2) - old[keep].reshape(-1, 1, 2)) ** 2, axis=2)).flatten()
            drop = np.where(dist < thr)
        r_old, r_new = np.delete(old[keep], drop, axis=0), np.delete(new[keep], drop, axis=0)
        for in_place in (False, True):
            d_new, d_old, d_flag = torch.from_numpy(new).cuda(), torch.from_numpy(old).cuda(), torch.from_numpy(flag).cuda()
            o_new = d_new if in_place else torch.full((max(n, 1), 2), -7.0, dtype=torch.float32, device="cuda")
            o_old = d_old if in_place else torch.full((max(n, 1), 2), -7.0, dtype=torch.float32, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            m = C.c_int(-1)
            rc = L.vofe_pair_filter(ctx, d_new.data_ptr(), d_flag.data_ptr(), d_old.data_ptr(), n, C.c_double(thr),
                                    o_new.data_ptr(), o_old.data_ptr(), cnt.data_ptr(), C.byref(m))
            assert rc == 0 and m.value == len(r_old) == int(cnt.item()), (n, thr, m.value, len(r_old))
            assert np.array_equal(o_new[:m.value].cpu().numpy(), r_new, equal_nan=True), (n, thr, in_place)
            assert np.array_equal(o_old[:m.value].cpu().numpy(), r_old, equal_nan=True), (n, thr, in_place)
            if not in_place and n:
                assert (o_new[m.value:] == -7.0).all() and (o_old[m.value:] == -7.0).all()     # nothing past the count
    # asynchronous form (no host count), then errors
    d = torch.zeros((4, 2), dtype=torch.float32, device="cuda")
    f = torch.ones(4, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert L.vofe_pair_filter(ctx, d.data_ptr(), f.data_ptr(), d.data_ptr(), 4, C.c_double(0.0), d.data_ptr(), d.data_ptr(), cnt.data_ptr(), None) == 0
    fe.synchronize()
    assert int(cnt.item()) == 4
    m = C.c_int(5)
    assert L.vofe_pair_filter(None, d.data_ptr(), f.data_ptr(), d.data_ptr(), 4, C.c_double(0.0), d.data_ptr(), d.data_ptr(), cnt.data_ptr(), C.byref(m)) == -1 and m.value == 0
    assert L.vofe_pair_filter(ctx, d.data_ptr(), f.data_ptr(), d.data_ptr(), -1, C.c_double(0.0), d.data_ptr(), d.data_ptr(), cnt.data_ptr(), None) == -1
    assert L.vofe_pair_filter(ctx, d.data_ptr(), f.data_ptr(), d.data_ptr(), 4, C.c_double(0.0), d.data_ptr(), d.data_ptr(), None, None) == -1
    assert L.vofe_pair_filter(ctx, None, f.data_ptr(), d.data_ptr(), 4, C.c_double(0.0), d.data_ptr(), d.data_ptr(), cnt.data_ptr(), None) == -1
    assert b"vofe_pair_filter" in L.vofe_last_error()


def test_bootstrap_tracker_device_frames_and_restart(fe):
    """Frames that are already on the device (torch tensors) take the copy path instead of the staged upload; detect()
    on a tracker that has been used starts a new bootstrap: both give the oracle's pairs."""
    import torch
    from visual_odometry_amd import BOOTSTRAP_PRESETS
    from visual_odometry_amd.bootstrap import BootstrapTracker
    params = dict(BOOTSTRAP_PRESETS["parking"])
    frames = _seq(240, 320, 77, 4)
    keep_all = lambda gn, go: np.ones((len(gn), 1), np.uint8)             # noqa: E731
    ro, rn, steps = oracle.bootstrap_correspondences(frames[1:], params, keep_all)
    bt = BootstrapTracker(240, 320, frontend=fe, **params)
    bt.detect(frames[0])
    bt.track(frames[3])                                                      # some other bootstrap first
    dev = [torch.from_numpy(f).cuda() for f in frames]
    bt.detect(dev[1])
    for i in (2, 3):
        go, gn = bt.track(dev[i])
        assert np.array_equal(go, steps[i - 2][0]) and np.array_equal(gn, steps[i - 2][1]), i
        bt.keep(keep_all(gn, go))
    go, gn = bt.pairs()
    assert np.array_equal(go, ro) and np.array_equal(gn, rn) and len(ro) > 50


def test_bootstrap_tracker_argument_and_order_errors(fe):
    from visual_odometry_amd import error
    from visual_odometry_amd.bootstrap import BootstrapTracker
    a, b = _seq(97, 131, 5, 2)
    bt = BootstrapTracker(97, 131, frontend=fe, maxCorners=50)
    with pytest.raises(error):
        bt.track(b)                                          # before detect
    with pytest.raises(error):
        bt.detect(np.zeros((97, 131), np.uint8))             # no corner: the reference fails there as well
    assert bt.detect(a) > 0
    with pytest.raises(error):
        bt.track(b[:50])                                     # wrong size
    go, gn = bt.track(b)
    with pytest.raises(error):
        bt.keep(np.ones(len(go) + 1, np.uint8))
    assert bt.keep(np.zeros((len(go), 1), np.uint8)) == 0    # RANSAC kept nothing
    with pytest.raises(error):
        bt.track(a)                                          # cv2 raises on an empty prevPts
    with pytest.raises(error):
        BootstrapTracker(97, 131, frontend=fe, maxCorners=0)


# ------------------------------------------------------------------ a C host on the C ABI
def test_c_host_program_runs_the_frame_loop_against_ground_truth():
    """examples/frame_loop.c: a C99 program that links libvofe.so and nothing else drives vofe_framestate over 23 frames of
    a textured plane under a known camera translation; it checks the tracked flows against the known image motion and
    the depth of every promoted candidate against the plane, and prints 'frame_loop ok'."""
    import os
    import subprocess
    import __graft_entry__ as g
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "frame_loop")
    if not os.path.exists(exe):
        exe = g.build_example()                       # (gcc, one second: not a HIP build)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0 and "frame_loop ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("promoted")][0]
    assert int(last.split()[1]) >= 50 and " 0 off the plane" in last, last


# ------------------------------------------------------------------ real textures (load-if-present)
_REAL_DIR = "/opt/conda/lib/python3.9/site-packages/skimage/data"


def _real(name):
    import os
    path = os.path.join(_REAL_DIR, name)
    if not os.path.exists(path):
        pytest.skip("%s is not in this image" % path)
    from PIL import Image
    return np.array(Image.open(path).convert("L"), dtype=np.uint8)


@pytest.mark.parametrize("name", ["camera.png", "brick.png", "coins.png", "text.png", "chessboard_GRAY.png", "moon.png"])
def test_real_texture_detector_and_pyramid(fe, name):
    """Photographs, scanned text (saturated 0/255 runs) and a chessboard (exact score ties): corner lists, score map
    and pyramid identical to the oracle."""
    import torch
    from visual_odometry_amd.frontend import Pyramid
    img = _real(name)
    h, w = img.shape
    assert np.array_equal(fe.corner_score(img), oracle.corner_score(img))
    for maxc, q, md in ((1000, 0.01, 10), (0, 0.05, 3), (200, 0.01, 0)):
        ref = oracle.goodFeaturesToTrack(img, maxc, q, md)
        got = fe.goodFeaturesToTrack(img, maxc, q, md)
        assert (ref is None) == (got is None), (name, maxc, q, md)
        if ref is not None:
            assert got.shape == ref.shape and np.array_equal(got, ref), (name, maxc, q, md)
    pyr = Pyramid(fe, 1, h, w, (21, 21), 3, True)
    pyr.build(torch.from_numpy(img).cuda(), 0)
    for lvl in range(pyr.levels):
        ref = oracle.pyramid_level(img, lvl)
        got, der = pyr.download(0, lvl, deriv=True)
        assert np.array_equal(got, ref) and np.array_equal(der, oracle.scharr(ref)), (name, lvl)
    pyr.free()


def test_real_stereo_pair_tracking(fe):
    """motorcycle_left/right: a real image pair with large horizontal disparities, occlusions and specular surfaces, so
    every exit of the tracker is taken (lost at a coarse level, left the image, failed forward-backward check)."""
    a, b = _real("motorcycle_left.png"), _real("motorcycle_right.png")
    assert a.shape == b.shape
    pts = oracle.goodFeaturesToTrack(a, 3000, 0.01, 7).reshape(-1, 2)
    for win, lvl in (((21, 21), 3), ((31, 31), 4), ((11, 11), 2)):
        rn, rs, re = oracle.calcOpticalFlowPyrLK(a, b, pts, winSize=win, maxLevel=lvl, criteria=(3, 30, 0.01), minEigThreshold=1e-3)
        gn, gs, ge = fe.calcOpticalFlowPyrLK(a, b, pts, winSize=win, maxLevel=lvl, criteria=(3, 30, 0.01), minEigThreshold=1e-3)
        assert np.array_equal(gs, rs) and 0.2 < rs.mean() < 1.0, (win, rs.mean())
        assert np.array_equal(gn, rn) and np.array_equal(ge, re), win
    rt, rst, rerr = oracle.track_keypoints(pts.T, a, b)
    gt, gst, gerr = fe.track_keypoints(pts.T, a, b)
    assert np.array_equal(gst, rst) and np.array_equal(gt, rt) and np.array_equal(gerr, rerr)
    assert 0 < rst.sum() < rs.sum()                      # the forward-backward check rejects some forward successes


# ------------------------------------------------------------------ seeded randomised differential sweeps
def test_randomised_lk_sweep_all_window_shapes(fe):
    """40 random (image size, window, levels, criteria, flags) combinations: every row-group template of the
    run-time-geometry kernel (windows 3..64 wide, 3..64 tall) and, on every eighth case, a window beyond them (the
    workgroup-per-keypoint fallback) against the oracle, bit for bit."""
    import os
    rng = np.random.default_rng(int(os.environ.get("VOFE_SWEEP_SEED", 2024)))       # soak runs: other seeds, more cases
    seen_r = set()
    for case in range(int(os.environ.get("VOFE_SWEEP_CASES", 40))):
        h, w = int(rng.integers(48, 260)), int(rng.integers(48, 320))
        ww, wh = int(rng.integers(3, 65)), int(rng.integers(3, 65))
        if case < 10:                                   # make sure the extremes are in
            ww, wh = [(3, 3), (64, 64), (64, 3), (3, 64), (5, 41), (33, 7), (21, 22), (22, 21), (13, 13), (47, 31)][case]
        elif case % 8 == 5:                             # beyond the wavefront kernels: the workgroup-per-keypoint fallback
            ww, wh = [(int(rng.integers(65, 104)), int(rng.integers(3, 100))), (int(rng.integers(33, 64)), int(rng.integers(65, 120)))][case % 16 == 5]
        lvl = int(rng.integers(0, 4))
        a, b = _seq(h, w, 300 + case)
        pts = np.concatenate([(rng.random((40, 2)) * [w - 1, h - 1]), rng.normal(0, 30, (6, 2)) + [w, h]]).astype(np.float32)
        crit = [(3, 30, 0.01), (1, 7, 0.0), (2, 0, 0.05), (3, 12, 0.2)][case % 4]
        flags = [0, 8, 4, 0][case % 4]
        init = (pts + rng.normal(0, 1.5, pts.shape)).astype(np.float32) if flags & 4 else None
        mineig = [1e-4, 1e-3, 1e-2][case % 3]
        n0, s0, e0 = oracle.calcOpticalFlowPyrLK(a, b, pts, init, (ww, wh), lvl, crit, flags, mineig, nthreads=8)
        n1, s1, e1 = fe.calcOpticalFlowPyrLK(a, b, pts, init, (ww, wh), lvl, crit, flags, mineig)
        assert np.array_equal(s0, s1), (case, h, w, ww, wh, lvl)
        assert np.array_equal(n0, n1, equal_nan=True) and np.array_equal(e0, e1, equal_nan=True), (case, h, w, ww, wh, lvl)
        if ww <= 64:
            groups = min(64 // ww, wh)
            seen_r.add(-(-wh // groups))
    assert len(seen_r) >= 12


def test_randomised_lk_21x21_sweep_all_kernels(fe, knob):
    """Random frames, point sets (interior, border, far outside, NaN), criteria and flags through the three 21x21
    kernels (one, two and four keypoints per wavefront) and the tracker glue: identical to the oracle, bit for bit."""
    import os
    rng = np.random.default_rng(int(os.environ.get("VOFE_SWEEP_SEED", 4242)))
    for case in range(int(os.environ.get("VOFE_SWEEP_CASES", 40)) // 2):
        h, w = int(rng.integers(40, 300)), int(rng.integers(40, 420))
        lvl = int(rng.integers(0, 5))
        a, b = _seq(h, w, 900 + case)
        n = int(rng.integers(1, 400))
        pts = np.concatenate([(rng.random((n, 2)) * [w - 1, h - 1]), rng.normal(0, 25, (8, 2)) + [[0, 0], [w, h], [w, 0], [0, h]] * 2])
        pts = pts.astype(np.float32)
        if case % 9 == 0:
            pts[0] = [np.nan, 5]
            pts[-1] = [1e9, -1e9]
        crit = [(3, 30, 0.01), (1, 3, 0.0), (2, 0, 0.03), (3, 50, 1e-4)][case % 4]
        flags = [0, 0, 4, 8][case % 4]
        init = (pts + rng.normal(0, 2.0, pts.shape)).astype(np.float32) if flags & 4 else None
        mineig = [1e-4, 1e-3, 5e-3][case % 3]
        with np.errstate(all="ignore"):
            n0, s0, e0 = oracle.calcOpticalFlowPyrLK(a, b, pts, init, (21, 21), lvl, crit, flags, mineig, nthreads=8)
            t0 = oracle.track_keypoints(pts.T, a, b, (21, 21), lvl, crit, mineig, 1.0, 8)
        for kern in ("x1", "lp", "x2", "x2p", "x4"):
            knob("VOFE_LK", kern)
            n1, s1, e1 = fe.calcOpticalFlowPyrLK(a, b, pts, init, (21, 21), lvl, crit, flags, mineig)
            assert np.array_equal(s0, s1), (case, kern, h, w, lvl)
            assert np.array_equal(n0, n1, equal_nan=True) and np.array_equal(e0, e1, equal_nan=True), (case, kern, h, w, lvl)
            t1 = fe.track_keypoints(pts.T, a, b, (21, 21), lvl, crit, mineig)
            for x, y in zip(t0, t1):
                assert np.array_equal(x, y, equal_nan=True), (case, kern, "fb")
        knob("VOFE_LK")


def test_randomised_pyramid_sweep_all_build_paths(fe, knob):
    """Random frame sizes (ragged last groups of 8 and 16, levels that end early, levels barely larger than the window),
    windows and level counts through every build path -- fused launch, per-level kernels as the level pipeline, as the plain
    sequence with the pyrDown kernel's own border bands, and with the border launch -- on random, not band-limited content:
    every level, its Scharr plane and its REFLECT_101 border, bit for bit, for a batch of three images."""
    import os
    import torch
    from visual_odometry_amd.frontend import Pyramid
    rng = np.random.default_rng(int(os.environ.get("VOFE_SWEEP_SEED", 31337)))
    paths = ({"VOFE_PYRAMID": "fused"}, {"VOFE_PYRAMID": "levels"}, {"VOFE_PYRAMID": "levels", "VOFE_PYR_STEP": "0"},
             {"VOFE_PYRAMID": "levels", "VOFE_PYR_BORDERS": "separate"}, {"VOFE_PYRAMID": "fused", "VOFE_PYR_BORDERS": "separate"},
             {"VOFE_PYRAMID": "levels", "VOFE_PYR_TAIL": "0"}, {"VOFE_PYRAMID": "levels", "VOFE_PYR_TAIL_TH": "8"})
    names = ("VOFE_PYRAMID", "VOFE_PYR_BORDERS", "VOFE_PYR_STEP", "VOFE_PYR_TAIL", "VOFE_PYR_TAIL_TH")
    for case in range(int(os.environ.get("VOFE_SWEEP_CASES", 40))):
        win = (int(rng.integers(3, 42)), int(rng.integers(3, 42)))
        h = int(rng.integers(win[1] + 1, 260))
        w = int(rng.integers(max(win[0] + 1, 8), 420))
        if case % 5 == 4:                                # frames of several tail tiles (64 x 16 top-level pixels each)
            h, w = int(rng.integers(200, 520)), int(rng.integers(400, 1100))
        if case < 8:                                     # widths around the group sizes, with the reference's window
            win = (21, 21)
            w = (63, 64, 65, 71, 72, 73, 127, 129)[case]
            h = (44, 45, 46, 47, 48, 49, 50, 51)[case]
        lvl = int(rng.integers(0, 5))
        imgs = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
        d = torch.from_numpy(imgs).cuda()
        sizes = oracle.pyramid_sizes(h, w, win, lvl)
        refs = [[oracle.pyramid_level(imgs[k], q) for q in range(len(sizes))] for k in range(3)]
        path = paths[case % len(paths)]
        for name in names:
            if name in path:
                knob(name, path[name])
            else:
                knob(name)
        pyr = Pyramid(fe, 3, h, w, win, lvl, True)
        pyr.build(d, 0)
        assert pyr.levels == len(sizes), (case, h, w, win, lvl)
        for k in range(3):
            for q in range(pyr.levels):
                got, der = pyr.download(k, q, deriv=True)
                what = (case, sorted(path.items()), h, w, win, lvl, k, q)
                assert np.array_equal(got, refs[k][q]), what
                assert np.array_equal(der, oracle.scharr(refs[k][q])), what
                pad = pyr.download_padded(k, q)
                assert np.array_equal(pad, np.pad(refs[k][q], ((win[1], win[1]), (win[0], win[0])), mode="reflect")), what
        pyr.free()
    for name in names:
        knob(name)


def test_randomised_gftt_sweep(fe):
    """40 random detector configurations (size, quality, minDistance incl. 0 and large, maxCorners, mask, Harris)."""
    import os
    rng = np.random.default_rng(int(os.environ.get("VOFE_SWEEP_SEED", 77)))
    for case in range(int(os.environ.get("VOFE_SWEEP_CASES", 40))):
        h, w = int(rng.integers(20, 300)), int(rng.integers(20, 400))
        a = _seq(h, w, 500 + case, 1)[0]
        if case % 7 == 0:
            a = (a // 32 * 32).astype(np.uint8)                       # posterised: plateaus and ties
        q = float(rng.choice([0.001, 0.01, 0.05, 0.3, 0.9]))
        md = float(rng.choice([0, 0.7, 1, 2.5, 7, 10, 19, 23.5, 40]))
        maxc = int(rng.choice([0, 1, 17, 200, 5000]))
        harris = bool(case % 5 == 0)
        mask = None
        if case % 3 == 0:
            mask = (rng.random((h, w)) > 0.4).astype(np.uint8)
        bs, ks = 3, 3
        if case % 4 == 1:                                             # cv2's optional blockSize / gradientSize
            bs, ks = int(rng.integers(1, 12)), int(rng.choice([1, 3, 5, 7]))
        got = fe.goodFeaturesToTrack(a, maxc, q, md, mask=mask, blockSize=bs, useHarrisDetector=harris, k=0.04, gradientSize=ks)
        ref = oracle.goodFeaturesToTrack(a, maxc, q, md, mask=mask, blockSize=bs, useHarrisDetector=harris, k=0.04, gradientSize=ks)
        assert (got is None) == (ref is None), (case, h, w, q, md, maxc, bs, ks)
        if ref is not None:
            assert np.array_equal(got, ref), (case, h, w, q, md, maxc, harris, bs, ks)


def test_randomised_score_sweep(fe, knob):
    """Score maps of random frames, bit for bit: sizes around the strip kernel's internal boundaries (4-column
    threads, 256-column waves, strip heights), every strip height forced in turn, smooth / noisy / posterised /
    saturated content, Shi-Tomasi and Harris."""
    import os
    rng = np.random.default_rng(int(os.environ.get("VOFE_SWEEP_SEED", 99)))
    for case in range(int(os.environ.get("VOFE_SWEEP_CASES", 40))):
        h = int(rng.choice([4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 33, 64, 65, int(rng.integers(4, 200))]))
        w = int(rng.choice([8, 9, 10, 11, 12, 13, 255, 256, 257, 258, 259, 260, 261, 511, 513, int(rng.integers(8, 700))]))
        kind = case % 4
        if kind == 0:
            a = _seq(h, w, 900 + case, 1)[0]
        elif kind == 1:
            a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 2:
            a = (_seq(h, w, 900 + case, 1)[0] // 64 * 64).astype(np.uint8)
        else:
            a = np.where(rng.random((h, w)) > 0.5, 255, 0).astype(np.uint8)
            a[: h // 2] = 255                                          # large flat part: zero scores, zero sqrt arguments
        rs = ("2", "4", "8", "16", None)[case % 5]
        if rs is None:
            knob("VOFE_SCORE_RS")
        else:
            knob("VOFE_SCORE_RS", rs)
        harris = bool(case % 3 == 0)
        assert np.array_equal(fe.corner_score(a, harris, 0.04), oracle.corner_score(a, harris, 0.04)), (case, h, w, kind, rs, harris)
