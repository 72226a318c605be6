"""ctypes front for the CPU oracle (oracle/vofe_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package; the
product package (visual-odometry_amd) never does.  PARITY UNPINNED against cv2: see
vofe_oracle.h.  Signatures mirror the cv2 calls made at reference continuous.py:145-151,
continuous.py:380-385 and initialization.py:41-52.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TERM_CRITERIA_COUNT = 1
TERM_CRITERIA_EPS = 2
OPTFLOW_USE_INITIAL_FLOW = 4
OPTFLOW_LK_GET_MIN_EIGENVALS = 8


def build(force=False):
    """Compile libvofe_oracle.so with the committed Makefile (g++ only)."""
    so = os.path.join(_HERE, "libvofe_oracle.so")
    src = os.path.join(_HERE, "vofe_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libvofe_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libvofe_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        u8p, f32p, i32p, i16p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int16))
        L.vofe_cpu_pyr_down_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
        L.vofe_cpu_scharr_s16x2.argtypes = [u8p, C.c_int, C.c_int, C.c_int, i16p, C.c_int]
        L.vofe_cpu_pyramid_levels.argtypes = [C.c_int] * 5 + [i32p, i32p]
        L.vofe_cpu_pyramid_level.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.vofe_cpu_lk_track.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, f32p, f32p, u8p, f32p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                        C.c_int, C.c_double, C.c_int]
        L.vofe_cpu_track_fb.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, f32p, f32p, u8p, f32p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_float, C.c_int]
        L.vofe_cpu_corner_score.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, f32p, C.c_int]
        L.vofe_cpu_gftt.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, u8p,
                                    C.c_int, C.c_double, f32p, C.c_int, i32p, C.c_int]
        L.vofe_cpu_corner_score_ex.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, f32p,
                                               C.c_int]
        L.vofe_cpu_corner_score_generic.argtypes = L.vofe_cpu_corner_score_ex.argtypes
        L.vofe_cpu_gftt_ex.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, u8p,
                                       C.c_int, C.c_int, C.c_int, C.c_double, f32p, C.c_int, i32p, C.c_int]
        L.vofe_cpu_set_variant.argtypes = [C.c_int] * 6
        L.vofe_cpu_set_simd.argtypes = [C.c_int]
        _LIB = L
    return _LIB


# Alternative arithmetics for the cv2-build dependent points of SURVEY Appendix A (vofe_oracle.h, vofe_cpu_set_variant).
# name -> keyword arguments of variant(); "canonical" is what the HIP kernels follow.
LK_VARIANTS = {
    "f32_raster": dict(lk_sum=1), "f32_lane4": dict(lk_sum=2), "f32_lane4_fma": dict(lk_sum=3),
    "f32_ui128": dict(lk_sum=4), "f32_ui128_fma": dict(lk_sum=5), "f32_lane8": dict(lk_sum=6),
    "f32_lane8_fma": dict(lk_sum=7),
}
SCORE_VARIANTS = {
    "sobel_fma": dict(sobel=1), "sobel_scale_after": dict(sobel=2), "sobel_row_taps": dict(sobel=3),
    "mineig_fma": dict(mineig_fma=1),
    "box_running": dict(box_running=1), "thresh_ipp": dict(thresh_ipp=1),
    "all_fma": dict(sobel=1, mineig_fma=1, box_running=1, thresh_ipp=1),
}
HARRIS_VARIANTS = {"harris_scalar_f64": dict(harris=1), "harris_vector_f32": dict(harris=2)}


class simd:
    """with oracle.simd(): ...  -- the tracker's window loops vectorised (AVX2), bit-identical to the scalar restatement.  Used
    by bench.py's cpu_baseline leg so that the CPU number beside the GPU's is not a scalar loop.  Process-wide."""

    def __enter__(self):
        lib().vofe_cpu_set_simd(1)
        return self

    def __exit__(self, *exc):
        lib().vofe_cpu_set_simd(0)
        return False


class variant:
    """with oracle.variant(lk_sum=1): ...   -- runs the oracle with one of the alternative arithmetics, then restores the
    canonical one.  Process-wide (the C++ side keeps it in a global): not for concurrent use."""

    def __init__(self, lk_sum=0, sobel=0, mineig_fma=0, box_running=0, harris=0, thresh_ipp=0):
        self.args = (lk_sum, sobel, mineig_fma, box_running, harris, thresh_ipp)

    def __enter__(self):
        if lib().vofe_cpu_set_variant(*self.args):
            raise ValueError("unknown oracle variant %r" % (self.args,))
        return self

    def __exit__(self, *exc):
        lib().vofe_cpu_set_variant(0, 0, 0, 0, 0, 0)
        return False


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _img(a):
    a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim != 2:
        raise ValueError("image must be a 2-D uint8 array")
    return np.ascontiguousarray(a)


def pyr_down(img):
    img = _img(img)
    h, w = img.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    rc = lib().vofe_cpu_pyr_down_u8(_p(img, C.c_uint8), h, w, w, _p(out, C.c_uint8), out.shape[1])
    assert rc == 0
    return out


def scharr(img):
    img = _img(img)
    h, w = img.shape
    out = np.empty((h, w, 2), np.int16)
    rc = lib().vofe_cpu_scharr_s16x2(_p(img, C.c_uint8), h, w, w, _p(out, C.c_int16), 2 * w)
    assert rc == 0
    return out


def pyramid_sizes(h, w, win, max_level):
    lw = (C.c_int * (max_level + 1))()
    lh = (C.c_int * (max_level + 1))()
    n = lib().vofe_cpu_pyramid_levels(h, w, win[0], win[1], max_level, lw, lh)
    return [(lh[i], lw[i]) for i in range(n)]


def pyramid_level(img, level):
    img = _img(img)
    h, w = img.shape
    for _ in range(level):
        h, w = (h + 1) // 2, (w + 1) // 2
    out = np.empty((h, w), np.uint8)
    rc = lib().vofe_cpu_pyramid_level(_p(img, C.c_uint8), img.shape[0], img.shape[1], img.shape[1], level,
                                      _p(out, C.c_uint8))
    assert rc == 0
    return out


def _pts(p):
    p = np.ascontiguousarray(np.asarray(p, np.float32).reshape(-1, 2))
    return p


def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts=None, winSize=(21, 21), maxLevel=3,
                         criteria=(TERM_CRITERIA_COUNT | TERM_CRITERIA_EPS, 30, 0.01), flags=0,
                         minEigThreshold=1e-4, nthreads=1):
    prevImg, nextImg = _img(prevImg), _img(nextImg)
    if prevImg.shape != nextImg.shape:
        raise ValueError("images must have the same size")
    p = _pts(prevPts)
    n = p.shape[0]
    if n == 0:
        return None, None, None
    nxt = _pts(nextPts).copy() if (flags & OPTFLOW_USE_INITIAL_FLOW) else np.zeros_like(p)
    st = np.empty(n, np.uint8)
    err = np.empty(n, np.float32)
    h, w = prevImg.shape
    rc = lib().vofe_cpu_lk_track(_p(prevImg, C.c_uint8), _p(nextImg, C.c_uint8), h, w, w, _p(p, C.c_float),
                                 _p(nxt, C.c_float), _p(st, C.c_uint8), _p(err, C.c_float), n,
                                 int(winSize[0]), int(winSize[1]), int(maxLevel), int(criteria[0]),
                                 int(criteria[1]), float(criteria[2]), int(flags), float(minEigThreshold),
                                 int(nthreads))
    if rc:
        raise ValueError("vofe_cpu_lk_track: bad arguments (rc=%d)" % rc)
    shape = np.shape(prevPts)                      # cv2: nextPts takes prevPts' own layout, status / err are (N,1)
    return (nxt.reshape(shape if len(shape) in (2, 3) and shape[-1] == 2 else (-1, 1, 2)), st.reshape(-1, 1),
            err.reshape(-1, 1))


def track_keypoints(prev_keypoints, img_prev, img_current, winSize=(21, 21), maxLevel=3,
                    criteria=(3, 30, 0.01), minEigThreshold=1e-3, fb_thresh=1.0, nthreads=1):
    """Twin of ContinuousVO._track_keypoints (reference continuous.py:116-164)."""
    prev_keypoints = np.asarray(prev_keypoints)
    if prev_keypoints.size == 0:
        return np.empty((2, 0)), np.empty((0,), np.uint8), np.empty((0,), np.float32)
    img_prev, img_current = _img(img_prev), _img(img_current)
    p = np.ascontiguousarray(prev_keypoints.T.reshape(-1, 2).astype(np.float32))
    n = p.shape[0]
    nxt = np.zeros_like(p)
    st = np.empty(n, np.uint8)
    err = np.empty(n, np.float32)
    h, w = img_prev.shape
    rc = lib().vofe_cpu_track_fb(_p(img_prev, C.c_uint8), _p(img_current, C.c_uint8), h, w, w, _p(p, C.c_float),
                                 _p(nxt, C.c_float), _p(st, C.c_uint8), _p(err, C.c_float), n,
                                 int(winSize[0]), int(winSize[1]), int(maxLevel), int(criteria[0]),
                                 int(criteria[1]), float(criteria[2]), float(minEigThreshold),
                                 float(fb_thresh), int(nthreads))
    if rc:
        raise ValueError("vofe_cpu_track_fb: bad arguments (rc=%d)" % rc)
    return nxt.T, st, err


def corner_score(img, harris=False, k=0.04, nthreads=1, blockSize=3, ksize=3, force_generic=False):
    """cornerMinEigenVal / cornerHarris score map.  blockSize 1..31, ksize (Sobel aperture) 1, 3, 5 or 7; block 3 / aperture 3
    (what the reference uses) takes the dedicated restatement, everything else the generic one (`force_generic`: also 3 / 3)."""
    img = _img(img)
    h, w = img.shape
    out = np.empty((h, w), np.float32)
    fn = lib().vofe_cpu_corner_score_generic if force_generic else lib().vofe_cpu_corner_score_ex
    rc = fn(_p(img, C.c_uint8), h, w, w, int(blockSize), int(ksize), int(bool(harris)), float(k), _p(out, C.c_float),
            int(nthreads))
    if rc:
        raise ValueError("vofe_cpu_corner_score: bad arguments (rc=%d)" % rc)
    return out


def goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, mask=None, blockSize=3,
                        useHarrisDetector=False, k=0.04, nthreads=1, gradientSize=3):
    img = _img(image)
    h, w = img.shape
    cap = maxCorners if maxCorners > 0 else h * w
    out = np.empty((max(cap, 1), 2), np.float32)
    n = C.c_int(0)
    m = None
    if mask is not None:
        m = np.ascontiguousarray(np.asarray(mask, np.uint8))
        if m.shape != img.shape:
            raise ValueError("mask must match the image size")
    rc = lib().vofe_cpu_gftt_ex(_p(img, C.c_uint8), h, w, w, int(maxCorners), float(qualityLevel),
                                float(minDistance), _p(m, C.c_uint8) if m is not None else None,
                                int(blockSize), int(gradientSize), int(bool(useHarrisDetector)), float(k),
                                _p(out, C.c_float), cap, C.byref(n), int(nthreads))
    if rc:
        raise ValueError("vofe_cpu_gftt: bad arguments (rc=%d)" % rc)
    if n.value == 0:
        return None
    return out[: n.value].reshape(-1, 1, 2).copy()


def new_feature_mask(features_current, existing_features, threshold=15):
    """Restatement of reference continuous.py:408-413 (scipy.spatial.distance.cdist, float64 Euclidean,
    then np.min(axis=1) > threshold) in plain numpy: d = sqrt(dx*dx + dy*dy) per pair, one rounding per
    operation.  PINNED: tests/test_oracle_cpu.py checks it against scipy's cdist itself."""
    f = np.asarray(features_current, dtype=np.float64).reshape(-1, 2)
    e = np.asarray(existing_features, dtype=np.float64).reshape(-1, 2)
    dx = f[:, None, 0] - e[None, :, 0]
    dy = f[:, None, 1] - e[None, :, 1]
    d = np.sqrt(dx * dx + dy * dy)
    return np.min(d, axis=1) > threshold


def filter_candidates(tracked_features, status, Fi, Ti, dist_min=2, dist_max=100, dist_triangulation=15):
    """Reference continuous.py:391-405 and :446-448, the reference's own numpy expressions on the same arrays (numpy
    IS the reference's arithmetic for these lines, so this restatement is pinned by construction): compaction by
    status, travel-distance band, triangulation flag.  Returns (Ci, Fi, Ti, new_landmarks)."""
    valid_idx = np.where(status == 1)[0]
    Ci = tracked_features[:, valid_idx]
    Fi = Fi[:, valid_idx]
    Ti = Ti[:, valid_idx]
    distances = np.sqrt(np.sum((Ci - Fi) ** 2, axis=0))
    good_candidates = (distances > dist_min) & (distances < dist_max)
    Ci = Ci[:, good_candidates]
    Fi = Fi[:, good_candidates]
    Ti = Ti[:, good_candidates]
    distance_matrix = np.sqrt(np.sum((Ci - Fi) ** 2, axis=0))
    new_landmarks = distance_matrix > dist_triangulation
    return Ci, Fi, Ti, new_landmarks


def _triangulate_points(proj1, proj2, pt1, pt2):
    """cv2.triangulatePoints for one point pair as OpenCV defines it (modules/calib3d/src/triangulate.cpp): the 4x4
    DLT matrix rows x*P[2]-P[0], y*P[2]-P[1] per view in float64, right singular vector of the smallest singular
    value, stored in the type of the first point array.  The SVD here is LAPACK's (numpy), OpenCV's is a one-sided
    Jacobi: equal up to rounding error and the sign of the vector, which the division by X4[3] removes.  cv2 is not
    importable here, so this one function is PARITY UNPINNED."""
    A = np.empty((4, 4))
    for j, (P, pt) in enumerate(((proj1, pt1), (proj2, pt2))):
        x, y = float(pt[0]), float(pt[1])
        A[2 * j] = x * P[2] - P[0]
        A[2 * j + 1] = y * P[2] - P[1]
    vt = np.linalg.svd(A)[2]
    return vt[3].reshape(4, 1).astype(np.asarray(pt1).dtype)


def _invert_transformation(T_3x4):
    """continuous.py:497-522."""
    T_4x4 = np.eye(4)
    T_4x4[:3, :4] = T_3x4
    R = T_4x4[:3, :3]
    t = T_4x4[:3, 3]
    R_inv = R.T
    t_inv = -R_inv @ t
    T_4x4_inv = np.eye(4)
    T_4x4_inv[:3, :3] = R_inv
    T_4x4_inv[:3, 3] = t_inv
    return T_4x4_inv[:3, :4]


def _compute_baseline_angle(point3D, t_cur, t_first):
    """continuous.py:274-309."""
    point3D, t_cur, t_first = point3D.flatten(), t_cur.flatten(), t_first.flatten()
    v_cur = point3D - t_cur
    v_first = point3D - t_first
    v_cur = v_cur / np.linalg.norm(v_cur)
    v_first = v_first / np.linalg.norm(v_first)
    cos_angle = np.clip(np.dot(v_cur, v_first), -1.0, 1.0)
    return float(np.arccos(cos_angle))


def triangulate_candidates(K, pose, Ci, Fi, Ti, new_landmarks, alpha_deg=1, margins=None):
    """Reference continuous.py:450-485: the per-candidate loop, with _triangulate_points standing in for
    cv2.triangulatePoints.  Returns (new_landmarks, points3D (3,P)).  `margins`, when a list, receives per flagged
    candidate how far its two decisions are from their thresholds (depth margin, angle margin in radians) so tests
    can leave out candidates whose decision hangs on SVD rounding."""
    new_landmarks = np.array(new_landmarks, dtype=bool).reshape(-1).copy()
    K = np.asarray(K, dtype=np.float64).reshape(3, 3)
    pose = np.asarray(pose, dtype=np.float64)
    t_cur = pose[:3, 3]
    points3D = np.empty((3, 0))
    for idx in range(new_landmarks.shape[0]):
        if not new_landmarks[idx]:
            continue
        Ti_idx_formatted = np.reshape(Ti[:12, idx], (3, 4))
        proj1 = K @ _invert_transformation(Ti_idx_formatted)
        proj2 = K @ _invert_transformation(pose[:3, :])
        point4D_hom = _triangulate_points(proj1, proj2, Fi[:, idx], Ci[:, idx])
        point3D = point4D_hom[:3] / point4D_hom[3]
        t_idx = Ti_idx_formatted[:3, 3]
        alpha = _compute_baseline_angle(point3D, t_cur, t_idx)
        if margins is not None:
            margins.append((idx, float(point3D[2, 0]) + t_idx[2], abs(alpha) - np.deg2rad(alpha_deg)))
        if point3D[2] < -t_idx[2]:
            new_landmarks[idx] = False
            continue
        if abs(alpha) > np.deg2rad(alpha_deg):
            points3D = np.hstack([points3D, point3D])
        else:
            new_landmarks[idx] = False
    return new_landmarks, points3D


def handle_candidates(K, img_current, img_prev, Ci, Fi, Ti, Pi, Xi, pose, max_corners=1000, quality_level=0.01,
                      min_distance=10, lk_params=None, threshold_new_keypoints=15, threshold_triangulation=15,
                      threshold_candidates_min=2, threshold_candidates_max=100, threshold_alpha_deg=1):
    """Reference ContinuousVO._handle_candidates (continuous.py:375-495) line by line, with the oracle's restatements
    where the reference calls cv2 (goodFeaturesToTrack, calcOpticalFlowPyrLK inside _track_keypoints,
    triangulatePoints) and scipy's own cdist where it calls cdist."""
    from scipy.spatial.distance import cdist
    features_current = goodFeaturesToTrack(img_current, maxCorners=max_corners, qualityLevel=quality_level,
                                           minDistance=min_distance)
    features_current = np.float32(features_current).reshape(-1, 2)
    tracked_features, status, _ = track_keypoints(Ci, img_prev, img_current, **(lk_params or {}))
    valid_idx = np.where(status == 1)[0]
    Ci = tracked_features[:, valid_idx]
    Fi = Fi[:, valid_idx]
    Ti = Ti[:, valid_idx]
    distances = np.sqrt(np.sum((Ci - Fi) ** 2, axis=0))
    good_candidates = (distances > threshold_candidates_min) & (distances < threshold_candidates_max)
    Ci = Ci[:, good_candidates]
    Fi = Fi[:, good_candidates]
    Ti = Ti[:, good_candidates]
    existing_features = np.vstack([Ci.T, Pi.T])
    distance_matrix = cdist(features_current, existing_features)
    min_distances = np.min(distance_matrix, axis=1)
    new_feature_mask = min_distances > threshold_new_keypoints
    new_features = features_current[new_feature_mask].T
    distance_matrix = np.sqrt(np.sum((Ci - Fi) ** 2, axis=0))
    new_landmarks = distance_matrix > threshold_triangulation
    if new_landmarks.sum() > 0:
        new_landmarks, points3D = triangulate_candidates(K, pose, Ci, Fi, Ti, new_landmarks, threshold_alpha_deg)
        Pi = np.hstack([Pi, Ci[:, new_landmarks]])
        Xi = np.hstack([Xi, points3D])
        no_new_landmark = np.where(new_landmarks == 0)[0]
        Ci = Ci[:, no_new_landmark]
        Fi = Fi[:, no_new_landmark]
        Ti = Ti[:, no_new_landmark]
    if new_features.size > 0:
        pose = np.asarray(pose)[:3, :].reshape((-1, 1))
        tiled_pose = np.tile(pose, (1, new_features.shape[1]))
        Ci = np.hstack([Ci, new_features])
        Fi = np.hstack([Fi, new_features])
        Ti = np.hstack([Ti, tiled_pose])
    return Pi, Xi, Ci, Fi, Ti


def bootstrap_correspondences(imgs, params, essential_mask):
    """The front-end part of Initialization.get_features_and_k, reference initialization.py:41-70, on this oracle's
    goodFeaturesToTrack / calcOpticalFlowPyrLK; `essential_mask(good_new, good_old)` stands in for the inlier mask of
    cv2.findEssentialMat (:67-68, CPU geometry, out of scope).  Returns (good_old, good_new) and the pairs after every step."""
    corners = goodFeaturesToTrack(imgs[0], params["maxCorners"], params["qualityLevel"], params["minDistance"])
    good_old = np.float32(corners).reshape(-1, 1, 2)
    good_new = good_old
    steps = []
    for i in range(len(imgs) - 1):
        tracked_points, status, _ = calcOpticalFlowPyrLK(imgs[i], imgs[i + 1], good_new, None, winSize=params["winSize"],
                                                         maxLevel=0, criteria=(3, 30, 0.01))
        good_old = good_old[status.flatten() == 1]
        good_new = tracked_points[status.flatten() == 1]
        distances = np.sqrt(np.sum((good_new - good_old) ** 2, axis=2)).flatten()
        drop = np.where(distances < params["dist_threshold_move"])
        good_old = np.delete(good_old, drop, axis=0)
        good_new = np.delete(good_new, drop, axis=0)
        steps.append((good_old.copy(), good_new.copy()))
        mask = np.asarray(essential_mask(good_new, good_old))
        good_old = good_old[mask.flatten() == 1]
        good_new = good_new[mask.flatten() == 1]
    return good_old, good_new, steps
