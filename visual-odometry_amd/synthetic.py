"""Seeded synthetic inputs of the BASELINE.json shapes.

No dataset ships with the reference (its data/ is git-ignored), so every measurement and parity
test runs on band-limited noise textures warped by a known KITTI-like motion.
"""
import numpy as np

CONFIGS = {
    # name: (H, W, seed, n_points, win, max_level)
    "parking": (480, 640, 3, 1000, (11, 11), 3),
    "kitti": (376, 1241, 0, 2000, (21, 21), 3),
    "malaga": (600, 800, 1, 3000, (21, 21), 3),
    "stress": (2160, 3840, 2, 20000, (31, 31), 4),
}


def _canvas(h, w, rng):
    from scipy.ndimage import gaussian_filter
    acc = np.zeros((h, w), np.float64)
    for sigma, wt in ((1.5, 0.5), (4.0, 0.3), (12.0, 0.2)):
        o = gaussian_filter(rng.standard_normal((h, w)), sigma, mode="reflect")
        acc += wt * o / o.std()
    lo, hi = np.percentile(acc, (0.5, 99.5))
    return np.clip((acc - lo) / (hi - lo), 0.0, 1.0) * 255.0


def sequence(h, w, seed=0, n_frames=2, noise_sigma=1.5, shift=(1.5, 0.4), zoom=0.002, occlusion=0.0, margin=32):
    """Frames t = 0..n_frames-1: canvas sampled under translation shift*t px + zoom 1+zoom*t about the image centre
    + N(0, noise_sigma) sensor noise, rounded to u8.  The defaults are SURVEY 8d's KITTI-like motion (1.5, 0.4) px
    per frame.  occlusion > 0 (harder-motion variant of bench.py): in every frame after the first, rectangles that
    cover about that fraction of the frame are replaced by unrelated texture, so the points under them cannot be
    tracked.  `margin` = canvas border in pixels (enlarge it for large shifts; out-of-canvas samples are mirrored)."""
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(seed)
    canvas = _canvas(h + 2 * margin, w + 2 * margin, rng)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = (w - 1) * 0.5, (h - 1) * 0.5
    frames = []
    for t in range(n_frames):
        z = 1.0 + zoom * t
        sx = (xs - cx) / z + cx + shift[0] * t + float(margin)
        sy = (ys - cy) / z + cy + shift[1] * t + float(margin)
        f = map_coordinates(canvas, [sy, sx], order=1, mode="reflect")
        f = f + rng.normal(0.0, noise_sigma, f.shape)
        frames.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
    if occlusion > 0:
        orng = np.random.default_rng(seed + 104729)
        other = _canvas(h, w, orng)
        for t in range(1, n_frames):
            covered = 0
            while covered < occlusion * h * w:
                rh, rw = int(orng.integers(20, 80)), int(orng.integers(20, 120))
                y0, x0 = int(orng.integers(0, max(1, h - rh))), int(orng.integers(0, max(1, w - rw)))
                frames[t][y0:y0 + rh, x0:x0 + rw] = np.clip(np.rint(other[y0:y0 + rh, x0:x0 + rw]), 0, 255).astype(np.uint8)
                covered += rh * rw
    return frames


def true_flow(pts, h, w, t0, t1):
    """Ground-truth position in frame t1 of points (n,2) given in frame t0 (same warp as sequence())."""
    cx, cy = (w - 1) * 0.5, (h - 1) * 0.5
    p = np.asarray(pts, np.float64).reshape(-1, 2)
    z0, z1 = 1.0 + 0.002 * t0, 1.0 + 0.002 * t1
    ux = (p[:, 0] - cx) / z0 + cx + 1.5 * t0
    uy = (p[:, 1] - cy) / z0 + cy + 0.4 * t0
    return np.stack([(ux - 1.5 * t1 - cx) * z1 + cx, (uy - 0.4 * t1 - cy) * z1 + cy], -1)


def pad_points(corners, n, h, w, seed):
    """detected corners padded to n with uniform-random f32 positions, ~2 % of all points
    within 10 px of the border (exercises the out-of-bounds branches)."""
    rng = np.random.default_rng(seed + 7919)
    pts = np.zeros((0, 2), np.float32) if corners is None else np.array(corners, np.float32).reshape(-1, 2)
    pts = pts[:n]
    short = n - len(pts)
    if short > 0:
        extra = np.stack([rng.uniform(0, w - 1, short), rng.uniform(0, h - 1, short)], -1)
        pts = np.concatenate([pts, extra.astype(np.float32)])
    nb = max(1, n // 50)
    idx = rng.choice(n, nb, replace=False)
    side = rng.integers(0, 4, nb)
    d = rng.uniform(0, 10, nb)
    bx = np.where(side == 0, d, np.where(side == 1, w - 1 - d, rng.uniform(0, w - 1, nb)))
    by = np.where(side == 2, d, np.where(side == 3, h - 1 - d, rng.uniform(0, h - 1, nb)))
    pts[idx] = np.stack([bx, by], -1).astype(np.float32)
    return np.ascontiguousarray(pts, np.float32)
