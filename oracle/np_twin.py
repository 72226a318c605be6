"""Independent numpy restatement of SURVEY.md Appendix A.  TEST INFRASTRUCTURE ONLY.

A second, structurally different statement of the same published algorithm (whole-array numpy
instead of the C++ loops of vofe_oracle.cpp), used by tests/ to pin the C++ oracle against
transcription slips.  Small inputs only (the LK part is a Python loop per point).  PARITY UNPINNED
against cv2 for the same reason as the C++ oracle.
"""
import numpy as np

F32 = np.float32


def _pad101(a, p):
    """REFLECT_101 padding of the first two axes; an axis of length 1 repeats its only element."""
    if np.isscalar(p):
        p = ((p, p), (p, p))
    for ax in (0, 1):
        w = [(0, 0)] * a.ndim
        w[ax] = tuple(p[ax])
        a = np.pad(a, w, mode="reflect" if a.shape[ax] > 1 else "edge")
    return a


def pyr_down(img):
    """A.3: [1 4 6 4 1]^2, stride 2, (sum+128)>>8, REFLECT_101."""
    h, w = img.shape
    a = _pad101(img.astype(np.int64), 2)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    dh, dw = (h + 1) // 2, (w + 1) // 2
    rows = sum(k[i] * a[:, i:i + 2 * dw:2][:, :dw] for i in range(5))       # horizontal pass
    out = sum(k[j] * rows[j:j + 2 * dh:2][:dh] for j in range(5))           # vertical pass
    return ((out + 128) >> 8).astype(np.uint8)


def pyramid(img, win, max_level):
    """A.3 buildOpticalFlowPyramid: list of unpadded levels with the early stop."""
    levels = [img]
    h, w = img.shape
    for _ in range(max_level):
        h, w = (h + 1) // 2, (w + 1) // 2
        if w <= win[0] or h <= win[1]:
            break
        levels.append(pyr_down(levels[-1]))
    return levels


def scharr(img):
    """A.4: Ix = [3 10 3]^T x [-1 0 1], Iy = [-1 0 1]^T x [3 10 3], int16 interleaved."""
    a = _pad101(img.astype(np.int32), 1)
    t0 = 3 * (a[:-2] + a[2:]) + 10 * a[1:-1]
    t1 = a[2:] - a[:-2]
    ix = t0[:, 2:] - t0[:, :-2]
    iy = 3 * (t1[:, :-2] + t1[:, 2:]) + 10 * t1[:, 1:-1]
    return np.stack([ix, iy], -1).astype(np.int16)


def corner_score(img, harris=False, k=0.04):
    """A.1 with the canonical operation order."""
    s1, s2 = F32(1.0 / 3060.0), F32(2.0 / 3060.0)
    a = _pad101(img, 1).astype(F32)
    ai = _pad101(img, 1).astype(np.int32)
    r = a[:, 2:] - a[:, :-2]                                   # row pass [-1 0 1], rows -1..h
    dx = s2 * r[1:-1] + s1 * (r[:-2] + r[2:])
    q = s2 * a[:, 1:-1] + s1 * (ai[:, :-2] + ai[:, 2:]).astype(F32)
    dy = q[2:] - q[:-2]
    out = []
    for c in (dx * dx, dx * dy, dy * dy):
        cp = _pad101(c, 1).astype(np.float64)
        rs = (cp[:, :-2] + cp[:, 1:-1]) + cp[:, 2:]
        out.append(((rs[:-2] + rs[1:-1]) + rs[2:]).astype(F32))
    c0, c1, c2 = out
    if not harris:
        a_, b_, c_ = c0 * F32(0.5), c1, c2 * F32(0.5)
        return (a_ + c_) - np.sqrt((a_ - c_) * (a_ - c_) + b_ * b_)
    kf = F32(k)
    return (c0 * c2 - c1 * c1) - kf * ((c0 + c2) * (c0 + c2))


def sobel_taps(ksize, order):
    """cv::getSobelKernels for one axis in closed form: binomial smoothing taps (order 0) = [1 1] convolved ksize-1 times;
    first-derivative taps (order 1) = the binomial of length ksize-1 convolved with [-1 1].  ksize 1 with order 1 -> 3 taps."""
    if ksize == 1 and order > 0:
        ksize = 3
    k = np.array([1], np.int64)
    for _ in range(ksize - order - 1):
        k = np.convolve(k, [1, 1])
    if order:
        k = np.convolve(k, [-1, 1])
    return k


def corner_score_generic(img, block=3, ksize=3, harris=False, k=0.04):
    """A.1 for any blockSize / Sobel aperture, same operation order as oracle/vofe_oracle.cpp corner_score_generic, written
    with whole-array shifts instead of per-pixel loops."""
    h, w = img.shape
    scale = 1.0 / (float(1 << (ksize - 1)) * block * 255.0)
    dt, st = sobel_taps(ksize, 1), sobel_taps(ksize, 0)
    ld, ls = len(dt), len(st)
    hd, hs = ld // 2, ls // 2
    fs = [F32(float(t) * scale) for t in st]
    fd = [F32(t) for t in dt]
    hm = max(hd, hs)
    a = _pad101(img, ((0, 0), (hm, hm)))
    af, ai = a.astype(F32), a.astype(np.int32)

    def col(j, half):                                   # image column x - half + j for every x
        return slice(hm - half + j, hm - half + j + w)
    r = fd[0] * af[:, col(0, hd)]
    for j in range(1, ld):
        r = r + fd[j] * af[:, col(j, hd)]
    if ls == 3:
        q = fs[1] * af[:, col(1, 1)] + fs[0] * (ai[:, col(0, 1)] + ai[:, col(2, 1)]).astype(F32)
    else:
        q = fs[0] * af[:, col(0, hs)]
        for j in range(1, ls):
            q = q + fs[j] * af[:, col(j, hs)]
    rp, qp = _pad101(r, ((hm, hm), (0, 0))), _pad101(q, ((hm, hm), (0, 0)))

    def row(t):
        return slice(hm + t, hm + t + h)
    dx = fs[hs] * rp[row(0)]
    for t in range(1, hs + 1):
        dx = dx + fs[hs + t] * (rp[row(t)] + rp[row(-t)])
    dy = np.zeros((h, w), F32)
    for t in range(1, hd + 1):
        dy = dy + fd[hd + t] * (qp[row(t)] - qp[row(-t)])
    an = block // 2
    out = []
    for c in (dx * dx, dx * dy, dy * dy):
        cp = _pad101(c, ((an, block - 1 - an), (an, block - 1 - an))).astype(np.float64)
        rs = cp[:, 0:w]
        for i in range(1, block):
            rs = rs + cp[:, i:i + w]
        tot = rs[0:h]
        for j in range(1, block):
            tot = tot + rs[j:j + h]
        out.append(tot.astype(F32))
    c0, c1, c2 = out
    if not harris:
        a_, b_, c_ = c0 * F32(0.5), c1, c2 * F32(0.5)
        return (a_ + c_) - np.sqrt((a_ - c_) * (a_ - c_) + b_ * b_)
    kf = F32(k)
    return (c0 * c2 - c1 * c1) - kf * ((c0 + c2) * (c0 + c2))


def gftt(img, max_corners, quality, min_distance, mask=None, harris=False, k=0.04, block=3, ksize=3, eig=None):
    """A.2, sequential greedy with a brute-force distance test (no grid: A.2 step 6 shows the grid
    never changes the outcome because cell >= minDistance - 0.5).  `eig`: a score map to select from instead of this
    module's own (tests feed scipy's float64 evaluation of the definition)."""
    if eig is None:
        eig = corner_score(img, harris, k) if (block == 3 and ksize == 3) else corner_score_generic(img, block, ksize, harris, k)
    h, w = eig.shape
    sel = eig if mask is None else eig[mask != 0]
    if sel.size == 0:
        return None
    thr = F32(float(sel.max()) * quality)
    e = np.where(eig > thr, eig, F32(0))
    ep = np.pad(e, 1, mode="constant", constant_values=-np.inf)
    dil = np.max([ep[j:j + h, i:i + w] for j in range(3) for i in range(3)], axis=0)
    ok = (e != 0) & (e == dil)
    ok[0, :] = ok[-1, :] = False
    ok[:, 0] = ok[:, -1] = False
    if mask is not None:
        ok &= mask != 0
    ys, xs = np.nonzero(ok)
    if ys.size == 0:
        return None
    idx = ys * w + xs
    order = np.lexsort((-idx, -e[ys, xs].astype(np.float64)))
    acc = []
    md2 = float(min_distance) ** 2
    for o in order:
        x, y = int(xs[o]), int(ys[o])
        if min_distance >= 1 and acc:
            p = np.asarray(acc, np.float64)
            if np.any((p[:, 0] - x) ** 2 + (p[:, 1] - y) ** 2 < md2):
                continue
        acc.append((x, y))
        if max_corners > 0 and len(acc) == max_corners:
            break
    return np.asarray(acc, F32).reshape(-1, 1, 2)


def _floor_i(v):
    if not (-1073741824.0 < v < 1073741824.0):
        return -2 ** 31
    return int(np.floor(v))


def _weights(fx, fy, ix, iy):
    a, b = F32(fx) - F32(ix), F32(fy) - F32(iy)
    one, sc = F32(1), F32(16384)
    w00 = int(np.rint((one - a) * (one - b) * sc))
    w01 = int(np.rint(a * (one - b) * sc))
    w10 = int(np.rint((one - a) * b * sc))
    return w00, w01, w10, 16384 - w00 - w01 - w10


def _sample(P, x0, y0, w, h, wt, shift):
    """Bilinear window sample of padded int plane P at integer origin (x0,y0)."""
    a = P[y0:y0 + h + 1, x0:x0 + w + 1].astype(np.int64)
    v = a[:-1, :-1] * wt[0] + a[:-1, 1:] * wt[1] + a[1:, :-1] * wt[2] + a[1:, 1:] * wt[3]
    return (v + (1 << (shift - 1))) >> shift


def lk(prev_img, next_img, pts, win=(21, 21), max_level=3, max_count=30, eps=0.01, flags=0,
       min_eig_thr=1e-4, init=None):
    """A.4/A.5 per point; returns (next (n,2) f32, status (n,) u8, err (n,) f32)."""
    w, h = win
    Ps, Ns = pyramid(prev_img, win, max_level), pyramid(next_img, win, max_level)
    top = len(Ps) - 1
    I = [_pad101(p.astype(np.int32), ((h, h), (w, w))) for p in Ps]
    J = [_pad101(p.astype(np.int32), ((h, h), (w, w))) for p in Ns]
    D = [np.pad(scharr(p).astype(np.int32), ((h, h), (w, w), (0, 0))) for p in Ps]
    hx, hy = F32((w - 1) * 0.5), F32((h - 1) * 0.5)
    eps2 = min(max(eps, 0.0), 10.0) ** 2
    scale20 = F32(1.0 / (1 << 20))
    n = len(pts)
    out = np.zeros((n, 2), F32)
    st = np.ones(n, np.uint8)
    err = np.zeros(n, F32)
    for i in range(n):
        px, py = F32(pts[i][0]), F32(pts[i][1])
        ox, oy = (F32(init[i][0]), F32(init[i][1])) if (flags & 4) else (F32(0), F32(0))
        for L in range(top, -1, -1):
            sc = F32(1.0 / (1 << L))
            ppx, ppy = px * sc, py * sc
            if L == top:
                npx, npy = (ox * sc, oy * sc) if (flags & 4) else (ppx, ppy)
            else:
                npx, npy = ox * F32(2), oy * F32(2)
            ox, oy = npx, npy
            rows, cols = Ps[L].shape
            ppx, ppy = ppx - hx, ppy - hy
            ipx, ipy = _floor_i(ppx), _floor_i(ppy)
            if ipx < -w or ipx >= cols or ipy < -h or ipy >= rows:
                if L == 0:
                    st[i] = 0
                    err[i] = 0
                continue
            wt = _weights(ppx, ppy, ipx, ipy)
            Ip = _sample(I[L], ipx + w, ipy + h, w, h, wt, 9)
            Ix = _sample(D[L][..., 0], ipx + w, ipy + h, w, h, wt, 14)
            Iy = _sample(D[L][..., 1], ipx + w, ipy + h, w, h, wt, 14)
            A11 = F32(int((Ix * Ix).sum())) * scale20
            A12 = F32(int((Ix * Iy).sum())) * scale20
            A22 = F32(int((Iy * Iy).sum())) * scale20
            Dd = A11 * A22 - A12 * A12
            me = (A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + F32(4) * A12 * A12)) / F32(2 * w * h)
            if flags & 8:
                err[i] = me
            if me < F32(min_eig_thr) or Dd < F32(1.1920928955078125e-07):
                if L == 0:
                    st[i] = 0
                continue
            Dd = F32(1) / Dd
            npx, npy = npx - hx, npy - hy
            pdx = pdy = F32(0)
            for j in range(max_count):
                inx, iny = _floor_i(npx), _floor_i(npy)
                if inx < -w or inx >= cols or iny < -h or iny >= rows:
                    if L == 0:
                        st[i] = 0
                    break
                wj = _weights(npx, npy, inx, iny)
                diff = _sample(J[L], inx + w, iny + h, w, h, wj, 9) - Ip
                b1 = F32(int((diff * Ix).sum())) * scale20
                b2 = F32(int((diff * Iy).sum())) * scale20
                dx = (A12 * b2 - A22 * b1) * Dd
                dy = (A12 * b1 - A11 * b2) * Dd
                npx, npy = npx + dx, npy + dy
                ox, oy = npx + hx, npy + hy
                if float(dx) * float(dx) + float(dy) * float(dy) <= eps2:
                    break
                if j > 0 and abs(float(dx + pdx)) < 0.01 and abs(float(dy + pdy)) < 0.01:
                    ox, oy = ox - dx * F32(0.5), oy - dy * F32(0.5)
                    break
                pdx, pdy = dx, dy
            if st[i] and L == 0 and not (flags & 8):
                ex, ey = ox - hx, oy - hy
                inx, iny = _floor_i(ex), _floor_i(ey)
                if inx < -w or inx >= cols or iny < -h or iny >= rows:
                    st[i] = 0
                    continue
                wj = _weights(ex, ey, inx, iny)
                diff = _sample(J[L], inx + w, iny + h, w, h, wj, 9) - Ip
                err[i] = F32(int(np.abs(diff).sum())) / F32(32 * w * h)
        out[i] = (ox, oy)
    return out, st, err
