#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[2..4] and a synthetic frame of configs[0]'s shape (Parking 640x480,
11x11 window, 1000 points; the real sequence and cv2 are not in the image) -- bench.py covers configs[1].

Front-end frame = what one process_frame asks of the feature front end: forward-backward
tracking of the frame's points + goodFeaturesToTrack on the new frame.  Reported device-resident
(images and points already in HBM, results left there except the corner list) in the
reference-call-equivalent mode (both pyramids rebuilt) and the cached mode (one pyramid per new
frame), and host-to-host through the Python mirror.  Prints one JSON object per config.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visual_odometry_amd import synthetic, _lib, SequenceTracker   # noqa: E402
from visual_odometry_amd.frontend import FrontEnd, Pyramid         # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def front_end_frame(fe, cfg, md, reps):
    H, W, seed, n, win, lvl = synthetic.CONFIGS[cfg]
    L = fe.L
    fr = synthetic.sequence(H, W, seed, 3)
    d = torch.from_numpy(np.stack(fr)).cuda()
    c = fe.goodFeaturesToTrack(d[0], n, 0.01, md)
    pts = synthetic.pad_points(c, n, H, W, seed)
    dp = torch.from_numpy(pts).cuda()
    dn = torch.empty_like(dp); ds = torch.empty(n, dtype=torch.uint8, device="cuda"); de = torch.empty(n, device="cuda")
    pyr = Pyramid(fe, 2, H, W, win, lvl, True)

    def track(slot_a, slot_b):
        _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, slot_a, pyr._h, slot_b, dp.data_ptr(), None, 1, dn.data_ptr(), ds.data_ptr(),
                                   de.data_ptr(), n, 3, 30, 0.01, 1e-3, 1.0))

    def ref_equiv():
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2, d.data_ptr(), W, H * W))
        track(0, 1)
        fe.goodFeaturesToTrack(d[1], n, 0.01, md)

    def cached():
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 1, 1, d[1].data_ptr(), W, 0))
        track(0, 1)
        fe.goodFeaturesToTrack(d[1], n, 0.01, md)

    def lk_only():
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2, d.data_ptr(), W, H * W))
        track(0, 1)

    t_ref, t_cached, t_lk = timed(ref_equiv, reps), timed(cached, reps), timed(lk_only, reps)
    tracked = int(ds.sum().item())
    pyr.free()
    trk = SequenceTracker(H, W, winSize=win, maxLevel=lvl, frontend=fe)
    trk.reset(fr[0])
    k = [0]

    def host():
        k[0] += 1
        trk.track(fr[k[0] % 3], pts.T)
        trk.detect(n, 0.01, md)
    t_host = timed(host, max(3, reps // 4))

    def host_overlapped():
        k[0] += 1
        trk.track_and_detect(fr[k[0] % 3], (pts.T,), n, 0.01, md)
    t_host_ov = timed(host_overlapped, max(3, reps // 4))
    trk.close()
    return {"config": cfg, "frame": "%dx%d, %d points, win %dx%d, %d levels, GFTT maxCorners %d minDistance %d" %
            (W, H, n, win[0], win[1], lvl + 1, n, md),
            "front_end_frame_us_reference_call_equivalent": round(t_ref * 1e6, 1),
            "front_end_frame_us_cached_pyramid": round(t_cached * 1e6, 1),
            "lk_frame_us": round(t_lk * 1e6, 1),
            "front_end_frame_us_host_numpy_in_out_cached": round(t_host * 1e6, 1),
            "front_end_frame_us_host_numpy_in_out_detector_overlapped": round(t_host_ov * 1e6, 1),
            "frames_per_s_cached": round(1 / t_cached, 1), "tracked_of_attempted": [tracked, n]}


def sequences_64(fe, reps):
    H, W, seed, n, win, lvl = synthetic.CONFIGS["kitti"]
    L = fe.L
    B = 64
    base = [synthetic.sequence(H, W, 100 + s, 2) for s in range(4)]
    pts = [synthetic.pad_points(fe.goodFeaturesToTrack(b[0], n, 0.01, 7), n, H, W, 100 + s) for s, b in enumerate(base)]
    d = torch.from_numpy(np.stack([base[s % 4][0] for s in range(B)] + [base[s % 4][1] for s in range(B)])).cuda()
    dp = torch.from_numpy(np.concatenate([pts[s % 4] for s in range(B)])).cuda()
    idx = torch.arange(B, dtype=torch.int32).repeat_interleave(n).cuda()
    N = n * B
    dn = torch.empty_like(dp); ds = torch.empty(N, dtype=torch.uint8, device="cuda"); de = torch.empty(N, device="cuda")
    pyr = Pyramid(fe, 2 * B, H, W, win, lvl, True)

    def step():
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2 * B, d.data_ptr(), W, H * W))
        _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, dp.data_ptr(), idx.data_ptr(), B, dn.data_ptr(), ds.data_ptr(),
                                   de.data_ptr(), N, 3, 30, 0.01, 1e-3, 1.0))
    t = timed(step, reps)
    out, cnt = None, None

    def gftt():
        nonlocal out, cnt
        out, cnt = fe.goodFeaturesToTrack_batch(d[:B], 1000, 0.01, 10)
    tg = timed(gftt, max(2, reps // 2))
    pyr.free()
    return {"config": "64 sequences (one launch set per frame index)", "lk_frames_per_s": round(B / t, 1),
            "us_per_64_frames": round(t * 1e6, 1), "gftt_batch64_us": round(tg * 1e6, 1),
            "gftt_us_per_image": round(tg * 1e6 / B, 2), "tracked_fraction": round(float(ds.float().mean()), 4)}


def frame_loop(fe, cfg, md, n_land, n_cand, max_corners, n_frames=8, rounds=6):
    """The reference's whole per-frame front end, host to host (numpy frame in; tracked landmarks out for the CPU pose
    estimation; pose + inlier mask in; updated state): landmark tracking + _handle_candidates (continuous.py:86-92,
    375-495).  (a) per-call path: SequenceTracker.track for the landmarks, FrontEnd.handle_candidates for the rest (every
    stage on the device, arrays through the host between stages); (b) DeviceFrameState: arrays resident, two device phases."""
    from visual_odometry_amd import DeviceFrameState
    H, W, seed, _, win, lvl = synthetic.CONFIGS[cfg]
    fr = synthetic.sequence(H, W, seed, n_frames)
    Kmat = np.array([[700.0, 0, W / 2], [0, 700.0, H / 2], [0, 0, 1.0]])

    def pose_at(t):
        T = np.eye(4)
        T[:3, 3] = [0.02 * t, 0.0, 0.6 * t]
        return T
    first = fe.goodFeaturesToTrack(fr[0], n_land + n_cand, 0.01, md).reshape(-1, 2)
    rng = np.random.default_rng(1)
    P0 = first[:n_land].T.copy()
    X0 = rng.normal(0, 5, (3, P0.shape[1])) + [[0], [0], [20]]
    C0 = first[n_land:].T.copy()
    F0 = C0.astype(np.float64)
    T0 = np.tile(pose_at(0)[:3, :].reshape(-1, 1), (1, C0.shape[1]))
    kw = dict(max_corners=max_corners, quality_level=0.01, min_distance=md)
    lk = dict(winSize=win, maxLevel=lvl)
    st = DeviceFrameState(H, W, Kmat, winSize=win, maxLevel=lvl, frontend=fe, **kw)
    trk = SequenceTracker(H, W, winSize=win, maxLevel=lvl, frontend=fe)
    t_dev = t_call = 0.0
    sizes = None
    for r in range(rounds + 1):
        st.reset(fr[0], P0, X0, C0, F0, T0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(1, n_frames):
            _, status = st.track(fr[t])
            st.update(pose_at(t), status == 1, wait=False)     # collected by the next track(), behind its frame upload
        st.wait()
        dt = time.perf_counter() - t0
        sizes = (st.K, st.M)
        trk.reset(fr[0])
        Pi, Xi, Ci, Fi, Ti = P0, X0, C0, F0, T0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(1, n_frames):
            (tp, ts, _), = trk.track(fr[t], Pi)
            keep = ts == 1
            Pi, Xi, Ci, Fi, Ti = fe.handle_candidates(Kmat, fr[t], fr[t - 1], Ci, Fi, Ti, tp[:, keep], Xi[:, keep], pose_at(t),
                                                     lk_params=lk, **kw)
        dc = time.perf_counter() - t0
        if r:                                     # the first round warms up
            t_dev += dt
            t_call += dc
    st.close()
    trk.close()
    nfr = rounds * (n_frames - 1)
    return {"config": cfg + " frame loop", "frame": "%dx%d, %d landmarks + %d candidates at the start, GFTT maxCorners %d minDistance %d, "
            "win %dx%d" % (W, H, P0.shape[1], C0.shape[1], max_corners, md, win[0], win[1]),
            "frame_us_device_resident_state": round(t_dev / nfr * 1e6, 1), "frame_us_per_call_path": round(t_call / nfr * 1e6, 1),
            "landmarks_candidates_at_the_end": sizes, "frames_timed": nfr,
            "note": "host to host: numpy frame in, landmark positions out, pose + mask in; pose estimation itself not included"}


def main():
    fe = FrontEnd(0)
    if len(sys.argv) > 1 and sys.argv[1] == "loop":
        for r in (frame_loop(fe, "malaga", 10, 1000, 2000, 3000), frame_loop(fe, "malaga", 10, 300, 700, 1000),
                  frame_loop(fe, "kitti", 10, 300, 700, 1000), frame_loop(fe, "parking", 10, 300, 700, 1000)):
            print(json.dumps(r), flush=True)
        fe.close()
        return
    res = [front_end_frame(fe, "parking", 10, 100), front_end_frame(fe, "malaga", 10, 100), front_end_frame(fe, "stress", 10, 20),
           sequences_64(fe, 8)]
    fe.close()
    for r in res:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
