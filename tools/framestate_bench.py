"""Where a frame of the resident frame state goes (host clock around the two calls of vofe_framestate):
python tools/framestate_bench.py [malaga|kitti|parking] [max_corners]

Variants: the frame handed over as a numpy array (copied into the pinned staging buffer by track) or written into the
staging buffer beforehand (the decoder's case: the copy is not part of the frame); update waited for, or queued and
collected by the next track."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import DeviceFrameState, FrontEnd, synthetic  # noqa: E402


def run(cfg="malaga", max_corners=1000, n_land=300, n_cand=700, md=10, n_frames=8, rounds=12):
    H, W, seed, _, win, lvl = synthetic.CONFIGS[cfg]
    fr = synthetic.sequence(H, W, seed, n_frames)
    Kmat = np.array([[700.0, 0, W / 2], [0, 700.0, H / 2], [0, 0, 1.0]])

    def pose_at(t):
        T = np.eye(4)
        T[:3, 3] = [0.02 * t, 0.0, 0.6 * t]
        return T
    fe = FrontEnd(0)
    first = fe.goodFeaturesToTrack(fr[0], n_land + n_cand, 0.01, md).reshape(-1, 2)
    fe.close()
    rng = np.random.default_rng(1)
    P0 = first[:n_land].T.copy()
    X0 = rng.normal(0, 5, (3, P0.shape[1])) + [[0], [0], [20]]
    C0 = first[n_land:].T.copy()
    F0 = C0.astype(np.float64)
    T0 = np.tile(pose_at(0)[:3, :].reshape(-1, 1), (1, C0.shape[1]))
    st = DeviceFrameState(H, W, Kmat, winSize=win, maxLevel=lvl, max_corners=max_corners, quality_level=0.01, min_distance=md)
    out = {"config": cfg, "frame": "%dx%d" % (W, H), "max_corners": max_corners, "landmarks": P0.shape[1], "candidates": C0.shape[1]}
    for staged in (False, True):
        for queued in (False, True):
            acc = {"copy": 0.0, "track": 0.0, "update": 0.0, "frame": 0.0}
            n = 0
            for r in range(rounds + 1):
                st.reset(fr[0], P0, X0, C0, F0, T0)
                for t in range(1, n_frames):
                    t0 = time.perf_counter()
                    if staged:
                        st.frame_buffer[...] = fr[t]
                    t1 = time.perf_counter()
                    _, status = st.track(st.frame_buffer if staged else fr[t])
                    t2 = time.perf_counter()
                    st.update(pose_at(t), status == 1, wait=not queued)
                    t3 = time.perf_counter()
                    if r:
                        acc["copy"] += t1 - t0
                        acc["track"] += t2 - t1
                        acc["update"] += t3 - t2
                        acc["frame"] += (t3 - t1) if staged else (t3 - t0)
                        n += 1
                st.wait()
            out["%s frame, update %s" % ("staged" if staged else "numpy", "queued" if queued else "waited for")] = {
                k: round(v / n * 1e6, 1) for k, v in acc.items()}
    st.close()
    return out


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "malaga"
    mc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    print(json.dumps(run(cfg, mc)))
