#!/usr/bin/env python3
"""bench.py -- frames/s of the feature front end's LK frame on MI355X (BASELINE.json configs[1]).

One "LK frame"  = a synthetic KITTI-shaped pair: Gaussian pyramids + Scharr planes
of BOTH 1241x376 frames, forward LK of 2000 keypoints (21x21 window, 4 levels, criteria (30, 0.01),
minEigThreshold 1e-3), backward LK from every forward result and the forward-backward check --
exactly what ContinuousVO._track_keypoints asks of cv2 per call (reference continuous.py:131-164).
One step = --batch independent LK frames (default 64) in one pyramid launch set + one LK launch.  Inputs (8 frames, point sets) are resident in HBM before the
timed region; every step goes through the C ABI (libvofe.so).  N > 1: each rank tracks its own
independent sequence (seed 100 + rank) -- replicas only, no collective on the data path.

Prints ONE JSON line on rank 0 (see the driver contract): metric / value (whole-job frames/s) /
roofline (dominant kernel lk_track_kernel, HIP-event timed inside the timed region) /
cpu_baseline (the CPU oracle, a port, timed on the host cores of this box on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
WIN = (21, 21)
MAX_LEVEL = 3
CRITERIA = (3, 30, 0.01)
MIN_EIG = 1e-3
N_FRAMES = 8


# HBM-side bytes of one LK launch, from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
# over this very command (profiles/r01_pmc_bench_fetch.txt, r01_pmc_bench_write.txt; KB x 1024), keyed by
# (pairs per launch, points per pair).  The kernel's loads are narrow (2- and 8-byte gathers), a pattern for
# which the guide gives no gfx950 FETCH_SIZE correction, so the raw value is reported; it is well below the
# algorithmic bytes because the patches are served from L2 / MALL.
LK7_PMC_TRAFFIC_BYTES = {(64, 2000): int((1747015.3 + 7919.3) * 1024), (16, 2000): int((433516.4 + 1810.0) * 1024),
                         (1, 2000): int((18078.0 + 141.4) * 1024)}


def lk_algorithmic_bytes(win, point_levels, iterations):
    """per point-level (w+1)^2 * (1 + 4) B of template patch, plus (w+1)^2 B of
    search-image window per Gauss-Newton iteration."""
    px = (win[0] + 1) * (win[1] + 1)
    return point_levels * px * 5 + iterations * px


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=64,
                    help="independent frame pairs per step: one pyramid launch set + one LK launch for all of them")
    ap.add_argument("--streams", type=int, default=2,
                    help="steps in flight per GPU (one ctx + HIP stream + pyramid object each); 1 = serial steps")
    ap.add_argument("--free-overlap", action="store_true",
                    help="do not order the LK launches of different streams (they then share the CUs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from visual_odometry_amd import synthetic, _lib, replicas
    from visual_odometry_amd.frontend import FrontEnd, Pyramid
    H, W, _, _, _, _ = synthetic.CONFIGS["kitti"]
    fe = FrontEnd(local_rank)
    L = fe.L
    dev = fe.device
    n, B = args.points, max(1, args.batch)
    NP = N_FRAMES - 1                                       # frame pairs of the synthetic sequence

    # ---- inputs, resident in HBM ------------------------------------------------------------
    seed = replicas.sequence_seed(rank, world)
    frames = synthetic.sequence(H, W, seed=seed, n_frames=N_FRAMES)
    d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    pts_host = []
    for t in range(NP):
        c = fe.goodFeaturesToTrack(d_frames[t], n, 0.01, 7)
        pts_host.append(synthetic.pad_points(c, n, H, W, seed + t))
    d_pts = [torch.from_numpy(p).to(dev) for p in pts_host]
    img_stride = H * W
    # A step tracks B independent frame pairs (configs[1]: the pairs do not depend on each other).  Phase f
    # of the input holds pairs (f + k) % NP, k < B, as [B previous frames | B next frames] with the point
    # sets concatenated and an image index per point.
    phases = []
    for f in range(NP):
        pairs = [(f + k) % NP for k in range(B)]
        imgs = torch.cat([d_frames[pairs], d_frames[[q + 1 for q in pairs]]]).contiguous()
        phases.append((imgs, torch.cat([d_pts[q] for q in pairs]).contiguous(), pairs))
    d_idx = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(n).contiguous()
    NB = n * B
    d_cnt = torch.zeros(3, dtype=torch.int64, device=dev)

    lk_chain = [None]                                       # completion event of the most recent LK launch (any stream)

    class Worker:
        """A ctx bound to its own HIP stream with its own pyramid object (2*slots images) and outputs."""
        def __init__(self, stream, pairs):
            self.stream, self.pairs = stream, pairs
            self.k = 0
            self.lk_done = [torch.cuda.Event() for _ in range(4)]
            with torch.cuda.stream(stream):
                self.fe = FrontEnd(local_rank)
                self.pyr = Pyramid(self.fe, 2 * pairs, H, W, WIN, MAX_LEVEL, True)
                self.nxt = torch.empty((n * pairs, 2), dtype=torch.float32, device=dev)
                self.st = torch.empty(n * pairs, dtype=torch.uint8, device=dev)
                self.err = torch.empty(n * pairs, dtype=torch.float32, device=dev)

        def step(self, f, ev=None):
            # reference-call-equivalent: both pyramids of every pair are rebuilt (continuous.py:145-151)
            imgs, pts, _ = phases[f]
            c = self.fe.ctx
            _lib.check(L.vofe_pyramid_build(c, self.pyr._h, 0, 2 * B, imgs.data_ptr(), W, img_stride))
            # LK launches of different streams run one after the other (two VALU-bound launches sharing the CUs gain
            # nothing); the bandwidth-bound pyramid kernels of one stream overlap the LK launch of the other
            if lk_chain[0] is not None:
                self.stream.wait_event(lk_chain[0])
            if ev is not None:
                ev[0].record(self.stream)
            _lib.check(L.vofe_track_fb(c, self.pyr._h, 0, self.pyr._h, B, pts.data_ptr(), d_idx.data_ptr(), B,
                                       self.nxt.data_ptr(), self.st.data_ptr(), self.err.data_ptr(), NB,
                                       CRITERIA[0], CRITERIA[1], CRITERIA[2], MIN_EIG, 1.0))
            if ev is not None:
                ev[1].record(self.stream)
            if args.streams > 1 and not args.free_overlap:
                done = self.lk_done[self.k % len(self.lk_done)]
                self.k += 1
                done.record(self.stream)
                lk_chain[0] = done

        def step_one(self, t):
            # one pair, strictly serial (latency figure)
            c = self.fe.ctx
            _lib.check(L.vofe_pyramid_build(c, self.pyr._h, 0, 2, d_frames[t].data_ptr(), W, img_stride))
            _lib.check(L.vofe_track_fb(c, self.pyr._h, 0, self.pyr._h, 1, d_pts[t].data_ptr(), None, 1,
                                       self.nxt.data_ptr(), self.st.data_ptr(), self.err.data_ptr(), n,
                                       CRITERIA[0], CRITERIA[1], CRITERIA[2], MIN_EIG, 1.0))

        def step_cached(self, t):
            # frame-level schedule (SequenceTracker): the previous frame's pyramid is still in its
            # slot, only the new frame's is built.  At the wrap of the short synthetic sequence frame 0 is rebuilt.
            c = self.fe.ctx
            if t == 0:
                _lib.check(L.vofe_pyramid_build(c, self.pyr._h, 0, 1, d_frames[0].data_ptr(), W, 0))
            _lib.check(L.vofe_pyramid_build(c, self.pyr._h, (t + 1) & 1, 1, d_frames[t + 1].data_ptr(), W, 0))
            _lib.check(L.vofe_track_fb(c, self.pyr._h, t & 1, self.pyr._h, (t + 1) & 1, d_pts[t].data_ptr(), None, 1,
                                       self.nxt.data_ptr(), self.st.data_ptr(), self.err.data_ptr(), n,
                                       CRITERIA[0], CRITERIA[1], CRITERIA[2], MIN_EIG, 1.0))

    torch.cuda.synchronize()
    workers = [Worker(torch.cuda.Stream(device=dev), B) for _ in range(max(1, args.streams))]
    S = len(workers)
    w0 = workers[0]

    def step(i, ev=None):
        workers[i % S].step(i % NP, ev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    # tracked keypoints per phase (status == 1 after the forward-backward check), untimed
    tracked = []
    for f in range(NP):
        w0.step(f)
        w0.stream.synchronize()
        tracked.append(int(w0.st.sum().item()))
    # work counters of the LK launch (point-levels, Gauss-Newton iterations): one untimed pass over the
    # phases, because the counting atomics would slow the timed launches
    _lib.check(L.vofe_lk_set_counters(w0.fe.ctx, d_cnt.data_ptr()))
    for f in range(NP):
        w0.step(f)
    torch.cuda.synchronize()
    _lib.check(L.vofe_lk_set_counters(w0.fe.ctx, None))
    cnt_launch = d_cnt.cpu().numpy().astype(np.float64) / NP
    # HIP events (on the worker's own stream) around every EV_EVERY-th LK launch of the timed region
    EV_EVERY = 4
    events = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for i in range(0, args.steps, EV_EVERY)}

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events.get(i))
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- figures reported beside `value` (this rank only, untimed for the metric) ------------------
    reps = 200
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(reps):
        w0.step_one(i % NP)
    torch.cuda.synchronize()
    serial_ms = (time.perf_counter() - t1) / reps * 1e3
    t3 = time.perf_counter()
    for i in range(reps):
        w0.step_cached(i % NP)
    torch.cuda.synchronize()
    cached_ms = (time.perf_counter() - t3) / reps * 1e3
    # corner detection on the same geometry (one call per frame in the reference, continuous.py:380-385).
    # Device-resident image in, corners copied back to the host.
    gftt_calls = 100
    for i in range(10):
        fe.goodFeaturesToTrack(d_frames[i % N_FRAMES], n, 0.01, 7)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for i in range(gftt_calls):
        fe.goodFeaturesToTrack(d_frames[i % N_FRAMES], n, 0.01, 7)
    torch.cuda.synchronize()
    gftt_us = (time.perf_counter() - t2) / gftt_calls * 1e6

    lk_ms = float(np.mean([a.elapsed_time(b) for a, b in events.values()])) if events else 0.0
    tsum = float(sum(tracked[i % NP] for i in range(args.steps)))
    elapsed, (tsum,) = replicas.fold(dist, dev, elapsed, [tsum])

    frames_total = args.steps * B * world
    fps = frames_total / elapsed

    out = None
    if rank == 0:
        alg_bytes = lk_algorithmic_bytes(WIN, float(cnt_launch[1]), float(cnt_launch[2]))
        achieved = alg_bytes / (lk_ms * 1e-3) / 1e9 if lk_ms > 0 else 0.0
        kname = "lk_track_kernel_x2<2> (two keypoints per wavefront)" if B * n >= 4000 else "lk_track_kernel<7, 21>"
        kernel = "%s: fwd + bwd + FB check of %d pairs x %d points in one launch" % (kname, B, n)
        out = {
            "metric": "frames/sec (LK frame: 2 pyramids + fwd/bwd LK + FB check; KITTI 1241x376, 2k pts)",
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/int32 bilinear + int64 window sums, f32 2x2 solve",
            "data": "synthetic (seeded band-limited texture, KITTI-like warp + N(0,1.5) noise; SURVEY 8d)",
            "config": {"workload": "BASELINE.json configs[1]: synthetic KITTI-shaped 1241x376 frame pairs, %d keypoints, "
                                   "21x21 window, 4 pyramid levels, criteria (30,0.01), minEig 1e-3, "
                                   "reference-call-equivalent (both pyramids rebuilt per pair)" % n,
                       "frames_per_step": B, "points": n, "sequences": world, "streams_per_gpu": S,
                       "parallelism": "replicas only (one sequence per GPU); a step is %d independent frame pairs in one "
                                      "launch set, %d steps in flight per GPU" % (B, S)},
            "single_pair_serial_ms": round(serial_ms, 4),
            "single_pair_serial_cached_pyramid_ms": round(cached_ms, 4),
            "cached_pyramid_note": "same LK work, one pyramid built per new frame (SequenceTracker schedule); serial, this rank",
            "gftt_us_per_call": round(gftt_us, 1),
            "gftt_config": "goodFeaturesToTrack(1241x376, maxCorners=%d, qualityLevel=0.01, minDistance=7), serial calls" % n,
            "tracked_keypoints_per_s": round(tsum / elapsed, 1),
            "attempted_points_per_s": round(frames_total * n / elapsed, 1),
            "roofline": {
                "kernel": kernel,
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": PEAK_HBM_GBS,
                "unit": "GB/s",
                "frac": round(achieved / PEAK_HBM_GBS, 5),
                "traffic": LK7_PMC_TRAFFIC_BYTES.get((B, n)),
                "avg_launch_ms": round(lk_ms, 5),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "point_levels_per_launch": float(cnt_launch[1]),
                "gn_iterations_per_launch": float(cnt_launch[2]),
                "valu_cycles_per_instruction": 4.27,
                "valu_note": "profiles/r01_pmc_lk_valu.txt + r01_ubench_valu_rates.txt: the launch issues one VALU instruction per "
                             "4.27 cycles per SIMD for its whole duration; the instructions that dominate it (v_dot2, v_mul_i32_i24, "
                             "v_add3, DPP adds) measure 4.1-4.2 cycles each in isolation, plain 2-operand ops 2.4",
                "note": "VALU-issue bound, not HBM (DESIGN.md 4/5): patches are L2/MALL resident, so the HBM fraction is "
                        "informational; bytes = point-levels*(w+1)^2*5 + iterations*(w+1)^2; traffic = PMC FETCH_SIZE + "
                        "WRITE_SIZE of this launch shape (profiles/r01_pmc_bench_*.txt)",
            },
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(frames, pts_host, args.cpu_seconds)
    for wk in workers:
        wk.pyr.free()
        wk.fe.close()
    fe.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def cpu_baseline(frames, pts, seconds):
    """The CPU oracle (a port of the OpenCV path, kind 'port') on this box's host cores, same workload,
    bounded sample.  Reported beside the GPU number; it is not the thing measured above."""
    import oracle
    threads = max(1, min(16, os.cpu_count() or 1))

    def run(nthreads, budget):
        done, t0 = 0, time.perf_counter()
        while True:
            t = done % (len(frames) - 1)
            oracle.track_keypoints(pts[t].T, frames[t], frames[t + 1], WIN, MAX_LEVEL, CRITERIA, MIN_EIG, 1.0, nthreads)
            done += 1
            dt = time.perf_counter() - t0
            if dt >= budget and done >= 3:
                return done / dt, done
    oracle.track_keypoints(pts[0].T, frames[0], frames[1], WIN, MAX_LEVEL, CRITERIA, MIN_EIG, 1.0, threads)   # warm-up
    fps_mt, n_mt = run(threads, seconds * 0.6)
    fps_1, n_1 = run(1, seconds * 0.4)
    return {"value": round(fps_mt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d LK frames of the same workload (oracle/vofe_oracle.cpp, std::thread over points)" % n_mt,
            "single_thread_value": round(fps_1, 3), "single_thread_sample_frames": n_1,
            "host_cpu_count": os.cpu_count()}


if __name__ == "__main__":
    main()
