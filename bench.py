#!/usr/bin/env python3
"""bench.py -- frames/s of the feature front end's LK frame on MI355X (BASELINE.json configs[1]).

One "LK frame"  = a synthetic KITTI-shaped pair: Gaussian pyramids + Scharr planes of BOTH 1241x376
frames, forward LK of 2000 keypoints (21x21 window, 4 levels, criteria (30, 0.01), minEigThreshold 1e-3), backward LK
from every forward result and the forward-backward check -- exactly what ContinuousVO._track_keypoints asks of cv2 per
call (reference continuous.py:131-164).  One LAUNCH SET = --batch independent LK frames (default 64): one pyramid
build of 2 x batch images + ONE tracker launch.  One STEP = --sets launch sets (default 40 = 2560 LK frames, ~55 ms),
so that the driver's --steps 20 times more than a second.  Inputs are resident in HBM before the timed region; every
step goes through the C ABI (libvofe.so).

Multi-GPU = replicas only (the reference's frame loop main.py:107-118 is strictly sequential, sequences are
the only unit that shards).  `python bench.py --gpus N` starts N fresh rank processes itself, one per GPU; the parent
never imports torch or calls HIP (a process that has initialised the GPU must not exec on this pool; the device count
comes from a throw-away child), and under torchrun (WORLD_SIZE set) the process is a rank already.  EVERY point of a
scaling curve runs the same workload: by default BASELINE configs[1] on every rank (rank r tracks its own KITTI-shaped
sequence, seed 0 when alone, 100 + r otherwise; per-GPU work fixed = "weak"); `--workload configs3` is BASELINE
configs[3] at any N including 1: rank r tracks sequences r, r + N, r + 2N, ... (seed 100 + s), --batch of them per
launch set (weak), or a fixed pool with --total-sequences 64 ("strong").  No collective on the data path; the process
group carries the barrier and the fold of (max time, summed frames, verification flags).  Every rank checks one frame
pair of its own launch set against the CPU oracle outside the timed region ("all_ranks_verified").

Prints ONE JSON line on rank 0: metric / value (whole-job frames/s) / roofline (dominant kernel: the tracker launch,
HIP-event timed on its own stream inside the timed region) / roofline_streaming (pyramid, Scharr and score-map kernels,
HIP-event timed in this run) / pcie_inclusive / hard_motion / cpu_baseline (the CPU oracle, a port, on this box's host
cores, which also verifies one launch set of the GPU path: "verified").
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
VALU_PEAK_IPC = 0.5            # wave64 VALU instructions per cycle per SIMD (SIMD-32: one wave64 instruction per 2 cycles)
N_SIMD = 1024                  # 256 CUs x 4
WIN = (21, 21)
MAX_LEVEL = 3
CRITERIA = (3, 30, 0.01)
MIN_EIG = 1e-3
N_FRAMES = 8

# Counter figures of the tracker launch that cannot be read inside a run (rocprofv3 --pmc needs a run of its own):
# SQ_INSTS_VALU, GRBM_GUI_ACTIVE (cycles) and FETCH_SIZE / WRITE_SIZE per launch of exactly this launch shape
# (tools/pmc_workload.py under tools/pmc_passes.sh), keyed by (pairs per launch, points per pair).
# Every entry names the kernel source it was measured on by content hash: when that file has changed since, the counter
# figures are not printed as a roofline fraction ("pmc_stale": true, VALU frac null) until the passes are re-run.
LK_PMC = {
    # (lk_track_kernel_x2<2, false, false>: 127 VGPRs, 4 wavefronts per SIMD, the counter-free instance.  History of this launch
    # shape: round 2's 150-VGPR kernel 590.7e6 instructions in 20955989.7 / 8 cycles = 0.220 per cycle per SIMD
    # (profiles/r02_pmc_inst_v5.txt); the 4-wave kernel 579.9e6 in 19008023.7 / 8 = 0.238 (r03_pmc_inst_v2.txt); after the
    # instruction-count work on the iteration and the level prologue 498.8e6 in 17163872.7 / 8 = 0.227 (r03_pmc_inst_v3.txt);
    # with the min-eig quotient behind the level loop and the hand-written reciprocal 492.1e6 in 17007398.0 / 8 = 0.226
    # (r03_pmc_inst_v4.txt))
    (64, 2000): {"valu_instr": 492105763.0, "cycles": 17007398.0 / 8, "fetch_kb": 1827692.3, "write_kb": 6475.1,
                 "kernel_source": ("lk.hip", "f9a996004dca89d4"),
                 "source": "profiles/r03_pmc_inst_v4.txt (SQ_INSTS_VALU, GRBM_GUI_ACTIVE / 8 XCDs), r03_pmc_fetch_v4.txt, r03_pmc_write_v4.txt"},
}
# corner_score_strip<16> on 128 KITTI frames per launch: instructions and cycles of the counter pass, and the issue bound its
# instruction mix allows (profiles/r03_score_kernel_experiments.txt has the derivation and what was tried against it)
SCORE_PMC = {
    128: {"valu_instr": 72438656.0, "cycles": 2805773.0 / 8, "mix_bound_cycles_per_instr": 3.6,
          "kernel_source": ("corners.hip", "01ecd1694d805a95"),
          "source": "profiles/r03_pmc_inst_v5.txt (SQ_INSTS_VALU, GRBM_GUI_ACTIVE / 8 XCDs; same counts as r03_pmc_inst_v1.txt: the strip "
                    "kernel is unchanged, corners.hip gained the any-geometry kernel); "
                    "mix: profiles/r03_score_kernel_experiments.txt"},
}


def source_hash(name):
    """First 16 hex digits of the SHA-256 of a kernel source file (what the PMC tables are keyed to)."""
    import hashlib
    with open(os.path.join(ROOT, "visual-odometry_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_is_current(entry):
    name, digest = entry["kernel_source"]
    try:
        return source_hash(name) == digest
    except OSError:
        return False


def lk_algorithmic_bytes(win, point_levels, iterations):
    """per point-level (w+1)^2 * (1 + 4) B of template patch, plus (w+1)^2 B of search-image window per
    Gauss-Newton iteration."""
    px = (win[0] + 1) * (win[1] + 1)
    return point_levels * px * 5 + iterations * px


def pyramid_level_sizes(h, w, win, max_level):
    out = [(h, w)]
    for _ in range(max_level):
        h, w = (h + 1) // 2, (w + 1) // 2
        if w <= win[0] or h <= win[1]:
            break
        out.append((h, w))
    return out


def streaming_algorithmic_bytes(name, h, w, win, max_level, frames):
    """SURVEY.md 8d, per launch over `frames` images: unpadded planes, each input read once, each output written once."""
    lv = pyramid_level_sizes(h, w, win, max_level)
    px = [a * b for a, b in lv]
    if name.startswith("pad_copy_level0"):
        return frames * (px[0] + (h + 2 * win[1]) * (w + 2 * win[0]))          # image in, padded level-0 plane out
    if name.startswith("pyr_down_u8 L"):
        k = int(name.split("L")[-1])
        return frames * (px[k - 1] + px[k])
    if name.startswith("pyr_step L") or name.startswith("pyr_tail L"):
        # level pipeline (csrc/pyramid.hip, vofe_launch_pyramid): launch k = pyrDown (k-1 -> k) + the k-th geometric share of the
        # Scharr blocks (64 x 8 pixels x 4 rows each) that ride on the pipeline, never beyond level k-1; its tail (default for
        # 3+ levels, VOFE_PYR_TAIL=0 turns it off) produces the last one or two levels and their derivative planes from level
        # `base` through LDS -- level `base` read once, every plane above it written once -- and carries the rest of the riders
        k = int(name.split("L")[-1])
        top = len(lv) - 1
        tail = len(lv) >= 3 and os.environ.get("VOFE_PYR_TAIL", "")[:1] != "0"
        base = max(1, top - 2) if tail else top
        riders = base + 1 if tail else top
        nblk = [((b + 511) // 512) * ((a + 3) // 4) for a, b in lv[:riders]]
        cum = [0]
        for q in range(riders):
            cum.append(cum[-1] + nblk[q])
        denom = 1.0 - 1.0 / (1 << riders)

        def share(j):
            return cum[riders] if j >= riders else min(int(cum[riders] * (1.0 - 1.0 / (1 << j)) / denom), cum[j])
        c0, c1 = share(k - 1), share(k)
        scharr = 0.0
        for q in range(riders):
            a, b = max(c0, cum[q]), min(c1, cum[q + 1])
            if b > a:
                scharr += 5.0 * px[q] * (b - a) / nblk[q]
        if name.startswith("pyr_tail"):
            return int(frames * (px[base] + 5 * sum(px[base + 1:]) + scharr))
        return int(frames * (px[k - 1] + px[k] + scharr))
    if name.startswith("scharr_deriv_s16x2 last"):       # the last level's Scharr pass alone
        return frames * 5 * px[-1]
    if name.startswith("scharr_deriv"):
        return frames * 5 * sum(px)
    if name.startswith("pyramid_borders"):
        band = sum((a + 2 * win[1]) * (b + 2 * win[0]) - a * b for a, b in lv[1:])
        return frames * 2 * band                                                  # border bands: read + written
    if name.startswith("corner_score"):
        return frames * 5 * px[0]
    return None


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=64,
                    help="independent frame pairs per launch set: one pyramid build + one tracker launch for all of them")
    ap.add_argument("--sets", type=int, default=40, help="launch sets per step")
    ap.add_argument("--streams", type=int, default=2,
                    help="launch sets in flight per GPU (one ctx + HIP stream + pyramid object each); 1 = serial")
    ap.add_argument("--free-overlap", action="store_true",
                    help="do not order the tracker launches of different streams (they then share the CUs)")
    ap.add_argument("--workload", choices=("configs1", "configs3"), default="configs1",
                    help="configs1: BASELINE configs[1] on every rank (its own sequence; the headline); configs3: BASELINE "
                         "configs[3], independent sequences sharded s mod N, --batch per GPU")
    ap.add_argument("--total-sequences", type=int, default=0,
                    help="> 0: shard this many sequences over the ranks (strong scaling, e.g. 64 = configs[3] as written); "
                         "0: --batch sequences per GPU (weak scaling)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_streaming / pcie_inclusive / hard_motion")
    ap.add_argument("--share-gpu", action="store_true",
                    help="test aid: every rank uses GPU 0 (rehearses the N-rank path on a one-GPU box; the value is no scaling result)")
    ap.add_argument("--stub", action="store_true",
                    help="test aid: CPU ranks over gloo with a stub step (exercises launcher, sharding, fold, JSON)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def count_gpus_in_a_child():
    """Visible GPUs, counted by a throw-away child process (torch.cuda.device_count() there): the parent stays free of
    torch and HIP, so nothing in it can have initialised a GPU before it starts (fork + exec) the ranks."""
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                       capture_output=True, text=True, timeout=600)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        sys.stderr.write("bench.py: could not count GPUs (%s)\n" % (r.stderr.strip().splitlines()[-1:] or ["no output"])[0])
        return 0


def launch_ranks(args, argv):
    """Parent of `bench.py --gpus N`: N fresh child processes, one per GPU.  This process never imports torch and never
    calls HIP (checked by the CPU tests through the flag handed to the ranks).  Child r gets RANK = LOCAL_RANK = r,
    binds device r and exits with status 2 when that device does not exist.  Returns the exit code."""
    n = args.gpus
    if not args.stub and not args.share_gpu:
        have = count_gpus_in_a_child()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible\n" % (n, have))
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VOFE_BENCH_CHILD="1",
                   VOFE_BENCH_PARENT_GPU_MODULES=",".join(sorted(m for m in ("torch", "visual_odometry_amd", "oracle")
                                                                 if m in sys.modules)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0:
                rc = rc or (code if code > 0 else 1)
                for q in pending:                         # a rank died: the others would wait at the barrier for ever
                    q.terminate()
        time.sleep(0.05)
    return rc


def generate_sequences(ids, h, w, n_frames):
    """Frames of sequences `ids` (seed 100 + s), generated by a few worker processes started before any GPU work."""
    from visual_odometry_amd import synthetic
    if len(ids) <= 2:
        return [synthetic.sequence(h, w, 100 + s, n_frames) for s in ids]
    import multiprocessing as mp
    workers = max(1, min(8, len(ids), (os.cpu_count() or 1) // max(1, int(os.environ.get("WORLD_SIZE", "1")))))
    with mp.get_context("spawn").Pool(workers) as pool:
        return pool.starmap(synthetic.sequence, [(h, w, 100 + s, n_frames) for s in ids])


# ------------------------------------------------------------------------------------------------ one rank
_RESULT_OUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner and gloo its connection lines to stdout when
    a process group comes up: from here on file descriptor 1 is stderr for everything (Python and native libraries), and
    emit_result() writes the line to the real stdout."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit_result(obj):
    out = _RESULT_OUT if _RESULT_OUT is not None else sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def run_rank(args):
    claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from visual_odometry_amd import synthetic, replicas
    H, W, _, _, _, _ = synthetic.CONFIGS["kitti"]
    n, B = args.points, max(1, args.batch)

    # ---- which sequences this rank owns ------------------------------------------------------------------------
    sharded = args.workload == "configs3" or args.total_sequences > 0
    if args.total_sequences > 0:
        my_ids = replicas.shard(args.total_sequences, rank, world)          # strong: a fixed pool, s mod N
        scaling = "strong"
    else:
        my_ids = replicas.shard(B * world, rank, world)                     # weak: B sequences per GPU
        scaling = "weak"
    if sharded:
        B = len(my_ids)
        if B == 0:
            raise SystemExit("bench.py: rank %d owns no sequence (%d sequences over %d ranks)" % (rank, args.total_sequences, world))

    if args.stub:
        return run_stub(args, rank, world, B, scaling, my_ids)

    # frames are generated before the GPU is touched (worker processes)
    if sharded:
        seq_frames = generate_sequences(my_ids, H, W, 2)
    else:
        seq_frames = None
        frames = synthetic.sequence(H, W, seed=replicas.sequence_seed(rank, world), n_frames=N_FRAMES)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0                                    # rehearsal: the ranks share the one GPU of the box
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d has no GPU (%d visible)\n" % (rank, torch.cuda.device_count()))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    ctrl = None                 # the process group that carries the barrier and the fold
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The ranks exchange nothing but the barrier and two small reductions of the timing.  ONE rendezvous, on gloo (it comes up
        # wherever a TCP store does: under torchrun the agent's store, under bench.py's own launcher rank 0's); RCCL, which the
        # launch contract names, is a second group on the same store.  If RCCL cannot come up on this node (every rank agrees on
        # that through the gloo group) the barrier stays on gloo and the line says so ("rendezvous").  No second rendezvous on
        # another port: under torchrun nobody would host it and the ranks would wait for ever.
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ok = 0.0
        try:
            g = dist.new_group(backend="nccl")
            x = torch.zeros(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(x, group=g)
            torch.cuda.synchronize()
            ok = 1.0
        except Exception as e:                                           # noqa: BLE001
            print("bench.py rank %d: RCCL did not come up (%s); the barrier stays on gloo" % (rank, str(e).splitlines()[0][:200]),
                  file=sys.stderr, flush=True)
            g = None
        flag = torch.tensor([ok], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ctrl = g if flag.item() == 1.0 else None
    on_rccl = ctrl is not None
    fold_dev = None if (dist is None or on_rccl) else "cpu"

    from visual_odometry_amd import _lib
    from visual_odometry_amd.frontend import FrontEnd, Pyramid
    fe = FrontEnd(local_rank)
    L = fe.L
    dev = fe.device
    img_stride = H * W

    # ---- inputs, resident in HBM ------------------------------------------------------------------------------
    # A launch set tracks B independent frame pairs laid out as [B previous frames | B next frames], the point sets
    # concatenated, an image index per point.
    phases = []             # (images (2B,H,W) u8, points (B*n,2) f32, host pairs [(prev, next, pts)] for verification)
    if not sharded:
        # BASELINE configs[1]: the seed-0 KITTI sequence; phase f holds its pairs (f + k) % NP, k < B
        NP = N_FRAMES - 1
        d_frames = torch.from_numpy(np.stack(frames)).to(dev)
        pts_host = []
        for t in range(NP):
            c = fe.goodFeaturesToTrack(d_frames[t], n, 0.01, 7)
            pts_host.append(synthetic.pad_points(c, n, H, W, replicas.sequence_seed(rank, world) + t))
        d_pts = [torch.from_numpy(p).to(dev) for p in pts_host]
        for f in range(NP):
            pairs = [(f + k) % NP for k in range(B)]
            imgs = torch.cat([d_frames[pairs], d_frames[[q + 1 for q in pairs]]]).contiguous()
            phases.append((imgs, torch.cat([d_pts[q] for q in pairs]).contiguous(),
                           [(frames[q], frames[q + 1], pts_host[q]) for q in pairs]))
    else:
        # BASELINE configs[3]: slot k of every launch set is the pair (frame 0, frame 1) of this rank's k-th sequence
        d_prev = torch.from_numpy(np.stack([q[0] for q in seq_frames])).to(dev)
        d_next = torch.from_numpy(np.stack([q[1] for q in seq_frames])).to(dev)
        det, cnt = fe.goodFeaturesToTrack_batch(d_prev, n, 0.01, 7)
        det = det.cpu().numpy()
        pts_host = [synthetic.pad_points(det[k, :cnt[k]], n, H, W, 100 + s) for k, s in enumerate(my_ids)]
        frames = seq_frames[0]
        d_frames = torch.from_numpy(np.stack(frames)).to(dev)
        d_pts = [torch.from_numpy(pts_host[0]).to(dev)]
        phases.append((torch.cat([d_prev, d_next]).contiguous(), torch.from_numpy(np.concatenate(pts_host)).to(dev),
                       [(seq_frames[k][0], seq_frames[k][1], pts_host[k]) for k in range(B)]))
    NPH = len(phases)
    d_idx = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(n).contiguous()
    NB = n * B
    d_cnt = torch.zeros(3, dtype=torch.int64, device=dev)

    lk_chain = [None]                                       # completion event of the most recent tracker launch (any stream)

    class Worker:
        """A ctx bound to its own HIP stream with its own pyramid object (2*B slots) and outputs."""
        def __init__(self, stream):
            self.stream = stream
            self.k = 0
            self.lk_done = [torch.cuda.Event() for _ in range(4)]
            with torch.cuda.stream(stream):
                self.fe = FrontEnd(local_rank)
                self.pyr = Pyramid(self.fe, 2 * B, H, W, WIN, MAX_LEVEL, True)
                self.nxt = torch.empty((NB, 2), dtype=torch.float32, device=dev)
                self.st = torch.empty(NB, dtype=torch.uint8, device=dev)
                self.err = torch.empty(NB, dtype=torch.float32, device=dev)

        def launch_set(self, f, ev=None):
            # reference-call-equivalent: both pyramids of every pair are rebuilt (continuous.py:145-151)
            imgs, pts, _ = phases[f]
            c = self.fe.ctx
            _lib.check(L.vofe_pyramid_build(c, self.pyr._h, 0, 2 * B, imgs.data_ptr(), W, img_stride))
            # tracker launches of different streams run one after the other (two VALU-bound launches sharing the CUs
            # gain nothing); the bandwidth-bound pyramid kernels of one stream overlap the tracker launch of the other
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
            # frame-level schedule (SequenceTracker): the previous frame's pyramid is still in its slot,
            # only the new frame's is built.  At the wrap of the short synthetic sequence frame 0 is rebuilt.
            c = self.fe.ctx
            if t == 0:
                _lib.check(L.vofe_pyramid_build(c, self.pyr._h, 0, 1, d_frames[0].data_ptr(), W, 0))
            _lib.check(L.vofe_pyramid_build(c, self.pyr._h, (t + 1) & 1, 1, d_frames[t + 1].data_ptr(), W, 0))
            _lib.check(L.vofe_track_fb(c, self.pyr._h, t & 1, self.pyr._h, (t + 1) & 1, d_pts[t].data_ptr(), None, 1,
                                       self.nxt.data_ptr(), self.st.data_ptr(), self.err.data_ptr(), n,
                                       CRITERIA[0], CRITERIA[1], CRITERIA[2], MIN_EIG, 1.0))

    torch.cuda.synchronize()
    workers = [Worker(torch.cuda.Stream(device=dev)) for _ in range(max(1, args.streams))]
    S = len(workers)
    w0 = workers[0]
    SETS = max(1, args.sets)

    def launch_set(i, ev=None):
        workers[i % S].launch_set(i % NPH, ev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            if on_rccl:
                dist.barrier(group=ctrl, device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup * SETS):
        launch_set(i)
    # tracked keypoints per phase (status == 1 after the forward-backward check), untimed; phase 0's outputs are kept
    # for the verification against the CPU oracle (cpu_baseline leg)
    tracked, gpu_phase0 = [], None
    for f in range(NPH):
        w0.launch_set(f)
        w0.stream.synchronize()
        tracked.append(int(w0.st.sum().item()))
        if f == 0:
            gpu_phase0 = (w0.nxt.cpu().numpy().copy(), w0.st.cpu().numpy().copy(), w0.err.cpu().numpy().copy())
    # work counters of the tracker launch (point-levels, Gauss-Newton iterations): one untimed pass over the phases,
    # because the counting atomics would slow the timed launches
    _lib.check(L.vofe_lk_set_counters(w0.fe.ctx, d_cnt.data_ptr()))
    for f in range(NPH):
        w0.launch_set(f)
    torch.cuda.synchronize()
    _lib.check(L.vofe_lk_set_counters(w0.fe.ctx, None))
    cnt_launch = d_cnt.cpu().numpy().astype(np.float64) / NPH
    # HIP events (on the worker's own stream) around every EV_EVERY-th tracker launch of the timed region
    total_sets = args.steps * SETS
    EV_EVERY = max(1, total_sets // 100)
    events = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for i in range(0, total_sets, EV_EVERY)}

    barrier()
    t0 = time.perf_counter()
    for i in range(total_sets):
        launch_set(i, events.get(i))
    barrier()
    elapsed = time.perf_counter() - t0

    lk_ms = float(np.mean([a.elapsed_time(b) for a, b in events.values()])) if events else 0.0
    tsum = float(sum(tracked[i % NPH] for i in range(total_sets)))
    frames_rank = float(total_sets * B)
    # every rank checks the first frame pair of its own launch set 0 (outputs taken before the timed region) against the
    # CPU oracle -- the checker, outside the timed region; rank 0 at N = 1 checks the whole set in the cpu_baseline leg
    rank_ok = 1.0
    if not args.no_cpu:
        rank_ok = 1.0 if verify_pairs(phases[0][2][:1], gpu_phase0, n)["ok"] else 0.0
    elapsed, (tsum, frames_total, ok_ranks) = replicas.fold(dist, fold_dev or dev, elapsed, [tsum, frames_rank, rank_ok], group=ctrl)
    fps = frames_total / elapsed

    out = None
    if rank == 0:
        # ---- figures reported beside `value` (rank 0 only, outside the timed region) ------------------------------
        reps = 200
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(reps):
            w0.step_one(i % len(d_pts))
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t1) / reps * 1e3
        cached_ms = None
        if not sharded:
            t3 = time.perf_counter()
            for i in range(reps):
                w0.step_cached(i % NP)
            torch.cuda.synchronize()
            cached_ms = (time.perf_counter() - t3) / reps * 1e3
        # corner detection on the same geometry (one call per frame in the reference, continuous.py:380-385):
        # device-resident image in, corners delivered to the host
        gftt_calls = 100
        nfr = d_frames.shape[0]
        for i in range(10):
            fe.goodFeaturesToTrack(d_frames[i % nfr], n, 0.01, 7)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(gftt_calls):
            fe.goodFeaturesToTrack(d_frames[i % nfr], n, 0.01, 7)
        torch.cuda.synchronize()
        gftt_us = (time.perf_counter() - t2) / gftt_calls * 1e6

        alg_bytes = lk_algorithmic_bytes(WIN, float(cnt_launch[1]), float(cnt_launch[2]))
        hbm_achieved = alg_bytes / (lk_ms * 1e-3) / 1e9 if lk_ms > 0 else 0.0
        x2 = NB >= 3200                     # LK_X2_MIN_POINTS of csrc/lk.hip
        kname = "lk_track_kernel_x2<2> (two keypoints per wavefront)" if x2 else "lk_track_kernel<7, 21>"
        kernel = "%s: fwd + bwd + FB check of %d pairs x %d points in one launch" % (kname, B, n)
        pmc = LK_PMC.get((B, n))
        roof = {"kernel": kernel, "avg_launch_ms": round(lk_ms, 5),
                "point_levels_per_launch": float(cnt_launch[1]), "gn_iterations_per_launch": float(cnt_launch[2])}
        hbm = {"bound": "hbm", "achieved": round(hbm_achieved, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": round(hbm_achieved / PEAK_HBM_GBS, 5), "algorithmic_bytes_per_launch": int(alg_bytes),
               "traffic": int((pmc["fetch_kb"] + pmc["write_kb"]) * 1024) if pmc else None,
               "note": "informational: the window patches are served from L2 / Infinity Cache, so counter traffic is "
                       "well below the algorithmic bytes (SURVEY 8d: point-levels*(w+1)^2*5 + iterations*(w+1)^2); "
                       "traffic = PMC FETCH_SIZE + WRITE_SIZE, raw (narrow 2/8/12-byte gathers: no gfx950 correction "
                       "is calibrated for that access shape)"}
        if pmc and not pmc_is_current(pmc):
            # the counter pass was made on another lk.hip: no fraction from it (re-run tools/pmc_passes.sh and update LK_PMC)
            roof.update(hbm)
            roof.update({"pmc_stale": True, "valu": {"bound": "valu", "achieved": None, "peak": VALU_PEAK_IPC, "frac": None,
                                                     "unit": "VALU instr/cycle/SIMD", "measured_on": pmc["kernel_source"][1],
                                                     "current_source": source_hash(pmc["kernel_source"][0])},
                         "traffic": None})
        elif pmc and lk_ms > 0:
            # the launch is bound by VALU issue: instructions and cycles per launch from the counter pass of this launch
            # shape (clock-free); the live HIP-event duration says what clock those cycles ran at in this run
            ipc = pmc["valu_instr"] / N_SIMD / pmc["cycles"]
            roof.update({"bound": "valu", "achieved": round(ipc, 4), "peak": VALU_PEAK_IPC,
                         "unit": "VALU instr/cycle/SIMD", "frac": round(ipc / VALU_PEAK_IPC, 4),
                         "traffic": hbm["traffic"], "valu_instructions_per_launch": pmc["valu_instr"],
                         "cycles_per_launch": pmc["cycles"], "implied_clock_ghz": round(pmc["cycles"] / (lk_ms * 1e-3) / 1e9, 3),
                         "pmc_source": pmc["source"],
                         "mix_bound_note": "the kernel's instructions are mostly the half-rate class (v_dot2_i32_i16, "
                                           "v_mad_i32_i24, v_add3, DPP adds: 4.1-4.2 cycles each in isolation, "
                                           "profiles/r01_ubench_valu_rates.txt), so the mix allows ~0.24 instr/cycle/SIMD; "
                                           "round 3: the search-region loads moved behind the minimum-eigenvalue test, 150 -> 127 VGPRs = 4 "
                                           "instead of 3 wavefronts per SIMD (0.220 -> 0.238 instr/cycle/SIMD: the launch runs at the bound "
                                           "its instruction mix sets), then 14 % fewer instructions per launch (579.9 M -> 498.8 M: position "
                                           "test of an iteration, packed reduction, weights, level prologue); the instructions removed were "
                                           "mostly full-rate ones, so the rate reads 0.227 while the launch got 9 % shorter",
                         "hbm": hbm})
        else:
            roof.update(hbm)
        out = {
            "metric": "frames/sec (LK frame: 2 pyramids + fwd/bwd LK + FB check; KITTI 1241x376, 2k pts)",
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "rendezvous": ("none" if dist is None else ("nccl" if on_rccl else "gloo")),
            **({"rehearsal": "%d ranks share one GPU (--share-gpu): launcher / sharding / fold exercised, not a scaling figure" % world}
               if args.share_gpu else {}),
            "vs_baseline": None,
            "dtype": "u8/int32 bilinear + int64 window sums, f32 2x2 solve",
            "data": "synthetic (seeded band-limited texture, KITTI-like warp + N(0,1.5) noise; SURVEY 8d)",
            "config": {"workload": ("BASELINE.json configs[3]: independent synthetic KITTI-shaped 1241x376 sequences (seed 100 + s), "
                                    "sequence s on rank s mod N, %d per GPU, %d keypoints each, 21x21 window, 4 levels" % (B, n))
                                   if sharded else
                                   ("BASELINE.json configs[1]: synthetic KITTI-shaped 1241x376 frame pairs, %d keypoints, "
                                    "21x21 window, 4 pyramid levels, criteria (30,0.01), minEig 1e-3, "
                                    "reference-call-equivalent (both pyramids rebuilt per pair)%s" % (
                                        n, "" if world == 1 else "; every rank runs this workload on its own sequence (seed 100 + rank)")),
                       "frames_per_launch_set": B, "launch_sets_per_step": SETS, "frames_per_step": B * SETS * world,
                       "points": n, "sequences": B * world if sharded else world, "streams_per_gpu": S,
                       "parallelism": "replicas only: sequences sharded s mod N over %d GPU(s), no collective on the data path; "
                                      "a launch set is %d independent frame pairs, %d launch sets in flight per GPU" % (world, B, S)},
            "timed_seconds": round(elapsed, 4),
            "single_pair_serial_ms": round(serial_ms, 4),
            "single_pair_serial_cached_pyramid_ms": round(cached_ms, 4) if cached_ms is not None else None,
            "cached_pyramid_note": "same LK work, one pyramid built per new frame (SequenceTracker schedule); serial, this rank",
            "gftt_us_per_call": round(gftt_us, 1),
            "gftt_config": "goodFeaturesToTrack(1241x376, maxCorners=%d, qualityLevel=0.01, minDistance=7), serial calls" % n,
            "tracked_keypoints_per_s": round(tsum / elapsed, 1),
            "attempted_points_per_s": round(frames_total * n / elapsed, 1),
            "roofline": roof,
        }
        # the figures beside the headline belong to the N = 1 line (the CPU baseline is timed on rank 0 at N = 1 only)
        if not args.no_extras and world == 1:
            out["roofline_streaming"] = streaming_rooflines(torch, fe, _lib, Pyramid, phases[0][0], H, W)
            out["pcie_inclusive"] = pcie_inclusive(fe, frames, pts_host[0], H, W, n)
            out["hard_motion"] = hard_motion(torch, fe, _lib, Pyramid, H, W, n, min(B, 64), d_cnt)
        out["verified"] = None
        out["all_ranks_verified"] = None if args.no_cpu else bool(ok_ranks == world)
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(phases[0][2], gpu_phase0, n, args.cpu_seconds)
            out["verified"] = out["cpu_baseline"].pop("verified")
    for wk in workers:
        wk.pyr.free()
        wk.fe.close()
    fe.close()
    if out is not None:
        emit_result(out)                      # the line is out before the groups are torn down
    if dist is not None:
        dist.barrier()                        # (default group: gloo)
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------ extras (rank 0)
def streaming_rooflines(torch, fe, _lib, Pyramid, imgs, H, W):
    """The HBM-bound kernels on the launch bench.py's steps make (2 x batch KITTI frames per launch), each bracketed
    by HIP events on the context's stream (vofe_profile_begin / vofe_profile_end); bytes per SURVEY 8d."""
    L = fe.L
    frames = int(imgs.shape[0])
    pyr = Pyramid(fe, frames, H, W, WIN, MAX_LEVEL, True)
    score = torch.empty((frames, H, W), dtype=torch.float32, device=imgs.device)
    reps = 10

    def once():
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, frames, imgs.data_ptr(), W, H * W))
        _lib.check(L.vofe_corner_score(fe.ctx, imgs.data_ptr(), frames, H, W, W, H * W, 3, 3, 0, 0.04, score.data_ptr()))
    for _ in range(3):
        once()
    fe.synchronize()
    _lib.check(L.vofe_profile_begin(fe.ctx))
    for _ in range(reps):
        once()
    cap = 64 * reps
    names = C.create_string_buffer(cap * 48)
    ms = (C.c_float * cap)()
    cnt = C.c_int(0)
    _lib.check(L.vofe_profile_end(fe.ctx, cap, names, ms, C.byref(cnt)))
    acc = {}
    for i in range(cnt.value):
        nm = names.raw[i * 48:(i + 1) * 48].split(b"\0", 1)[0].decode()
        acc.setdefault(nm, []).append(ms[i])
    pyr.free()
    out = []
    for nm, ts in acc.items():
        t = float(np.mean(ts))
        b = streaming_algorithmic_bytes(nm, H, W, WIN, MAX_LEVEL, frames)
        e = {"kernel": nm, "frames_per_launch": frames, "avg_launch_ms": round(t, 5), "launches_timed": len(ts)}
        if b is not None and t > 0:
            gbs = b / (t * 1e-3) / 1e9
            e.update({"bound": "hbm", "algorithmic_bytes_per_launch": int(b), "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                      "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)})
        sp = SCORE_PMC.get(frames) if nm.startswith("corner_score") else None
        if sp:
            # the score map is bound by VALU issue, not by HBM: its own roofline beside the byte figure
            if pmc_is_current(sp):
                ipc = sp["valu_instr"] / N_SIMD / sp["cycles"]
                e["valu"] = {"bound": "valu", "achieved": round(ipc, 4), "peak": VALU_PEAK_IPC, "frac": round(ipc / VALU_PEAK_IPC, 4),
                             "unit": "VALU instr/cycle/SIMD", "valu_instructions_per_pixel": round(sp["valu_instr"] / (frames * H * W), 2),
                             "mix_bound": round(1.0 / sp["mix_bound_cycles_per_instr"], 4),
                             "frac_of_mix_bound": round(ipc * sp["mix_bound_cycles_per_instr"], 4),
                             "note": "~100 of the 239 instructions per 4-pixel row step are half-rate class (OpenCV's f64 box sums and "
                                     "their conversions, u8 -> f32, the rounded sqrt), so back-to-back issue would take 3.6 cycles per "
                                     "instruction: at that bound the kernel would reach ~30 % of HBM peak, not 60 %",
                             "pmc_source": sp["source"]}
            else:
                e["valu"] = {"bound": "valu", "achieved": None, "frac": None, "pmc_stale": True}
        out.append(e)
    return out


def pcie_inclusive(fe, frames, pts, H, W, n):
    """Host arrays in, host arrays out, as the reference hands them over (continuous.py:145-151, main.py:70-79): the
    new image goes up, (nextPts, status, err) come back.  Never `value`."""
    from visual_odometry_amd.sequence import SequenceTracker
    kp = np.ascontiguousarray(pts.T)
    nf = len(frames)
    reps = 100
    for i in range(5):
        fe.track_keypoints(kp, frames[i % (nf - 1)], frames[i % (nf - 1) + 1], WIN, MAX_LEVEL, CRITERIA, MIN_EIG)
    t0 = time.perf_counter()
    for i in range(reps):
        fe.track_keypoints(kp, frames[i % (nf - 1)], frames[i % (nf - 1) + 1], WIN, MAX_LEVEL, CRITERIA, MIN_EIG)
    per_call = (time.perf_counter() - t0) / reps
    trk = SequenceTracker(H, W, WIN, MAX_LEVEL, CRITERIA, MIN_EIG, frontend=fe)
    trk.reset(frames[0])
    for i in range(5):
        trk.track(frames[(i + 1) % nf], kp)
    t0 = time.perf_counter()
    for i in range(reps):
        trk.track(frames[(i + 1) % nf], kp)
    cached = (time.perf_counter() - t0) / reps
    trk.close()
    for i in range(5):
        fe.goodFeaturesToTrack(frames[i % nf], n, 0.01, 7)
    t0 = time.perf_counter()
    for i in range(reps):
        fe.goodFeaturesToTrack(frames[i % nf], n, 0.01, 7)
    det = (time.perf_counter() - t0) / reps
    # the reference's whole per-frame front end (continuous.py:86-92 landmark tracking + :375-495 _handle_candidates) on
    # the native frame-state object: numpy frame in, landmark positions out, pose + inlier mask in; the reference's
    # own settings (1000 corners / 10 px), 300 landmarks + 700 candidates at the start
    from visual_odometry_amd import DeviceFrameState
    first = fe.goodFeaturesToTrack(frames[0], 1000, 0.01, 10).reshape(-1, 2)
    P0, C0 = first[:300].T.copy(), first[300:].T.copy()
    X0 = np.random.default_rng(1).normal(0, 5, (3, P0.shape[1])) + [[0], [0], [20]]

    def pose_at(t):
        T = np.eye(4)
        T[:3, 3] = [0.02 * t, 0.0, 0.6 * t]
        return T
    T0 = np.tile(pose_at(0)[:3, :].reshape(-1, 1), (1, C0.shape[1]))
    st = DeviceFrameState(H, W, np.array([[718.856, 0, 607.19], [0, 718.856, 185.2], [0, 0, 1.0]]), winSize=WIN, maxLevel=MAX_LEVEL,
                          criteria=CRITERIA, minEigThreshold=MIN_EIG, max_corners=1000, quality_level=0.01, min_distance=10)
    t_fs, n_fs = 0.0, 0
    for r in range(13):
        st.reset(frames[0], P0, X0, C0, C0.astype(np.float64), T0)
        t0 = time.perf_counter()
        for t in range(1, nf):
            _, status = st.track(frames[t])
            st.update(pose_at(t), status == 1)
        if r:
            t_fs += time.perf_counter() - t0
            n_fs += nf - 1
    st.close()
    fs_frame = t_fs / n_fs
    return {"track_keypoints_us": round(per_call * 1e6, 1), "track_keypoints_frames_per_s": round(1.0 / per_call, 1),
            "front_end_frame_us": round(fs_frame * 1e6, 1), "front_end_frames_per_s": round(1.0 / fs_frame, 1),
            "front_end_frame": "vofe_framestate (track + update): landmark tracking, candidate tracking, detection (1000 corners, 10 px), "
                               "compaction, de-duplication, triangulation, promotion, append; %d landmarks + %d candidates at the start" % (
                                   P0.shape[1], C0.shape[1]),
            "sequence_tracker_track_us": round(cached * 1e6, 1), "sequence_tracker_frames_per_s": round(1.0 / cached, 1),
            "goodFeaturesToTrack_us": round(det * 1e6, 1),
            "note": "serial calls with numpy in / numpy out through frontend.py: track_keypoints uploads both 1241x376 frames "
                    "and rebuilds both pyramids (reference-call-equivalent); SequenceTracker.track uploads the new frame "
                    "only and builds one pyramid; %d points, results read back" % kp.shape[1]}


def hard_motion(torch, fe, _lib, Pyramid, H, W, n, B, d_cnt):
    """Informational: the same launch on a harder sequence (24 px/frame flow + 1 %/frame zoom -> 10-40 px, 5 % of each
    frame occluded), so the sensitivity of the headline to the Gauss-Newton iteration count is on record."""
    from visual_odometry_amd import synthetic
    L = fe.L
    fr = synthetic.sequence(H, W, seed=0, n_frames=N_FRAMES, shift=(22.0, 9.0), zoom=0.01, occlusion=0.05, margin=232)
    d_fr = torch.from_numpy(np.stack(fr)).to(fe.device)
    NP = N_FRAMES - 1
    pts = []
    for t in range(NP):
        pts.append(synthetic.pad_points(fe.goodFeaturesToTrack(d_fr[t], n, 0.01, 7), n, H, W, t))
    pairs = [k % NP for k in range(B)]
    imgs = torch.cat([d_fr[pairs], d_fr[[q + 1 for q in pairs]]]).contiguous()
    d_p = torch.from_numpy(np.concatenate([pts[q] for q in pairs])).to(fe.device)
    idx = torch.arange(B, dtype=torch.int32, device=fe.device).repeat_interleave(n).contiguous()
    NB = n * B
    nxt = torch.empty((NB, 2), dtype=torch.float32, device=fe.device)
    st = torch.empty(NB, dtype=torch.uint8, device=fe.device)
    err = torch.empty(NB, dtype=torch.float32, device=fe.device)
    pyr = Pyramid(fe, 2 * B, H, W, WIN, MAX_LEVEL, True)

    def once():
        _lib.check(L.vofe_pyramid_build(fe.ctx, pyr._h, 0, 2 * B, imgs.data_ptr(), W, H * W))
        _lib.check(L.vofe_track_fb(fe.ctx, pyr._h, 0, pyr._h, B, d_p.data_ptr(), idx.data_ptr(), B, nxt.data_ptr(),
                                   st.data_ptr(), err.data_ptr(), NB, CRITERIA[0], CRITERIA[1], CRITERIA[2], MIN_EIG, 1.0))
    d_cnt.zero_()
    _lib.check(L.vofe_lk_set_counters(fe.ctx, d_cnt.data_ptr()))
    once()
    fe.synchronize()
    _lib.check(L.vofe_lk_set_counters(fe.ctx, None))
    c = d_cnt.cpu().numpy().astype(np.float64)
    tracked = float(st.sum().item()) / NB
    for _ in range(3):
        once()
    fe.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    fe.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pyr.free()
    return {"frames_per_s": round(B / dt, 1), "ms_per_launch_set": round(dt * 1e3, 4), "frames_per_launch_set": B,
            "gn_iterations_per_point": round(c[2] / max(c[0], 1), 2), "level_passes_per_point": round(c[1] / max(c[0], 1), 2),
            "tracked_fraction": round(tracked, 4), "streams": 1,
            "motion": "22 x 9 px per frame translation + 1 % per frame zoom (flow 10-40 px), 5 % of every frame occluded, N(0,1.5) noise"}


def verify_pairs(pairs, gpu_phase0, n, threads=None):
    """Frame pairs [(prev, next, pts)] = slots 0.. of launch set 0 of the GPU path (gpu_phase0 = its outputs, taken before
    the timed region) against oracle.track_keypoints, pair by pair.  The checker: never timed, never shipped."""
    import oracle
    threads = threads or max(1, min(16, os.cpu_count() or 1))
    distinct = {}
    for k, (a, b, p) in enumerate(pairs):                      # configs[1] repeats its 7 pairs over the 64 slots
        distinct.setdefault((id(a), id(b), id(p)), (a, b, p, []))[3].append(k)
    g_nxt, g_st, g_err = gpu_phase0
    checked, exact, status_equal, max_dpos = 0, True, 1.0, 0.0
    for a, b, p, slots in distinct.values():
        t0_, s0_, e0_ = oracle.track_keypoints(p.T, a, b, WIN, MAX_LEVEL, CRITERIA, MIN_EIG, 1.0, threads)
        for k in slots:
            o = k * n
            gs, gn, ge = g_st[o:o + n], g_nxt[o:o + n].T, g_err[o:o + n]
            status_equal = min(status_equal, float(np.mean(gs == s0_)))
            ok = (gs == 1) & (s0_ == 1)
            if ok.any():
                max_dpos = max(max_dpos, float(np.abs(gn - t0_)[:, ok].max()))
            exact = exact and np.array_equal(gs, s0_) and np.array_equal(gn, t0_) and np.array_equal(ge, e0_)
            checked += 1
    return {"ok": bool(status_equal >= 0.995 and max_dpos <= 0.05), "bit_exact": bool(exact), "pairs_checked": checked,
            "min_status_agreement": round(status_equal, 6), "max_position_difference_px": max_dpos,
            "what": "every frame pair of one launch set of the timed GPU path (outputs taken before the timed "
                    "region) against oracle.track_keypoints; bars 99.5 % status, 0.05 px"}


def cpu_baseline(pairs, gpu_phase0, n, seconds):
    """The CPU oracle (a port of the OpenCV path, kind 'port') on this box's host cores, same workload, bounded sample.
    Reported beside the GPU number, never the thing measured above.  The same oracle doubles as the checker of the
    run: every pair of launch set 0 of the GPU path is compared with the oracle's triple ("verified")."""
    import oracle
    threads = max(1, min(16, os.cpu_count() or 1))
    seen, work = set(), []
    for a, b, p in pairs:
        if (id(a), id(b), id(p)) not in seen:
            seen.add((id(a), id(b), id(p)))
            work.append((a, b, p))
    verified = verify_pairs(pairs, gpu_phase0, n, threads)

    def run(nthreads, budget):
        done, t0 = 0, time.perf_counter()
        while True:
            a, b, p = work[done % len(work)]
            oracle.track_keypoints(p.T, a, b, WIN, MAX_LEVEL, CRITERIA, MIN_EIG, 1.0, nthreads)
            done += 1
            dt = time.perf_counter() - t0
            if dt >= budget and done >= 3:
                return done / dt, done
    # timed with the port's vector loops (AVX2 window loops, bit-identical to the scalar ones the checker above ran: the CPU
    # number beside the GPU's should not be a scalar loop when the reference's cv2 runs SIMD code); the scalar rate is kept
    with oracle.simd():
        fps_mt, n_mt = run(threads, seconds * 0.5)
        fps_1, n_1 = run(1, seconds * 0.3)
    fps_s1, n_s1 = run(1, seconds * 0.2)
    return {"value": round(fps_mt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d forward-backward LK frames of the same workload, pyramids included (oracle/vofe_oracle.cpp: AVX2 window "
                      "loops, std::thread over points and pyramid rows)" % n_mt,
            "single_thread_value": round(fps_1, 3), "single_thread_sample_frames": n_1,
            "single_thread_scalar_value": round(fps_s1, 3), "single_thread_scalar_sample_frames": n_s1,
            "host_cpu_count": os.cpu_count(), "verified": verified}


# ------------------------------------------------------------------------------------------------ stub rank (CPU test)
def run_stub(args, rank, world, B, scaling, my_ids):
    """`--stub`: the rank logic without a GPU -- gloo process group, a sleep for a launch set -- so the launcher, the
    sharding, the barrier / fold and the JSON line are exercised by the CPU tests."""
    import torch
    import torch.distributed as dist
    from visual_odometry_amd import replicas
    d = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        d = dist
        dist.barrier()
    total_sets = args.steps * max(1, args.sets)
    t0 = time.perf_counter()
    for _ in range(total_sets):
        time.sleep(0.0005)
    if d is not None:
        d.barrier()
    elapsed = time.perf_counter() - t0
    elapsed, (frames_total, id_sum) = replicas.fold(d, torch.device("cpu"), elapsed, [float(total_sets * B), float(sum(my_ids))])
    if rank == 0:
        emit_result({"metric": "frames/sec (stub)", "value": round(frames_total / elapsed, 2), "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
                          "stub": True, "frames_total": frames_total, "sequence_id_sum": id_sum, "workload": args.workload,
                          "parent_gpu_modules": os.environ.get("VOFE_BENCH_PARENT_GPU_MODULES"),
                     "sequences_rank0": my_ids[:4], "sequences_per_rank": B})
    if d is not None:
        d.barrier()
        d.destroy_process_group()
    return 0


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
