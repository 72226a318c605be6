"""N > 1 path on CPU: world_size-2 gloo runs of the only cross-rank logic (replicas.fold / shard) and of bench.py's own
launcher (`bench.py --gpus N` starts N rank processes; --stub replaces the GPU launch sets by a sleep)."""
import json
import subprocess
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from visual_odometry_amd import replicas
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    elapsed, (frames, tracked) = replicas.fold(dist, torch.device("cpu"), 1.0 + rank, [100, 1000 * (rank + 1)])
    q.put((rank, elapsed, frames, tracked, replicas.sequence_seed(rank, world), replicas.shard(64, rank, world)[:3]))
    dist.barrier()
    dist.destroy_process_group()


def test_fold_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, elapsed, frames, tracked, seed, sh in res:
        assert elapsed == 2.0 and frames == 200.0 and tracked == 3000.0      # max time, summed units
        assert seed == 100 + rank and sh == [rank, rank + 2, rank + 4]


def test_fold_single_process():
    from visual_odometry_amd import replicas
    assert replicas.fold(None, None, 1.5, [3, 4]) == (1.5, [3.0, 4.0])
    assert replicas.sequence_seed(0, 1) == 0 and replicas.shard(5, 1, 2) == [1, 3]


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, capture_output=True,
                          text=True, timeout=240)


def _line(proc):
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert proc.returncode == 0 and len(lines) == 1, (proc.returncode, proc.stdout, proc.stderr[-2000:])
    # the contract is ONE line on stdout: the process-group banners (gloo here, RCCL on GPUs) go to stderr
    assert proc.stdout.strip() == lines[0], proc.stdout
    return json.loads(lines[0])


def test_bench_launcher_starts_n_ranks_world2_gloo():
    """`bench.py --gpus 2` (no WORLD_SIZE in the environment): the parent starts two rank processes, rank r owns the
    sequences s = r mod 2, rank 0 prints ONE line with n_gpus 2 and the whole-job frame count."""
    out = _line(_bench("--gpus", "2", "--stub", "--steps", "3", "--sets", "4", "--batch", "8"))
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    assert out["sequences_per_rank"] == 8 and out["sequences_rank0"] == [0, 2, 4, 6]
    assert out["frames_total"] == 2 * 3 * 4 * 8                      # both ranks' launch sets, summed by the fold
    assert out["sequence_id_sum"] == sum(range(16))                  # every sequence of the pool is owned exactly once
    assert out["value"] > 0 and out["higher_is_better"] is True
    # the parent of the ranks must never have imported torch (or anything else that can initialise a GPU): a process that
    # has touched the GPU must not exec on the GPU pool, and the parent forks + execs N ranks
    assert out["parent_gpu_modules"] == "" and out["workload"] == "configs1"
    out3 = _line(_bench("--gpus", "2", "--stub", "--steps", "3", "--sets", "4", "--batch", "8", "--workload", "configs3"))
    assert out3["workload"] == "configs3" and out3["sequences_rank0"] == [0, 2, 4, 6] and out3["frames_total"] == 2 * 3 * 4 * 8
    # strong scaling: a fixed pool of 64 sequences (BASELINE configs[3] as written) over 2 ranks
    out = _line(_bench("--gpus", "2", "--stub", "--steps", "2", "--sets", "2", "--total-sequences", "64"))
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["sequences_per_rank"] == 32
    assert out["frames_total"] == 2 * 2 * 64 and out["sequence_id_sum"] == sum(range(64))


def test_bench_rank_under_an_external_launcher():
    """Under torchrun the process is a rank already (WORLD_SIZE set): no second level of processes; --gpus 1 runs in
    the calling process and reports n_gpus 1."""
    out = _line(_bench("--gpus", "1", "--stub", "--steps", "2", "--sets", "2"))
    assert out["n_gpus"] == 1 and out["frames_total"] == 2 * 2 * 64
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {"WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "2", "--sets", "2"],
                           env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), **env), stdout=subprocess.PIPE, text=True)
          for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in ps]
    assert all(p.returncode == 0 for p in ps)
    lines = [l for o in outs for l in o.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_bench_refuses_more_gpus_than_visible():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible")
    p = _bench("--gpus", "2", "--steps", "1")
    assert p.returncode == 2 and "GPU(s) are visible" in p.stderr


def test_bench_parent_is_free_of_torch_and_hip():
    """bench.py's module level and its launcher import neither torch nor the product package nor the oracle: importing
    the file and running the parent's code path up to the spawn leaves sys.modules without them (the device count comes
    from a throw-away child)."""
    code = ("import sys, runpy; sys.argv = ['bench.py']; m = runpy.run_path(%r, run_name='bench_module');"
            "n = m['count_gpus_in_a_child'](); "
            "bad = [k for k in ('torch', 'visual_odometry_amd', 'oracle') if k in sys.modules];"
            "print('COUNT', n, 'BAD', bad)") % os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "BAD []" in r.stdout and "COUNT %d" % torch.cuda.device_count() in r.stdout, r.stdout
