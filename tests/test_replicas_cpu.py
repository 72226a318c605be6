"""N > 1 path on CPU: world_size-2 gloo run of the only cross-rank logic (replicas.fold / shard)."""
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
