"""Multi-GPU = replicas only: independent sequences, one per GPU / rank, no exchange
step on the data path.  These helpers hold the only cross-rank logic: which sequence a rank owns
and how per-rank results are folded into the whole-job numbers (max time, summed units)."""


def sequence_seed(rank, world):
    """Seed of the synthetic sequence a rank tracks: the BASELINE KITTI pair (seed 0) when alone,
    else sequence 100 + rank of the 64-sequence config."""
    return 0 if world == 1 else 100 + rank


def shard(n_sequences, rank, world):
    """Sequences s = rank, rank + world, ... (s mod G assignment)."""
    return list(range(rank, n_sequences, world))


def fold(dist, device, elapsed, units, group=None):
    """(max over ranks of elapsed, sum over ranks of each entry of units).  dist may be None; `group` is the process group
    that carries the two reductions (None = the default group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed), [float(u) for u in units]
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    u = torch.tensor(list(units), dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()), [float(x) for x in u.tolist()]
