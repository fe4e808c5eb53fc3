"""Multi-GPU sharding of the walk job array (one process per GPU, torch.distributed).

The (start node x num_walks) job array is embarrassingly parallel: the
graph is replicated on every GPU, rank r walks the contiguous slice [lo_r, hi_r) of the *shuffled*
job array, and the walk shards are either left on their GPUs (``gather=False``) or gathered once at
the end (RCCL over xGMI when the backend is ``nccl``; ``gloo`` on CPU tensors in the unit tests).  The only cross-rank dependency is the stream
address of each shard: rank r's first draw is double #(draws of all earlier shards) of the single
MT19937 stream, obtained from an all-gather of per-shard draw counts.
"""
import numpy as np

from .engine import shard_bounds

__all__ = ["sharded_walk_matrix", "shard_bounds"]


def _dist():
    import torch.distributed as dist

    return dist


def sharded_walk_matrix(run_shard, count_draws, starts, walk_length, group=None, dst=None,
                        max_rounds=None, gather=True):
    """Walk ``starts`` cooperatively across the ranks of ``group``.

    run_shard(starts_slice, stream_skip) -> (walks, actual_draws)
        walks: torch tensor [n, walk_length + 2] (int32 view of the uint32 matrix), on the device
        the backend communicates from; actual_draws = sum(len - 1) of the shard.
    count_draws(starts_slice) -> nominal number of draws of the slice (no dead ends assumed)

    Returns the full [n_jobs, walk_length + 2] tensor on every rank (``dst=None``) or only on
    rank ``dst`` (others get ``None``).  ``gather=False`` skips the collective: every rank gets
    ``(walks_of_its_shard, (lo, hi))`` -- rows [lo, hi) of the full matrix, left where they were
    produced (the jobs are independent; nothing downstream of the walks needs them on one device).
    """
    import torch

    dist = _dist()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_jobs = int(starts.shape[0])
    bounds = shard_bounds(n_jobs, world)
    lo, hi = bounds[rank]
    mine = starts[lo:hi]

    def allgather_counts(value, device):
        t = torch.tensor([int(value)], dtype=torch.int64, device=device)
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        return [int(o.item()) for o in outs]

    comm_device = None
    draws = count_draws(mine)
    walks = None
    skip = None
    rounds = 0
    limit = max_rounds if max_rounds is not None else world + 1
    while True:
        if comm_device is None:
            # first round: need a device for the count exchange before any walk tensor exists
            probe, _ = run_shard(mine[:0], 0)
            comm_device = probe.device
        counts = allgather_counts(draws, comm_device)
        new_skip = sum(counts[:rank])
        if new_skip != skip:
            skip = new_skip
            walks, actual = run_shard(mine, skip)
            draws = int(actual)
        rounds += 1
        # a shard that consumed fewer draws than announced (dead ends) shifts every later shard
        after = allgather_counts(draws, comm_device)
        if after == counts or rounds >= limit:
            break

    if not gather:
        return walks, (lo, hi)
    # one gather of the shards (row counts differ by at most one: pad to the widest)
    width = walk_length + 2
    rows = max(b[1] - b[0] for b in bounds)
    padded = torch.zeros((rows, width), dtype=walks.dtype, device=walks.device)
    padded[: hi - lo] = walks
    if dst is None:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
    else:
        parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
        dist.gather(padded, parts, dst=dst, group=group)
        if rank != dst:
            return None
    return torch.cat([parts[r][: bounds[r][1] - bounds[r][0]] for r in range(world)], dim=0)


def to_uint32_numpy(t):
    """int32 torch tensor (bit pattern of the uint32 walk matrix) -> NumPy uint32."""
    return t.detach().cpu().numpy().view(np.uint32)
