"""Multi-GPU sharding of the walk job array (one process per GPU, torch.distributed).

The (start node x num_walks) job array is embarrassingly parallel: the
graph is replicated on every GPU, rank r walks the contiguous slice [lo_r, hi_r) of the *shuffled*
job array, and the walk shards are either left on their GPUs (``gather=False``) or gathered once at
the end (RCCL over xGMI when the backend is ``nccl``; ``gloo`` on CPU tensors in the unit tests).  The only cross-rank dependency is the stream
address of each shard: rank r's first draw is double #(draws of all earlier shards) of the single
MT19937 stream, obtained from an all-gather of per-shard draw counts.
"""
import numpy as np

from .engine import auto_rank0_share, shard_bounds

__all__ = ["sharded_walk_matrix", "shard_bounds", "auto_rank0_share", "RowGather", "isolated_row_filler"]


def _dist():
    import torch.distributed as dist

    return dist


class RowGather:
    """Assembles the contiguous ``[n_rows, width]`` walk matrix on rank ``dst`` from row shards with point-to-point
    transfers (RCCL send/recv over xGMI when the backend is ``nccl``) straight INTO row slices of the preallocated
    matrix: no padded staging tensors, no concatenation afterwards.

    ``bounds[r]`` = rows [lo, hi) rank r produces; every rank may hand its rows over in several pieces
    (``post(lo, hi, rows)`` on the sender, ``expect([(lo, hi, src), ...])`` on the receiver: consecutive sub-ranges of
    a shard, in the same order on both sides -- the transfer of one piece overlaps the walk kernel of the next).  ``known`` (optional bool array over all rows): rows whose content the receiver can
    write itself and which are therefore NOT sent -- walks from starts without neighbours are ``[start, 0, ..., 0, 1]``
    (reference src/pecanpy/pecanpy.py:190-193), half of an R-MAT job array; ``fill_known(full, idx)`` writes them.
    ``dst``'s own rows are expected to be produced in place (``own_rows()`` is a view of the matrix).
    """

    def __init__(self, n_rows, width, bounds, dtype, device, dst=0, group=None, known=None, fill_known=None):
        import torch

        self.dist = _dist()
        self.torch = torch
        self.group, self.dst = group, dst
        self.rank = self.dist.get_rank(group)
        self.bounds = bounds
        self.known = None if known is None else np.asarray(known, dtype=bool)
        self.device = device
        self.full = torch.empty((n_rows, width), dtype=dtype, device=device) if self.rank == dst else None
        self._works, self._scatters, self._sent = [], [], []
        self._side = None      # CUDA receiver: side stream on which every piece is scattered as soon as IT has landed
        if self.rank == dst and torch.device(device).type == "cuda":
            self._side = torch.cuda.Stream(device=device)
        self._sel_cache = {}
        self._known_idx, self._fill_known = None, fill_known
        if self.rank == dst and self.known is not None and fill_known is not None:
            lo, hi = bounds[dst]
            other = self.known.copy()
            other[lo:hi] = False                      # (own rows: written by the walk kernel itself)
            self._known_idx = torch.from_numpy(np.flatnonzero(other)).to(device)
        self.prefill()

    def prefill(self):
        """Receiver: writes the rows nobody sends (once per assembled matrix; a benchmark calls it every pass)."""
        if self._known_idx is not None and self._known_idx.numel():
            self._fill_known(self.full, self._known_idx)

    def own_rows(self):
        lo, hi = self.bounds[self.dst]
        return self.full[lo:hi]

    def _sel(self, lo, hi, device, offset=0):
        """index tensor (relative to lo, plus offset) of the rows of [lo, hi) that travel; None: all of them.  Cached:
        a benchmark posts the same pieces every pass."""
        if self.known is None:
            return None
        key = (lo, hi, str(device), offset)
        if key not in self._sel_cache:
            k = self.known[lo:hi]
            self._sel_cache[key] = None if not k.any() else self.torch.from_numpy(np.flatnonzero(~k) + offset).to(device)
        return self._sel_cache[key]

    def post(self, lo, hi, rows):
        """Sender: starts the transfer of ``rows`` = its rows [lo, hi) (minus the known ones)."""
        torch = self.torch
        if hi <= lo or self.rank == self.dst:
            return
        sel = self._sel(lo, hi, rows.device)
        buf = rows if sel is None else rows.index_select(0, sel)
        if buf.shape[0] == 0:
            return
        buf = buf.contiguous().to(self.device)
        self._sent.append(buf)                        # (kept alive until the transfer is over)
        self._works += self.dist.batch_isend_irecv([self.dist.P2POp(self.dist.isend, buf, self.dst, self.group)])

    def expect(self, pieces):
        """Receiver: starts the receives matching one ``post`` of every sender -- ``pieces`` = [(lo, hi, src), ...] --
        as ONE group, so that the transfers of different peers run side by side (one xGMI link each)."""
        torch = self.torch
        if self.rank != self.dst:
            return
        ops = []
        for lo, hi, src in pieces:
            if hi <= lo:
                continue
            sel = self._sel(lo, hi, self.device, lo)
            if sel is None:
                ops.append(self.dist.P2POp(self.dist.irecv, self.full[lo:hi], src, self.group))
            elif sel.numel():
                stage = torch.empty((sel.numel(), self.full.shape[1]), dtype=self.full.dtype, device=self.device)
                ops.append(self.dist.P2POp(self.dist.irecv, stage, src, self.group))
                self._scatters.append((sel, stage))
        if ops:
            works = self.dist.batch_isend_irecv(ops)
            group_scatters, self._scatters = self._scatters, []
            if self._side is not None and group_scatters:
                # scatter THIS group's pieces as soon as its receives are over, on the side stream: work.wait() on a
                # CUDA stream is a stream-level dependency, the host goes on to walk the next chunk, and at the end of
                # the pass only the last (smallest) group's scatter is left -- not all of them (round 3's finish())
                with torch.cuda.stream(self._side):
                    for w in works:
                        w.wait()
                    for idx, stage in group_scatters:
                        self.full.index_copy_(0, idx, stage)
                        stage.record_stream(self._side)
                self._works += works        # (finish() still waits for them: direct row-slice receives have no scatter)
            else:
                self._works += works
                self._pending = getattr(self, "_pending", []) + group_scatters

    def finish(self):
        """Waits for every transfer; the receiver then holds the complete matrix (returned; others get None)."""
        for w in self._works:
            w.wait()
        for idx, stage in getattr(self, "_pending", []):   # (CPU receiver / no side stream: scatter now)
            self.full.index_copy_(0, idx, stage)
        self._pending = []
        if self._side is not None:
            self.torch.cuda.current_stream(self.device).wait_stream(self._side)
        self._works, self._scatters, self._sent = [], [], []
        return self.full


def isolated_row_filler(starts, walk_length, device):
    """``fill_known`` for RowGather: rows of walks whose start has no neighbour are ``[start, 0, ..., 0, 1]``."""
    import torch

    st = torch.from_numpy(np.ascontiguousarray(starts).view(np.int32)).to(device)

    def fill(full, idx):
        full.index_fill_(0, idx, 0)
        full[idx, 0] = st[idx]
        full[idx, walk_length + 1] = 1

    return fill


def sharded_walk_matrix(run_shard, count_draws, starts, walk_length, group=None, dst=None,
                        max_rounds=None, gather=True, bounds=None):
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
    if bounds is None:
        bounds = shard_bounds(n_jobs, world)      # (uneven bounds -- e.g. a smaller shard for the assembling rank -- may be passed in)
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
    width = walk_length + 2
    if dst is not None:
        # the shards go straight into their rows of ONE preallocated matrix on rank dst (no pads, no cat)
        rg = RowGather(n_jobs, width, bounds, walks.dtype, walks.device, dst=dst, group=group)
        if rank == dst:
            rg.own_rows().copy_(walks)
            rg.expect([(bounds[r][0], bounds[r][1], r) for r in range(world) if r != dst])
        else:
            rg.post(lo, hi, walks)
        return rg.finish()
    # every rank wants the whole matrix: one all-gather of the shards (row counts differ by at most one: pad to the widest)
    rows = max(b[1] - b[0] for b in bounds)
    padded = torch.zeros((rows, width), dtype=walks.dtype, device=walks.device)
    padded[: hi - lo] = walks
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([parts[r][: bounds[r][1] - bounds[r][0]] for r in range(world)], dim=0)


def to_uint32_numpy(t):
    """int32 torch tensor (bit pattern of the uint32 walk matrix) -> NumPy uint32."""
    return t.detach().cpu().numpy().view(np.uint32)
