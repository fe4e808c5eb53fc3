"""Multi-GPU sharding logic on CPU: world_size 2, gloo backend.

The per-rank walk operator is replaced by the CPU oracle (allowed here: tests may use the oracle
as the engine-under-test's stand-in); what is checked is the product's sharding code -- shard
bounds, stream addressing of each shard (stream_skip from an all-gather of draw counts), the
re-run when a shard consumed fewer draws than announced (dead ends), and the final gather.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle as orc
from pecanpy_amd.sharding import RowGather, isolated_row_filler, shard_bounds, sharded_walk_matrix, to_uint32_numpy
from pecanpy_amd.synth import rmat_csr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _graph(kind):
    if kind == "undirected":
        return rmat_csr(9, seed=5)
    # directed graph with sinks: dead ends shift the stream address of later shards
    rng = np.random.default_rng(1)
    n = 60
    adj = rng.random((n, n)) < 0.06
    np.fill_diagonal(adj, False)
    adj[rng.choice(n, 8, replace=False), :] = False
    indptr = np.zeros(n + 1, dtype=np.uint32)
    indptr[1:] = np.cumsum(adj.sum(1))
    indices = np.nonzero(adj)[1].astype(np.uint32)
    return indptr, indices, np.ones(indices.size, dtype=np.float32)


def _worker(rank, world, port, kind, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    indptr, indices, data = _graph(kind)
    L, seed = 12, 3
    starts = orc.shuffled_starts(indptr.size - 1, 3, seed)
    has = indptr[1:] != indptr[:-1]
    calls = []

    def run_shard(sl, skip):
        calls.append(len(sl))
        mat = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, sl, L, seed, stream_skip=skip)
        actual = int((mat[:, -1].astype(np.int64) - 1).sum())
        return torch.from_numpy(mat.view(np.int32).copy()), actual

    def count_draws(sl):
        return int(has[sl].sum()) * L

    full = sharded_walk_matrix(run_shard, count_draws, starts, L)
    if rank == 0:
        ret["full"] = to_uint32_numpy(full)
        ret["runs_rank0"] = sum(1 for c in calls if c)
    only0 = sharded_walk_matrix(run_shard, count_draws, starts, L, dst=0)
    assert (only0 is None) == (rank != 0)
    # no collective on the data path: every rank keeps rows [lo, hi) of the same matrix
    local, (lo, hi) = sharded_walk_matrix(run_shard, count_draws, starts, L, gather=False)
    assert (lo, hi) == tuple(shard_bounds(starts.size, world)[rank])
    ret[f"local{rank}"] = (lo, hi, to_uint32_numpy(local))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["undirected", "directed_with_sinks"])
def test_two_rank_shards_reassemble_the_single_stream(kind):
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, kind, ret), nprocs=2, join=True)
        indptr, indices, data = _graph(kind)
        starts = orc.shuffled_starts(indptr.size - 1, 3, 3)
        want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 12, 3)
        assert np.array_equal(ret["full"], want)
        for r in range(2):
            lo, hi, rows = ret[f"local{r}"]
            assert np.array_equal(rows, want[lo:hi])
        if kind == "directed_with_sinks":
            assert (want[:, -1] < 13).any()          # the case really has mid-walk dead ends


def test_shard_bounds_cover_the_job_array():
    for n, w in [(10, 3), (7, 8), (41943040, 8), (1, 2)]:
        b = shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


def _unseeded_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pecanpy_amd import pecanpy as node2vec

    indptr, indices, data = rmat_csr(8, seed=2)
    g = node2vec.SparseOTF.from_csr(indptr, indices, data, p=0.5, q=2, random_state=None)
    seed = g._call_seed()
    ret[f"seed{rank}"] = seed
    ret[f"starts{rank}"] = g._start_array(3, seed)
    dist.destroy_process_group()


def test_unseeded_ranks_shuffle_the_same_job_array():
    """random_state=None (the CLI default) under torch.distributed: every rank must shard the SAME shuffled job
    array and address the same stream, or nodes get too many / too few walks."""
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_unseeded_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret["seed0"] is not None and ret["seed0"] == ret["seed1"]
        assert np.array_equal(ret["starts0"], ret["starts1"])
        n = ret["starts0"].size // 3
        assert np.array_equal(np.bincount(ret["starts0"], minlength=n), np.full(n, 3))


def _gather_worker(rank, world, port, ret, share=None):
    """bench.py's data path: every shard walked in chunks, each chunk's rows posted while the next is walked; rows of
    isolated starts are not sent (rank 0 writes them itself); rank 0 walks its own shard in place."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    indptr, indices, data = rmat_csr(9, seed=5)
    L, seed, n_chunks = 12, 3, 3
    starts = orc.shuffled_starts(indptr.size - 1, 3, seed)
    has = (indptr[1:] != indptr[:-1])[starts]
    n_jobs = starts.size
    bounds = shard_bounds(n_jobs, world, share)   # (share: a smaller shard for rank 0, which also assembles the matrix)
    dev = torch.device("cpu")
    rg = RowGather(n_jobs, L + 2, bounds, torch.int32, dev, dst=0, known=~has, fill_known=isolated_row_filler(starts, L, dev))
    lo, hi = bounds[rank]
    from pecanpy_amd.engine import tapered_bounds   # (bench.py's chunking: decreasing sizes)

    chunks = [(lo + a, lo + b) for a, b in tapered_bounds(hi - lo, n_chunks)]
    for c in range(n_chunks):
        a, b = chunks[c]
        skip = int(has[:a].sum()) * L
        mat = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts[a:b], L, seed, stream_skip=skip)
        rows = torch.from_numpy(mat.view(np.int32).copy())
        if rank == 0:
            rg.full[a:b] = rows                  # (the walk kernel writes rank 0's rows in place)
            rg.expect([(bounds[r][0] + tapered_bounds(bounds[r][1] - bounds[r][0], n_chunks)[c][0],
                        bounds[r][0] + tapered_bounds(bounds[r][1] - bounds[r][0], n_chunks)[c][1], r) for r in range(1, world)])
        else:
            rg.post(a, b, rows)
    full = rg.finish()
    if rank == 0:
        want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, L, seed)
        ret["ok"] = bool(np.array_equal(to_uint32_numpy(full), want))
        ret["isolated"] = int((~has).sum())
        ret["shards"] = [hi_ - lo_ for lo_, hi_ in bounds]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_chunked_gather_into_row_slices_skipping_isolated_rows(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["ok"] and ret["isolated"] > 0


@pytest.mark.parametrize("world,share", [(2, 0.6), (3, 0.4), (3, 0.0)])
def test_chunked_gather_with_a_smaller_shard_for_the_assembling_rank(world, share):
    """bench.py --rank0-share: rank 0 walks fewer jobs (it also prefills and scatters); every shard's stream address
    follows from the bounds alone, so the assembled matrix is the single-stream matrix whatever the split."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gather_worker, args=(world, _free_port(), ret, share), nprocs=world, join=True)
    assert ret["ok"] and ret["isolated"] > 0
    shards = ret["shards"]
    assert shards[0] < min(shards[1:]) and max(shards[1:]) - min(shards[1:]) <= 1


def test_weighted_shard_bounds():
    from pecanpy_amd.engine import auto_rank0_share

    for n, w, sh in [(41943040, 8, 0.5), (10, 3, 0.4), (7, 8, 0.5), (100, 2, 0.9), (5, 4, 0.0), (1000, 4, 1.0), (1000, 1, 0.3)]:
        b = shard_bounds(n, w, sh)
        assert len(b) == w and b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1)) and all(hi >= lo for lo, hi in b)
        if w > 1 and sh < 1.0:
            assert b[0][1] - b[0][0] <= round(sh * n / w) + 1
            rest = [hi - lo for lo, hi in b[1:]]
            assert max(rest) - min(rest) <= 1
    assert shard_bounds(41943040, 8, 1.0) == shard_bounds(41943040, 8)
    assert auto_rank0_share(1) == 1.0 and auto_rank0_share(8) == 0.5 and 0.5 < auto_rank0_share(4) < auto_rank0_share(2) < 1.0
    assert auto_rank0_share(8, gather=False) == 1.0


def _uneven_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    indptr, indices, data = _graph("directed_with_sinks")
    L, seed = 12, 3
    starts = orc.shuffled_starts(indptr.size - 1, 3, seed)
    has = indptr[1:] != indptr[:-1]

    def run_shard(sl, skip):
        mat = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, sl, L, seed, stream_skip=skip)
        return torch.from_numpy(mat.view(np.int32).copy()), int((mat[:, -1].astype(np.int64) - 1).sum())

    bounds = shard_bounds(starts.size, world, 0.5)
    full = sharded_walk_matrix(run_shard, lambda sl: int(has[sl].sum()) * L, starts, L, dst=0, bounds=bounds)
    if rank == 0:
        ret["full"] = to_uint32_numpy(full)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_walk_matrix_with_uneven_bounds_and_dead_ends():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_uneven_worker, args=(3, _free_port(), ret), nprocs=3, join=True)
    indptr, indices, data = _graph("directed_with_sinks")
    starts = orc.shuffled_starts(indptr.size - 1, 3, 3)
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 12, 3)
    assert np.array_equal(ret["full"], want)
