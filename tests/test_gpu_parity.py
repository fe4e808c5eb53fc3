"""GPU parity tests: the HIP path through the C ABI vs the CPU oracle and the golden fixtures.

Bit-exact comparisons (integer node-id output).  Run with ``-m gpu`` on an MI355X.
"""
import glob
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from pecanpy_amd.engine import WalkEngine
from pecanpy_amd.synth import rmat_csr

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sparse_fixtures(extend):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        b = os.path.basename(f)
        if "_SparseOTF_" not in b:
            continue
        if ("n2vplus" in b) != extend:
            continue
        out.append(f)
    return out


def _diff_report(got, want):
    bad = np.nonzero((got != want).any(axis=1))[0]
    if bad.size == 0:
        return ""
    i = int(bad[0])
    j = int(np.nonzero(got[i] != want[i])[0][0])
    return f"{bad.size}/{got.shape[0]} walks differ; first: walk {i} step {j}: got {got[i, j]} want {want[i, j]}"


@pytest.mark.parametrize("path", _sparse_fixtures(False), ids=lambda f: os.path.basename(f)[:-4])
def test_golden_sparse_otf(path):
    z = np.load(path)
    eng = WalkEngine.from_csr(z["indptr"], z["indices"], z["data"])
    got = eng.simulate("SparseOTF", float(z["p"]), float(z["q"]), False, z["starts"],
                       int(z["walk_length"]), seed=int(z["seed"]))
    assert got.shape == z["walks"].shape
    assert np.array_equal(got, z["walks"]), _diff_report(got, z["walks"])


@pytest.mark.parametrize("path", _sparse_fixtures(True), ids=lambda f: os.path.basename(f)[:-4])
def test_golden_sparse_otf_node2vec_plus(path):
    z = np.load(path)
    eng = WalkEngine.from_csr(z["indptr"], z["indices"], z["data"])
    eng.set_thresholds(z["thr"])
    got = eng.simulate("SparseOTF", float(z["p"]), float(z["q"]), True, z["starts"],
                       int(z["walk_length"]), seed=int(z["seed"]))
    assert np.array_equal(got, z["walks"]), _diff_report(got, z["walks"])


def _thresholds(indptr, data, gamma):
    """Same NumPy expression as the reference (sparse_rw.py:22-35), via the host mirror."""
    from pecanpy_amd import pecanpy as node2vec

    g = node2vec.SparseOTF(gamma=gamma, extend=True)
    g.indptr, g.data = indptr, data
    g.set_node_ids(None, implicit_ids=True, num_nodes=indptr.size - 1)
    with np.errstate(all="ignore"):
        return g.get_noise_thresholds()


@pytest.mark.parametrize("gamma,p,q", [(0.0, 0.5, 2), (0.5, 0.5, 2), (0.0, 1.3, 0.4)])
def test_rmat_weighted_node2vec_plus_vs_oracle(gamma, p, q):
    indptr, indices, data = rmat_csr(11, seed=8, weighted=True)
    thr = _thresholds(indptr, data, gamma)
    thr = np.nan_to_num(thr, nan=0.0)  # isolated vertices (mean of an empty row); never read
    starts = orc.shuffled_starts(indptr.size - 1, 2, 5)
    want = orc.walks_sparse_otf(indptr, indices, data, p, q, starts, 30, 5, thr=thr)
    eng = WalkEngine.from_csr(indptr, indices, data)
    eng.set_thresholds(thr)
    got = eng.simulate("SparseOTF", p, q, True, starts, 30, seed=5)
    assert np.array_equal(got, want), _diff_report(got, want)


def test_node2vec_plus_on_unweighted_graph_equals_node2vec():
    indptr, indices, data = rmat_csr(10, seed=4)
    starts = orc.shuffled_starts(indptr.size - 1, 2, 1)
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 20, 1)
    eng = WalkEngine.from_csr(indptr, indices, data)
    eng.set_thresholds(np.ones(indptr.size - 1, dtype=np.float32))
    got = eng.simulate("SparseOTF", 0.5, 2, True, starts, 20, seed=1)
    assert np.array_equal(got, want)


def _check_vs_oracle(indptr, indices, data, p, q, num_walks, L, seed, stream_skip=0, job_slice=None):
    n = indptr.size - 1
    starts = orc.shuffled_starts(n, num_walks, seed)
    if job_slice is not None:
        starts = starts[job_slice]
    want, ost = orc.walks_sparse_otf(indptr, indices, data, p, q, starts, L, seed,
                                     stream_skip=stream_skip, return_stats=True)
    eng = WalkEngine.from_csr(indptr, indices, data)
    got = eng.simulate("SparseOTF", p, q, False, starts, L, seed=seed, stream_skip=stream_skip)
    assert np.array_equal(got, want), _diff_report(got, want)
    st = eng.last_stats
    assert st["total_steps"] == ost.total_steps
    assert st["overflow_reads"] == ost.overflow_reads
    return st


@pytest.mark.parametrize("scale,p,q", [(10, 0.5, 2), (12, 0.25, 4), (13, 2, 0.5), (12, 1, 1), (12, 1, 0.25),
                                       (12, 4, 0.125), (11, 0.0625, 16), (12, 8, 1), (11, 2.0 ** -40, 2.0 ** 40)])
def test_rmat_unweighted_vs_oracle(scale, p, q):
    indptr, indices, data = rmat_csr(scale, seed=scale)
    _check_vs_oracle(indptr, indices, data, p, q, 2, 40, seed=scale)


@pytest.mark.parametrize("p,q", [(0.5, 2), (0.3, 1.7), (3.0, 0.37)])
def test_rmat_weighted_vs_oracle(p, q):
    indptr, indices, data = rmat_csr(11, seed=5, weighted=True)
    _check_vs_oracle(indptr, indices, data, p, q, 2, 30, seed=9)


def test_unweighted_non_dyadic_pq():
    indptr, indices, data = rmat_csr(11, seed=6)
    _check_vs_oracle(indptr, indices, data, 0.3, 1.7, 2, 30, seed=2)


def _hub_graph(n_leaves, extra_edges, seed, weighted):
    """A few very high degree hubs (rows longer than one LDS mask segment) plus random edges."""
    rng = np.random.default_rng(seed)
    n = n_leaves + 3
    src = [np.full(n_leaves, 0), np.full(n_leaves // 2, 1), np.full(n_leaves // 3, 2)]
    dst = [np.arange(3, 3 + n_leaves), 3 + rng.choice(n_leaves, n_leaves // 2, replace=False),
           3 + rng.choice(n_leaves, n_leaves // 3, replace=False)]
    src.append(rng.integers(0, n, extra_edges))
    dst.append(rng.integers(0, n, extra_edges))
    src.append(np.array([0, 0, 1]))
    dst.append(np.array([1, 2, 2]))
    s, d = np.concatenate(src), np.concatenate(dst)
    keep = s != d
    s, d = s[keep], d[keep]
    from pecanpy_amd.synth import csr_from_edges, hash_edge_weights

    indptr, indices, data = csr_from_edges(np.concatenate([s, d]), np.concatenate([d, s]), n)
    if weighted:
        data = hash_edge_weights(indptr, indices, seed)
    return indptr, indices, data


@pytest.mark.parametrize("weighted", [False, True])
def test_hub_rows_longer_than_mask_segment(weighted):
    indptr, indices, data = _hub_graph(70000, 50000, 1, weighted)
    assert (np.diff(indptr.astype(np.int64)).max()) > 32768
    n = indptr.size - 1
    rng = np.random.default_rng(0)
    starts = np.concatenate([np.array([0, 1, 2] * 20), rng.integers(0, n, 400)]).astype(np.uint32)
    want, ost = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 12, 4, return_stats=True)
    eng = WalkEngine.from_csr(indptr, indices, data)
    got = eng.simulate("SparseOTF", 0.5, 2, False, starts, 12, seed=4)
    assert np.array_equal(got, want), _diff_report(got, want)
    assert eng.last_stats["overflow_reads"] == ost.overflow_reads


def _with_loops(indptr, indices, loops):
    from pecanpy_amd.synth import csr_from_edges

    n = indptr.size - 1
    rows = np.repeat(np.arange(n), np.diff(indptr.astype(np.int64)))
    return csr_from_edges(np.concatenate([rows, loops]), np.concatenate([indices.astype(np.int64), loops]), n)


@pytest.mark.parametrize("p,q,lane", [(0.5, 2, 1), (0.25, 4, 1), (2, 0.5, 1), (0.3, 1.7, 2)])
def test_self_loops_stay_on_the_lane_kernel(p, q, lane):
    """Self loops (the reference accepts them, graph.py:238-268; a walker may step u -> u and then has prev == cur): unit
    graphs keep the lane index (round 6: prev's own position is taken out of the lists whose source has a loop, as the
    reference takes it out of the common neighbours, sparse_rw.py:79-87) -- same kernels, no 25x cliff -- and match the oracle."""
    from pecanpy_amd.synth import csr_from_edges

    rng = np.random.default_rng(3)
    n = 3000
    s, d = rng.integers(0, n, 40000), rng.integers(0, n, 40000)
    loops = rng.choice(n, 300, replace=False)
    indptr, indices, data = csr_from_edges(np.concatenate([s, d, loops]), np.concatenate([d, s, loops]), n)
    rows = np.repeat(np.arange(n), np.diff(indptr.astype(np.int64)))
    assert (rows == indices).sum() >= 300
    st = _check_vs_oracle(indptr, indices, data, p, q, 2, 30, seed=8)
    assert st["lane_kernel"] == lane, st
    # R-MAT with loops on its hubs (long lists, overflow reads through looped vertices) and on every 7th vertex
    indptr, indices, _ = rmat_csr(13, seed=13)
    deg = np.diff(indptr.astype(np.int64))
    loops = np.unique(np.concatenate([np.argsort(deg)[-40:], np.arange(0, indptr.size - 1, 7)]))
    indptr, indices, data = _with_loops(indptr, indices, loops)
    st = _check_vs_oracle(indptr, indices, data, p, q, 4, 60, seed=5)
    assert st["lane_kernel"] == lane and st["total_steps"] > 10**6, st


def test_self_loops_on_every_vertex_and_in_directed_graphs():
    """A loop on EVERY vertex of a ring lattice (every arrival has prev among the raw common neighbours; the lists sweep
    the inline capacity), and a directed graph with loops, sinks and rows of one entry."""
    from pecanpy_amd.synth import csr_from_edges, ring_lattice_csr

    for k in (5, 12):
        indptr, indices, _ = ring_lattice_csr(600, k)
        indptr, indices, data = _with_loops(indptr, indices, np.arange(600))
        for p, q in ((0.5, 2), (4, 0.25), (1, 1)):
            st = _check_vs_oracle(indptr, indices, data, p, q, 3, 40, seed=2)
            assert st["lane_kernel"] == 1
    rng = np.random.default_rng(9)
    m = 2500
    src, dst = rng.integers(0, m, 30000), rng.integers(0, m, 30000)
    keep = (src % 40 != 0)                                   # vertices 0, 40, ... have no out-edges
    lp = rng.choice(np.arange(m)[np.arange(m) % 40 != 0], 400, replace=False)
    indptr, indices, data = csr_from_edges(np.concatenate([src[keep], lp]), np.concatenate([dst[keep], lp]), m)
    st = _check_vs_oracle(indptr, indices, data, 0.5, 2, 2, 30, seed=4)
    assert st["lane_kernel"] == 1 and st["dead_end_walks"] > 0 and st["stream_addressing"] == 0


def test_self_loops_weighted_graphs_take_the_wave_kernel():
    """Weighted graphs with self loops get no lane index (the node2vec+ tables pair the two directions' lists entry by
    entry): the wave-per-walk kernel's eager step serves them, exact as before."""
    indptr, indices, _ = rmat_csr(11, seed=4)
    indptr, indices, _ = _with_loops(indptr, indices, np.arange(0, indptr.size - 1, 5))
    from pecanpy_amd.synth import hash_edge_weights

    data = hash_edge_weights(indptr, indices, 3)
    st = _check_vs_oracle(indptr, indices, data, 0.5, 2, 2, 30, seed=6)
    assert st["lane_kernel"] == 0
    thr = np.nan_to_num(_thresholds(indptr, data, 0.0), nan=0.0)
    starts = orc.shuffled_starts(indptr.size - 1, 2, 5)
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 30, 5, thr=thr)
    eng = WalkEngine.from_csr(indptr, indices, data)
    eng.set_thresholds(thr)
    got = eng.simulate("SparseOTF", 0.5, 2, True, starts, 30, seed=5)
    assert np.array_equal(got, want), _diff_report(got, want)


def test_overflow_reads_are_mirrored():
    """choice == degree (float32 CDF short of 1) must read the next row's first neighbour
    exactly like the reference; RMAT-14 produces a few of those."""
    indptr, indices, data = rmat_csr(14, seed=14)
    st = _check_vs_oracle(indptr, indices, data, 0.5, 2, 4, 80, seed=0)
    assert st["total_steps"] > 10**6


def test_stream_skip_shard():
    indptr, indices, data = rmat_csr(11, seed=7)
    n = indptr.size - 1
    starts = orc.shuffled_starts(n, 3, 1)
    full = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 25, 1)
    cut = starts.size // 3 + 5
    skip = int((full[:cut, -1].astype(np.int64) - 1).sum())
    eng = WalkEngine.from_csr(indptr, indices, data)
    assert eng.count_stream_draws(starts[:cut], 25) == skip
    got = eng.simulate("SparseOTF", 0.5, 2, False, starts[cut:], 25, seed=1, stream_skip=skip)
    assert np.array_equal(got, full[cut:]), _diff_report(got, full[cut:])


def test_long_walks_and_large_offsets():
    """walk_length > 128 and a stream offset of 10^7 + 17 doubles (jump-ahead + expansion; the offsets the last
    shards of a BASELINE-size run start from -- 1.4e9 doubles and beyond -- are covered by
    test_device_stream_at_deep_offsets below and tests/test_gpu_scale.py::test_full_size_oracle_slice_deep_in_the_stream)."""
    indptr, indices, data = rmat_csr(9, seed=2)
    n = indptr.size - 1
    starts = orc.shuffled_starts(n, 1, 3)[:64]
    skip = 10**7 + 17
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 200, 3, stream_skip=skip)
    eng = WalkEngine.from_csr(indptr, indices, data)
    got = eng.simulate("SparseOTF", 0.5, 2, False, starts, 200, seed=3, stream_skip=skip)
    assert np.array_equal(got, want), _diff_report(got, want)


def test_device_stream_at_deep_offsets():
    """The device's jump-ahead tree + expansion kernels against the host jump-ahead (pw_mt_random_sample, itself pinned
    to NumPy's RandomState in tests/test_stream.py and the mt19937 goldens) at the stream depths of a multi-GPU
    BASELINE run and beyond 2^32 words: offsets 2^31 and 2.2e9 DOUBLES (= 2^32 and 4.4e9 MT19937 words), unaligned
    starts, spans that cross generator boundaries; and against NumPy itself where that is cheap."""
    import ctypes as C

    from pecanpy_amd import _lib

    indptr, indices, data = rmat_csr(8, seed=1)
    eng = WalkEngine.from_csr(indptr, indices, data)
    lib = _lib.load()
    for seed, off, n in ((0, 2**31, 5000), (0, 2**31 + 5, 700000), (7, 2_200_000_000, 300001), (3, 1_400_000_123, 4097),
                         (0, 1_607_000_000, 1000), (123456789, 4_000_000_001, 3000), (5, 0, 1000), (5, 311, 2)):
        got = eng.stream_sample(seed, off, n)
        want = np.zeros(n, dtype=np.float64)
        _lib.check(lib.pw_mt_random_sample(C.c_uint32(seed), C.c_uint64(off), C.c_uint64(n), want.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(got, want), (seed, off, n, int(np.flatnonzero(got != want)[0]))
    rs = np.random.RandomState(11)
    rs.random_sample(3_000_000)
    assert np.array_equal(eng.stream_sample(11, 3_000_000, 4000), rs.random_sample(4000))


@pytest.mark.parametrize("parts", [2, 3, 5, 8, 11])
def test_host_call_walked_in_parts_equals_one_call(parts, monkeypatch):
    """pw_simulate walks large job arrays in parts (the copy-out of one part under the kernels of the next); part k + 1
    is addressed into the stream by the draws the earlier parts actually consumed -- undirected graph and a directed
    one with dead ends, against the oracle."""
    # (from 8 parts on the device holds a RING of three part buffers instead of the matrix, and every part expands its own
    #  stretch of the stream -- round 6)
    monkeypatch.setenv("PECANPY_AMD_PARTS", str(parts))
    indptr, indices, data = rmat_csr(11, seed=7)
    starts = orc.shuffled_starts(indptr.size - 1, 3, 1)
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 25, 1)
    eng = WalkEngine.from_csr(indptr, indices, data)
    got = eng.simulate("SparseOTF", 0.5, 2, False, starts, 25, seed=1)
    assert np.array_equal(got, want), _diff_report(got, want)
    assert eng.last_stats["total_steps"] == int((want[:, -1].astype(np.int64) - 1).sum())
    z = np.load(os.path.join(GOLDEN, "sink_SparseOTF_p0.5_q2.npz"))
    eng = WalkEngine.from_csr(z["indptr"], z["indices"], z["data"])
    got = eng.simulate("SparseOTF", 0.5, 2, False, z["starts"], int(z["walk_length"]), seed=int(z["seed"]))
    assert np.array_equal(got, z["walks"]), _diff_report(got, z["walks"])
    assert eng.last_stats["stream_addressing"] == 0


def test_dense_bits_handle_ignores_padding_bits():
    """pw_dense_create_bits with garbage in the bits beyond column n - 1 of every row's last word: the handle
    clears them in its copy -- same walks as from clean rows, on the register-only kernel and on the complete one."""
    n = 1000                                             # 1000 & 63 = 40: 24 padding bits per row
    rs = np.random.RandomState(4)
    adj = rs.random_sample((n, n)) < 0.2
    adj = np.triu(adj, 1)
    adj = adj | adj.T
    clean = orc.pack_adjacency(adj)
    dirty = clean.copy()
    dirty[:, -1] |= np.uint64(0xFFFFFF) << np.uint64(40)
    starts = orc.shuffled_starts(n, 4, 1)
    want = orc.walks_dense_otf_bits(clean, n, 0.5, 2, starts, 30, 1)
    for p, q in ((0.5, 2.0), (0.3, 1.7)):
        a = WalkEngine.from_dense_bits(clean, n).simulate("DenseOTF", p, q, False, starts, 30, seed=1)
        b = WalkEngine.from_dense_bits(dirty, n).simulate("DenseOTF", p, q, False, starts, 30, seed=1)
        assert np.array_equal(a, b), (p, q)
        if (p, q) == (0.5, 2.0):
            assert np.array_equal(a, want)


def _dense_fixtures():
    return sorted(f for f in glob.glob(os.path.join(GOLDEN, "*.npz")) if "_DenseOTF_" in os.path.basename(f))


def _dense_from_csr(indptr, indices, data):
    n = indptr.size - 1
    mat = np.zeros((n, n), dtype=np.float64)
    for i in range(n):
        sl = slice(indptr[i], indptr[i + 1])
        mat[i, indices[sl]] = data[sl]
    return mat


@pytest.mark.parametrize("path", _dense_fixtures(), ids=lambda f: os.path.basename(f)[:-4])
def test_golden_dense_otf(path):
    z = np.load(path)
    eng = WalkEngine.from_dense(_dense_from_csr(z["indptr"], z["indices"], z["data"]))
    extend = bool(z["extend"])
    if extend:
        eng.set_thresholds(z["thr"])
    got = eng.simulate("DenseOTF", float(z["p"]), float(z["q"]), extend, z["starts"],
                       int(z["walk_length"]), seed=int(z["seed"]))
    assert np.array_equal(got, z["walks"]), _diff_report(got, z["walks"])


@pytest.mark.parametrize("weighted,extend,p,q", [(False, False, 0.5, 2), (False, False, 0.3, 1.7),
                                                  (True, False, 0.5, 2), (True, True, 0.5, 2),
                                                  (True, True, 1.5, 0.3)])
def test_dense_er_vs_oracle(weighted, extend, p, q):
    from pecanpy_amd.synth import er_dense_mask

    n = 700
    rng = np.random.default_rng(3)
    adj = er_dense_mask(n, 0.25, seed=2)
    adj[5, :] = False  # one isolated vertex
    adj[:, 5] = False
    mat = adj.astype(np.float64)
    if weighted:
        w = np.triu(rng.random((n, n)) + 0.1, 1)
        mat = mat * (w + w.T)
    thr = None
    if extend:
        thr = np.zeros(n, dtype=np.float32)
        for i in range(n):
            row = mat[i, adj[i]]
            thr[i] = row.mean() + 0.5 * row.std() if row.size else 0.0
    starts = orc.shuffled_starts(n, 2, 7)
    want, ost = orc.walks_dense_otf(mat, p, q, starts, 25, 7, thr=thr, return_stats=True)
    eng = WalkEngine.from_dense(mat)
    if extend:
        eng.set_thresholds(thr)
    got = eng.simulate("DenseOTF", p, q, extend, starts, 25, seed=7)
    assert np.array_equal(got, want), _diff_report(got, want)
    assert eng.last_stats["total_steps"] == ost.total_steps


def _pack_bits(adj):
    n = adj.shape[0]
    wpr = (n + 63) // 64
    padded = np.zeros((n, wpr * 64), dtype=np.uint8)
    padded[:, :n] = adj
    return np.packbits(padded, axis=1, bitorder="little").view(np.uint64).reshape(n, wpr)


@pytest.mark.parametrize("n,density,p,q", [(700, 0.25, 0.5, 2), (700, 0.25, 0.3, 1.7), (17000, 0.01, 0.5, 2)])
def test_dense_bits_kernel_vs_oracle(n, density, p, q):
    """Column-space DenseOTF kernel on packed adjacency rows (graph created without the float64
    matrix), incl. a graph wider than one 16384-column segment."""
    from pecanpy_amd.synth import er_dense_mask

    adj = er_dense_mask(n, density, seed=3)
    adj[7, :] = False
    adj[:, 7] = False
    starts = orc.shuffled_starts(n, 1, 5)[:1500]
    want, ost = orc.walks_dense_otf(adj.astype(np.float64), p, q, starts, 20, 5, return_stats=True)
    eng = WalkEngine.from_dense_bits(_pack_bits(adj), n)
    got = eng.simulate("DenseOTF", p, q, False, starts, 20, seed=5)
    assert np.array_equal(got, want), _diff_report(got, want)
    assert eng.last_stats["total_steps"] == ost.total_steps
    import torch

    dbits = torch.from_numpy(_pack_bits(adj).view(np.int64)).to("cuda:0")
    eng2 = WalkEngine.from_dense_bits(dbits, n)
    assert np.array_equal(eng2.simulate("DenseOTF", p, q, False, starts, 20, seed=5), want)


@pytest.mark.parametrize("p,q", [(0.5, 2), (2, 0.25), (1, 1)])
def test_dense_bits_exact_search_equals_float64_chain_kernel(p, q, monkeypatch):
    """The column-space kernel decides the CDF search in exact arithmetic; the compressed-row DenseOTF
    kernel runs the float64 chain.  Same graph, 1.2e6 transitions: identical walks.  (Round 6: a matrix handle takes the
    column-space kernels at every size; PECANPY_AMD_DENSE_SMALL_ROWS=1 is rounds 2-5's rule -- compressed rows up to 12 000 rows.)"""
    from pecanpy_amd.synth import er_dense_mask

    n = 5000
    adj = er_dense_mask(n, 0.2, seed=9)
    starts = orc.shuffled_starts(n, 3, 1)
    chain = WalkEngine.from_dense(adj.astype(np.float64))
    bits = WalkEngine.from_dense_bits(_pack_bits(adj), n)           # packed rows: column-space kernel
    monkeypatch.setenv("PECANPY_AMD_DENSE_SMALL_ROWS", "1")         # n <= 12000: compressed-row kernel
    a = chain.simulate("DenseOTF", p, q, False, starts, 80, seed=2)
    monkeypatch.delenv("PECANPY_AMD_DENSE_SMALL_ROWS")
    c = chain.simulate("DenseOTF", p, q, False, starts, 80, seed=2)   # the same handle through the column-space kernels
    assert np.array_equal(a, c)
    b = bits.simulate("DenseOTF", p, q, False, starts, 80, seed=2)
    assert np.array_equal(a, b), _diff_report(b, a)
    assert chain.last_stats["total_steps"] == bits.last_stats["total_steps"] > 10**6


def test_dense_and_sparse_agree_on_unweighted_graph():
    """reference test/test_walk.py:58-81: identical tables for SparseOTF and DenseOTF."""
    indptr, indices, data = rmat_csr(9, seed=11)
    starts = orc.shuffled_starts(indptr.size - 1, 2, 2)
    a = WalkEngine.from_csr(indptr, indices, data).simulate("SparseOTF", 0.5, 2, False, starts, 20, seed=2)
    b = WalkEngine.from_dense(_dense_from_csr(indptr, indices, data)).simulate("DenseOTF", 0.5, 2, False, starts, 20, seed=2)
    # float32 vs float64 chains can differ only through rounding at the sampled boundary: on this
    # graph the reference's two paths agree walk for walk, and so must ours
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 20, 2)
    assert np.array_equal(a, want)
    wantd = orc.walks_dense_otf(_dense_from_csr(indptr, indices, data), 0.5, 2, starts, 20, 2)
    assert np.array_equal(b, wantd)


def _alias_fixtures():
    names = ("_PreComp_", "_FirstOrderUnweighted_", "_PreCompFirstOrder_")
    return sorted(f for f in glob.glob(os.path.join(GOLDEN, "*.npz")) if any(n in os.path.basename(f) for n in names))


@pytest.mark.parametrize("path", _alias_fixtures(), ids=lambda f: os.path.basename(f)[:-4])
def test_golden_alias_and_first_order_modes(path):
    z = np.load(path)
    mode = str(z["mode"])
    eng = WalkEngine.from_csr(z["indptr"], z["indices"], z["data"])
    extend = bool(z["extend"])
    if extend:
        eng.set_thresholds(z["thr"])
    got = eng.simulate(mode, float(z["p"]), float(z["q"]), extend, z["starts"], int(z["walk_length"]),
                       seed=int(z["seed"]))
    assert np.array_equal(got, z["walks"]), _diff_report(got, z["walks"])


def test_precomp_tables_match_oracle_bitwise():
    indptr, indices, data = rmat_csr(8, seed=3, weighted=True)
    eng = WalkEngine.from_csr(indptr, indices, data)
    eng.precomp_build(0.5, 2, False, False)
    aip, aj, aq = eng.precomp_export(False)
    oip, oj, oq = orc.precomp_tables(indptr, indices, data, 0.5, 2)
    assert np.array_equal(aip, oip) and np.array_equal(aj, oj)
    assert np.array_equal(aq.view(np.uint32), oq.view(np.uint32))
    starts = orc.shuffled_starts(indptr.size - 1, 3, 4)
    want = orc.walks_precomp(indptr, indices, data, 0.5, 2, starts, 30, 4)
    got = eng.simulate("PreComp", 0.5, 2, False, starts, 30, seed=4)
    assert np.array_equal(got, want), _diff_report(got, want)
    for precomp in (False, True):
        want = orc.walks_first_order(indptr, indices, data, starts, 30, 4, precomp=precomp)
        got = eng.simulate("PreCompFirstOrder" if precomp else "FirstOrderUnweighted", 1, 1, False, starts, 30, seed=4)
        assert np.array_equal(got, want), _diff_report(got, want)


def _sink_heavy_graph():
    rng = np.random.default_rng(5)
    n = 400
    adj = rng.random((n, n)) < 0.02
    np.fill_diagonal(adj, False)
    adj[rng.choice(n, 160, replace=False), :] = False          # 40 % sinks
    indptr = np.zeros(n + 1, dtype=np.uint32)
    indptr[1:] = np.cumsum(adj.sum(1))
    indices = np.nonzero(adj)[1].astype(np.uint32)
    data = np.ones(indices.size, dtype=np.float32)
    return rng, n, indptr, indices, data


def test_sink_heavy_directed_graph_is_exact_by_default(monkeypatch):
    """Many mid-walk dead ends (40 % sinks): the single-stream semantics of the reference (pecanpy.py:198-206) is
    inherently sequential here -- after 32 whole-array re-addressing passes the engine goes on BLOCK-WISE, still exact:
    the walks are the oracle's, draw for draw, and stream_addressing stays 0.  Also through pw_simulate's parts."""
    rng, n, indptr, indices, data = _sink_heavy_graph()
    L, seed = 30, 2
    starts = orc.shuffled_starts(n, 8, seed)
    want, ost = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, L, seed, return_stats=True)
    eng = WalkEngine.from_csr(indptr, indices, data)
    got = eng.simulate("SparseOTF", 0.5, 2, False, starts, L, seed=seed)
    st = dict(eng.last_stats)
    assert st["stream_addressing"] == 0 and st["dead_end_walks"] > 0 and st["repair_rounds"] > 32, st
    assert np.array_equal(got, want), _diff_report(got, want)
    assert st["total_steps"] == ost.total_steps
    for blk in ("1", "64"):                                    # (window sizes of the block-wise phase: same walks)
        monkeypatch.setenv("PECANPY_AMD_REPAIR_BLOCK", blk)
        assert np.array_equal(eng.simulate("SparseOTF", 0.5, 2, False, starts, L, seed=seed), want), blk
    monkeypatch.delenv("PECANPY_AMD_REPAIR_BLOCK")
    monkeypatch.setenv("PECANPY_AMD_PARTS", "3")              # every part's addressing is exact: the split does not matter
    parts = eng.simulate("SparseOTF", 0.5, 2, False, starts, L, seed=seed)
    assert eng.last_stats["stream_addressing"] == 0
    assert np.array_equal(parts, want)
    # non-dyadic p, q and the wave-per-walk kernel take the same repair path
    want2 = orc.walks_sparse_otf(indptr, indices, data, 0.3, 1.7, starts, L, seed)
    assert np.array_equal(eng.simulate("SparseOTF", 0.3, 1.7, False, starts, L, seed=seed), want2)


def test_sink_heavy_repair_has_a_time_budget_and_ignores_value_zero_switches(monkeypatch):
    """the block-wise repair is bounded -- PECANPY_AMD_REPAIR_SECONDS (default 600) -- and fails with a message that
    names the ways out instead of running on silently; and the opt-in switches are read by VALUE: PECANPY_AMD_NOMINAL_STREAM=0
    leaves the exact addressing on."""
    from pecanpy_amd._lib import PwError

    rng, n, indptr, indices, data = _sink_heavy_graph()
    L, seed = 30, 2
    starts = orc.shuffled_starts(n, 8, seed)
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, L, seed)
    eng = WalkEngine.from_csr(indptr, indices, data)
    monkeypatch.setenv("PECANPY_AMD_NOMINAL_STREAM", "0")
    got = eng.simulate("SparseOTF", 0.5, 2, False, starts, L, seed=seed)
    assert eng.last_stats["stream_addressing"] == 0 and np.array_equal(got, want)
    monkeypatch.delenv("PECANPY_AMD_NOMINAL_STREAM")
    monkeypatch.setenv("PECANPY_AMD_REPAIR_SECONDS", "0.000001")
    monkeypatch.setenv("PECANPY_AMD_REPAIR_BLOCK", "1")
    with pytest.raises(PwError, match="PECANPY_AMD_REPAIR_SECONDS=0 lifts"):
        eng.simulate("SparseOTF", 0.5, 2, False, starts, L, seed=seed)
    monkeypatch.setenv("PECANPY_AMD_REPAIR_SECONDS", "0")     # no limit
    assert np.array_equal(eng.simulate("SparseOTF", 0.5, 2, False, starts, L, seed=seed), want)


def test_sink_heavy_directed_graph_nominal_slots_are_an_opt_in(monkeypatch):
    """PECANPY_AMD_NOMINAL_STREAM=1: after 32 passes one fixed slot of walk_length draws per walk (reported in the stats) --
    deterministic, every walk equals the oracle run of that walk alone at its slot; decided once for the whole array
    (pw_simulate does not split such a call into parts)."""
    rng, n, indptr, indices, data = _sink_heavy_graph()
    L, seed = 30, 2
    starts = orc.shuffled_starts(n, 8, seed)
    monkeypatch.setenv("PECANPY_AMD_NOMINAL_STREAM", "1")
    eng = WalkEngine.from_csr(indptr, indices, data)
    got = eng.simulate("SparseOTF", 0.5, 2, False, starts, L, seed=seed)
    st = eng.last_stats
    assert st["stream_addressing"] == 1 and st["dead_end_walks"] > 0
    assert st["total_steps"] == int((got[:, -1].astype(np.int64) - 1).sum())
    has = (indptr[1:] != indptr[:-1])[starts]
    slot = np.concatenate([[0], np.cumsum(has)[:-1]]) * L
    for i in rng.choice(starts.size, 60, replace=False):
        want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts[i:i + 1], L, seed, stream_skip=int(slot[i]))
        assert np.array_equal(got[i], want[0]), i
    again = eng.simulate("SparseOTF", 0.5, 2, False, starts, L, seed=seed)
    assert np.array_equal(again, got)
    monkeypatch.setenv("PECANPY_AMD_PARTS", "3")
    assert np.array_equal(eng.simulate("SparseOTF", 0.5, 2, False, starts, L, seed=seed), got)


def _sharded_worker(rank, world, port, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"            # both ranks share the one GPU of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pecanpy_amd import pecanpy as node2vec

    indptr, indices, data = rmat_csr(10, seed=6)
    g = node2vec.SparseOTF.from_csr(indptr, indices, data, p=0.5, q=2, random_state=4)
    mat = g.simulate_walks_array(3, 25)
    if rank == 0:
        ret["mat"] = mat
    dist.destroy_process_group()


def test_sharded_simulate_walks_two_ranks_one_gpu():
    """End-to-end multi-process path (shard bounds, stream_skip, device kernels, gather) with two ranks
    on the same GPU; communication over gloo because RCCL refuses two ranks on one device."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_sharded_worker, args=(2, port, ret), nprocs=2, join=True)
        indptr, indices, data = rmat_csr(10, seed=6)
        starts = orc.shuffled_starts(indptr.size - 1, 3, 4)
        want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 25, 4)
        assert np.array_equal(ret["mat"], want)


def test_unseeded_alias_modes_sample_the_reference_distribution():
    """random_state=None: nothing to reproduce bit for bit (the reference seeds itself from the OS),
    so the alias / first-order modes run all walks in parallel on per-walk counter-based draws.
    Check the sampled second-order transition frequencies against the oracle's probabilities."""
    rng = np.random.default_rng(11)
    n = 12
    w = np.triu(rng.random((n, n)) + 0.2, 1) * (rng.random((n, n)) < 0.6)
    w = (w + w.T).astype(np.float32)
    indptr = np.zeros(n + 1, dtype=np.uint32)
    indptr[1:] = np.cumsum((w != 0).sum(1))
    indices = np.nonzero(w)[1].astype(np.uint32)
    data = w[w != 0].astype(np.float32)
    eng = WalkEngine.from_csr(indptr, indices, data)
    starts = np.tile(np.arange(n, dtype=np.uint32), 20000)
    L = 6
    for mode, p, q in [("PreComp", 0.5, 2.0), ("PreCompFirstOrder", 1, 1), ("FirstOrderUnweighted", 1, 1)]:
        mat = eng.simulate(mode, p, q, False, starts, L, seed=None)
        assert mat[:, -1].min() == L + 1
        # transitions (prev=a, cur=b) -> next, for the most frequent (a, b)
        a, b, c = mat[:, 1], mat[:, 2], mat[:, 3]
        key = a.astype(np.int64) * n + b
        top = np.bincount(key).argmax()
        sel = key == top
        pa, pb = int(top // n), int(top % n)
        nb = indices[indptr[pb]:indptr[pb + 1]]
        counts = np.array([(c[sel] == x).sum() for x in nb], dtype=np.float64)
        if mode == "PreComp":
            probs = orc.sparse_probs(indptr, indices, data, p, q, pb, pa).astype(np.float64)
        elif mode == "PreCompFirstOrder":
            row = data[indptr[pb]:indptr[pb + 1]].astype(np.float64)
            probs = row / row.sum()
        else:
            probs = np.full(nb.size, 1.0 / nb.size)
        tot = counts.sum()
        assert tot > 2000
        sigma = np.sqrt(tot * probs * (1 - probs)) + 1.0
        assert np.all(np.abs(counts - tot * probs) < 6 * sigma), (mode, counts / tot, probs)
    # two unseeded runs differ
    m1 = eng.simulate("FirstOrderUnweighted", 1, 1, False, starts[:1000], L, seed=None)
    m2 = eng.simulate("FirstOrderUnweighted", 1, 1, False, starts[:1000], L, seed=None)
    assert not np.array_equal(m1, m2)


def test_mode_classes_drop_in():
    from pecanpy import pecanpy  # the alias package
    from ref_test_walk import IDS, MAT, WALKS

    g = pecanpy.SparseOTF.from_mat(MAT, IDS, p=1, q=1, random_state=0)
    assert g.simulate_walks(2, 3) == WALKS["SparseOTF"]
    g = pecanpy.DenseOTF.from_mat(MAT, IDS, p=1, q=1, random_state=0)
    assert g.simulate_walks(2, 3) == WALKS["DenseOTF"]
    for label, cls in [("PreComp", pecanpy.PreComp), ("FirstOrderUnweighted", pecanpy.FirstOrderUnweighted)]:
        g = cls.from_mat(MAT, IDS, p=1, q=1, random_state=0)
        assert g.simulate_walks(2, 3) == WALKS[label]
    g = pecanpy.PreComp.from_mat(MAT, IDS, p=1, q=1, random_state=0)
    g.preprocess_transition_probs()
    assert g.alias_j.size == int((np.diff(g.indptr.astype(np.int64)) ** 2).sum())
