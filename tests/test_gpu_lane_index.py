"""The lane index itself (csrc/walk_lanes.hip.h: one set intersection per adjacent pair, row of the larger endpoint in
LDS, both lists written at once; lists of up to 20 uint16 positions inside the 64-byte edge line, longer ones and rows
beyond 65536 entries in the overflow array), decoded by pw_lane_index_export and compared with a NumPy restatement of
its definition: for every CSR entry e = (u -> v), the ascending positions in row v of N(u) & N(v) -- the set the
reference's isnotin() recomputes on every step (src/pecanpy/rw/sparse_rw.py:142-230)."""
import numpy as np
import pytest

from pecanpy_amd.engine import WalkEngine
from pecanpy_amd.synth import csr_from_edges, holme_kim_csr, ring_lattice_csr, rmat_csr

pytestmark = pytest.mark.gpu
NOT_FOUND = 0xFFFFFFFF


def expected_lists(indptr, indices, entries):
    """(n_in, rev_pos, list) of the given CSR entries by the definition."""
    ip = indptr.astype(np.int64)
    rows = np.searchsorted(ip, entries, side="right") - 1
    out = []
    for e, u in zip(entries, rows):
        v = int(indices[e])
        ru, rv = indices[ip[u]:ip[u + 1]], indices[ip[v]:ip[v + 1]]
        pos = np.flatnonzero(np.isin(rv, ru, assume_unique=True))
        k = np.searchsorted(rv, u)
        rev = int(k) if k < rv.size and rv[k] == u else NOT_FOUND
        out.append((pos.size, rev, pos.astype(np.uint32)))
    return out


def check(indptr, indices, sample=None, seed=0):
    eng = WalkEngine.from_csr(indptr, indices, None)
    n_in, rev, off, ent = eng.lane_index()
    nnz = indices.size
    assert off[-1] == eng.index_info()["lane_list_entries"] == ent.size
    entries = np.arange(nnz) if sample is None or sample >= nnz else np.sort(np.random.default_rng(seed).choice(nnz, sample, replace=False))
    bad = []
    for e, (cnt, rv, pos) in zip(entries, expected_lists(indptr, indices, entries)):
        got = ent[off[e]:off[e + 1]]
        if n_in[e] != cnt or rev[e] != rv or not np.array_equal(got, pos):
            bad.append((int(e), int(n_in[e]), cnt, int(rev[e]), rv, got[:6].tolist(), pos[:6].tolist()))
    assert not bad, bad[:5]
    # lists are ascending and inside the row they index
    deg_v = np.diff(indptr.astype(np.int64))[indices]
    assert (n_in <= deg_v).all()
    return eng, n_in


def test_lists_of_rmat_graphs_match_the_definition():
    for scale in (8, 11, 13):
        indptr, indices, _ = rmat_csr(scale, seed=scale)
        check(indptr, indices, sample=None if scale < 13 else 20000)


def test_lists_with_every_length_around_the_inline_capacity():
    """Ring lattices: list lengths 2k - t - 1 sweep through 20 (the inline capacity of an edge line) for k = 12."""
    for k in (3, 11, 12, 40):
        indptr, indices, _ = ring_lattice_csr(400, k)
        _, n_in = check(indptr, indices)
        if k == 12:
            assert n_in.min() < 20 < n_in.max() and (n_in == 20).any() and (n_in == 21).any()


def test_lists_of_a_clustered_power_law_graph():
    indptr, indices, _ = holme_kim_csr(1 << 13, 8, 0.8, seed=2)
    check(indptr, indices, sample=30000)


def test_rows_split_into_segments_and_rows_with_uint32_positions():
    """A 70 000-entry hub (nine 8192-position segments, uint32 list entries), a 20 000-entry hub (three segments, uint16
    entries), their mutual edge, and a sparse random background that gives them thousands of common neighbours."""
    rng = np.random.default_rng(5)
    n = 90000
    a = np.arange(2, 70002)
    b = rng.choice(np.arange(2, n), 20000, replace=False)
    bg_s, bg_d = rng.integers(2, n, 250000), rng.integers(2, n, 250000)
    s = np.concatenate([np.zeros(a.size, np.int64), np.ones(b.size, np.int64), [0], bg_s])
    d = np.concatenate([a, b, [1], bg_d])
    keep = s != d
    s, d = s[keep], d[keep]
    indptr, indices, _ = csr_from_edges(np.concatenate([s, d]), np.concatenate([d, s]), n)
    deg = np.diff(indptr.astype(np.int64))
    assert deg[0] > 65536 and 8192 < deg[1] <= 65536
    eng, n_in = check(indptr, indices, sample=4000, seed=1)
    # every entry of the two hub rows and of the rows pointing at them (the uint32 lists)
    ip = indptr.astype(np.int64)
    hub_entries = np.concatenate([np.arange(ip[0], ip[0] + 300), np.arange(ip[1], ip[1] + 300)])
    into_hub0 = np.flatnonzero(indices == 0)[:300]
    into_hub1 = np.flatnonzero(indices == 1)[:300]
    _, _, off, ent = eng.lane_index()
    n_in2, rev, _, _ = eng.lane_index()
    for e, (cnt, rv, pos) in zip(np.concatenate([hub_entries, into_hub0, into_hub1]),
                                 expected_lists(indptr, indices, np.concatenate([hub_entries, into_hub0, into_hub1]))):
        assert n_in2[e] == cnt and rev[e] == rv and np.array_equal(ent[off[e]:off[e + 1]], pos), e


def test_directed_graph_entries_without_reverse_edge():
    rng = np.random.default_rng(8)
    m = 3000
    src, dst = rng.integers(0, m, 40000), rng.integers(0, m, 40000)
    keep = (src != dst) & (src % 50 != 0)          # vertices 0, 50, 100, ... have no out-edges (dead ends)
    indptr, indices, _ = csr_from_edges(src[keep], dst[keep], m)
    eng, n_in = check(indptr, indices)
    _, rev, _, _ = eng.lane_index()
    assert (rev == NOT_FOUND).any() and (rev != NOT_FOUND).any()
