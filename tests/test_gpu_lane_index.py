"""The lane index itself (csrc/walk_lanes.hip.h: one set intersection per adjacent pair, row of the larger endpoint in
LDS, both lists written at once; lists of up to 20 uint16 positions inside the 64-byte edge line, longer ones and rows
beyond 65536 entries in the overflow array), decoded by pw_lane_index_export and compared with a NumPy restatement of
its definition: for every CSR entry e = (u -> v), the ascending positions in row v of N(u) & N(v) -- the set the
reference's isnotin() recomputes on every step (src/pecanpy/rw/sparse_rw.py:142-230) -- less u's own position when u has a
self loop (the reference takes prev out of the common neighbours, sparse_rw.py:79-87)."""
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
        # (self loop at u: prev's own position in row v is not a common neighbour -- the reference takes it out,
        #  `non_com_nbr[prev_ptr] = False`, sparse_rw.py:79-87)
        pos = pos[pos != rev]
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


def test_self_loops_keep_the_index_and_lose_prev(monkeypatch):
    """Round 6: graphs with self loops (the reference accepts them, graph.py:238-268) keep the lane index.  u in N(u) makes u a
    member of N(u) & N(v) for every neighbour v: its position leaves the list of (u -> v); a loop at v is an ordinary common
    neighbour; the self entry (u -> u) lists the whole row but u.  Ring lattices with a loop on EVERY vertex sweep the list
    lengths through the inline capacity (21 -> 20 entries: the list moves into its line), a hub with a loop gives uint32
    lists and segments, a directed graph gives rows of ONE entry that point at a looped vertex."""
    rng = np.random.default_rng(12)
    for k in (3, 11, 12):
        indptr, indices, _ = ring_lattice_csr(400, k)
        n = indptr.size - 1
        rows = np.repeat(np.arange(n), np.diff(indptr.astype(np.int64)))
        loops = np.arange(n) if k != 3 else np.arange(0, n, 3)
        indptr, indices, _ = csr_from_edges(np.concatenate([rows, loops]), np.concatenate([indices.astype(np.int64), loops]), n)
        _, n_in = check(indptr, indices)
        if k == 12:
            assert (n_in == 20).any() and (n_in == 21).any()
    indptr, indices, _ = rmat_csr(12, seed=5)
    n = indptr.size - 1
    rows = np.repeat(np.arange(n), np.diff(indptr.astype(np.int64)))
    loops = rng.choice(n, 400, replace=False)
    loops[:3] = np.argsort(np.diff(indptr.astype(np.int64)))[-3:]             # ... the three largest hubs among them
    indptr, indices, _ = csr_from_edges(np.concatenate([rows, loops]), np.concatenate([indices.astype(np.int64), loops]), n)
    check(indptr, indices)
    # a 70 000-entry hub with a loop (uint32 positions, nine segments), a 20 000-entry one (uint16, three segments)
    m = 90000
    a = np.arange(2, 70002)
    b = rng.choice(np.arange(2, m), 20000, replace=False)
    bg_s, bg_d = rng.integers(2, m, 250000), rng.integers(2, m, 250000)
    s = np.concatenate([np.zeros(a.size, np.int64), np.ones(b.size, np.int64), [0], bg_s])
    d = np.concatenate([a, b, [1], bg_d])
    keep = s != d
    s, d = s[keep], d[keep]
    lp = np.concatenate([[0, 1], rng.choice(np.arange(2, m), 500, replace=False)])
    indptr, indices, _ = csr_from_edges(np.concatenate([s, d, lp]), np.concatenate([d, s, lp]), m)
    ip = indptr.astype(np.int64)
    eng, _ = check(indptr, indices, sample=3000, seed=3)
    n_in, rev, off, ent = eng.lane_index()
    sel = np.concatenate([np.arange(ip[0], ip[0] + 200), np.arange(ip[1], ip[1] + 200), np.flatnonzero(indices == 0)[:200],
                          np.flatnonzero((indices == np.repeat(np.arange(m), np.diff(ip))))[:300]])     # (incl. the self entries)
    for e, (cnt, rv, pos) in zip(sel, expected_lists(indptr, indices, sel)):
        assert n_in[e] == cnt and rev[e] == rv and np.array_equal(ent[off[e]:off[e + 1]], pos), int(e)
    # directed: h -> k only (rows of one entry), k looped
    src = np.concatenate([np.arange(100, 400), np.arange(0, 100), rng.integers(0, 100, 600)])
    dst = np.concatenate([np.arange(100, 400) % 100, np.arange(0, 100), rng.integers(0, 100, 600)])
    indptr, indices, _ = csr_from_edges(src, dst, 400)
    assert (np.diff(indptr.astype(np.int64))[100:] == 1).all()
    _, n_in = check(indptr, indices)
    assert (n_in[indptr[100]:] == 1).all()       # (h -> k: k's own position in row k is the one common neighbour)


def test_directed_graph_entries_without_reverse_edge():
    rng = np.random.default_rng(8)
    m = 3000
    src, dst = rng.integers(0, m, 40000), rng.integers(0, m, 40000)
    keep = (src != dst) & (src % 50 != 0)          # vertices 0, 50, 100, ... have no out-edges (dead ends)
    indptr, indices, _ = csr_from_edges(src[keep], dst[keep], m)
    eng, n_in = check(indptr, indices)
    _, rev, _, _ = eng.lane_index()
    assert (rev == NOT_FOUND).any() and (rev != NOT_FOUND).any()


def test_directed_entry_into_a_wide_row_without_reverse_edge():
    """a directed entry h -> k with NO reverse entry, deg(h) <= 8192 (single segment), deg(k) > 65536
    (uint32 positions) and 1..20 common neighbours.  Its list is short enough for an edge line but a wide row's list
    never lives there: the FILL pass has to write it (the COUNT pass's inline shortcut does not apply).  Checked through
    the decoded index AND through walks that arrive by such entries, against the oracle."""
    from oracle import pyoracle as orc

    rng = np.random.default_rng(11)
    hub_deg, n = 70000, 70400
    hub_dst = np.arange(1, hub_deg + 1)                       # 0 -> 1..70000: a row of 70 000 entries
    feeders = np.arange(hub_deg + 1, hub_deg + 301)           # 300 vertices h -> 0, none of them a neighbour of 0
    f_src, f_dst = [], []
    for h in feeders:
        k = int(rng.integers(1, 21))                          # 1..20 common neighbours with the hub
        common = rng.choice(np.arange(1, hub_deg + 1), k, replace=False)
        common[0] = hub_deg - int(rng.integers(0, 3000))      # ... one of them beyond position 65535
        other = rng.choice(feeders[feeders != h], 3, replace=False)
        for t in np.concatenate([[0], common, other]):
            f_src.append(h); f_dst.append(int(t))
    # the hub's neighbours lead back to the feeders and to each other, so that walks keep arriving by h -> 0
    b_src = rng.integers(1, hub_deg + 1, 200000)
    b_dst = np.where(rng.random(200000) < 0.5, rng.choice(feeders, 200000), rng.integers(1, hub_deg + 1, 200000))
    every = np.arange(1, hub_deg + 1)                         # ... and every one of them has an out-edge: no dead ends
    src = np.concatenate([np.zeros(hub_deg, np.int64), np.array(f_src), b_src, every])
    dst = np.concatenate([hub_dst, np.array(f_dst), b_dst, feeders[every % feeders.size]])
    keep = src != dst
    indptr, indices, _ = csr_from_edges(src[keep], dst[keep], n)
    ip = indptr.astype(np.int64)
    assert ip[1] - ip[0] == hub_deg
    eng = WalkEngine.from_csr(indptr, indices, None)
    n_in, rev, off, ent = eng.lane_index()
    into_hub = np.concatenate([np.flatnonzero(indices[ip[h]:ip[h + 1]] == 0) + ip[h] for h in feeders])
    assert into_hub.size == feeders.size and (rev[into_hub] == NOT_FOUND).all()
    assert (n_in[into_hub] >= 1).all() and (n_in[into_hub] <= 20).all()
    for e, (cnt, rv, pos) in zip(into_hub, expected_lists(indptr, indices, into_hub)):
        assert n_in[e] == cnt and np.array_equal(ent[off[e]:off[e + 1]], pos), (int(e), ent[off[e]:off[e + 1]], pos)
        assert pos.max() > 65535
    some = np.sort(np.random.default_rng(2).choice(indices.size, 3000, replace=False))
    for e, (cnt, rv, pos) in zip(some, expected_lists(indptr, indices, some)):
        assert n_in[e] == cnt and rev[e] == rv and np.array_equal(ent[off[e]:off[e + 1]], pos), int(e)
    # walks from the feeders: the second step of most of them is taken on the hub row having arrived by h -> 0
    starts = np.repeat(feeders.astype(np.uint32), 40)
    np.random.RandomState(4).shuffle(starts)
    data = np.ones(indices.size, dtype=np.float32)
    for p, q in ((0.5, 2.0), (0.3, 1.7)):
        want, ost = orc.walks_sparse_otf(indptr, indices, data, p, q, starts, 30, 9, return_stats=True)
        got = eng.simulate("SparseOTF", p, q, False, starts, 30, seed=9)
        st = dict(eng.last_stats)
        assert st["lane_kernel"] in (1, 2) and st["stream_addressing"] == 0, st
        assert np.array_equal(got, want), (p, q)
        via = (want[:, 1] == 0).sum()
        assert via > 1000                                     # (walks whose first step went h -> 0)


def test_logged_build_equals_the_two_pass_build(monkeypatch):
    """Round 5 (PECANPY_AMD_INDEX_LOGGED=1): the COUNT pass logs its matches and lane_scatter_kernel copies lists and pivots to their places (one
    intersection per pair instead of two).  The decoded index and the walks over it equal those of the COUNT + FILL build
    (the default) -- on an R-MAT graph and on the hub graph with multi-segment rows and uint32 positions."""
    graphs = [rmat_csr(13, seed=3)[:2]]
    rng = np.random.default_rng(4)
    m, hub_deg = 90000, 70000                                # one row beyond 65536 entries (64-bit log words), rows beyond 8192 (segments)
    src = np.concatenate([np.zeros(hub_deg, dtype=np.int64), np.ones(20000, dtype=np.int64), rng.integers(2, m, 400000)])
    dst = np.concatenate([rng.choice(np.arange(2, m), hub_deg, replace=False), rng.choice(np.arange(2, m), 20000, replace=False),
                          rng.integers(2, m, 400000)])
    keep = src != dst
    graphs.append(csr_from_edges(np.concatenate([src[keep], dst[keep]]), np.concatenate([dst[keep], src[keep]]), m)[:2])
    for indptr, indices in graphs:
        n = indptr.size - 1
        starts = np.random.default_rng(1).integers(0, n, 20000).astype(np.uint32)
        starts[:64] = 0
        monkeypatch.setenv("PECANPY_AMD_INDEX_LOGGED", "1")
        logged = WalkEngine.from_csr(indptr, indices, None)
        monkeypatch.delenv("PECANPY_AMD_INDEX_LOGGED")
        a = logged.lane_index()
        wa = logged.simulate("SparseOTF", 0.5, 2, False, starts, 30, seed=3)
        twopass = WalkEngine.from_csr(indptr, indices, None)
        b = twopass.lane_index()
        wb = twopass.simulate("SparseOTF", 0.5, 2, False, starts, 30, seed=3)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert logged.last_stats["lane_kernel"] == 1 and np.array_equal(wa, wb)   # (the pivots are exercised by the walks' searches)


def test_partial_index_under_a_byte_budget(monkeypatch):
    """PECANPY_AMD_INDEX_BUDGET forcing a half-stored index (VERDICT r03 missing #2): edge lines always, the LONGEST lists
    left out; a step that arrives by an entry without its list is decided by one wavefront (lanes_eager_kernel) and the
    walk stays in the lane kernel -- bit-exact against the oracle, and equal to the fully indexed engine at RMAT-16."""
    import torch

    from oracle import pyoracle as orc

    indptr, indices, data = rmat_csr(13, seed=3)
    full = WalkEngine.from_csr(indptr, indices, None)
    n_in, rev, off, ent = full.lane_index()
    deg_v = np.diff(indptr.astype(np.int64))[indices]
    outside = ~((deg_v <= 65536) & (n_in <= 20))                      # lists that live in the overflow array
    units = ((n_in[outside].astype(np.int64) * 2 + 15) // 16)
    budget = int(units.sum() * 16 // 2)
    monkeypatch.setenv("PECANPY_AMD_INDEX_BUDGET", str(budget))
    part = WalkEngine.from_csr(indptr, indices, None)
    monkeypatch.delenv("PECANPY_AMD_INDEX_BUDGET")
    assert part.index_info()["index_bytes"] < full.index_info()["index_bytes"]
    n2, rev2, off2, ent2 = part.lane_index()
    assert np.array_equal(n2, n_in) and np.array_equal(rev2, rev)     # the records are all there
    lens = np.diff(off)
    dropped = np.array([lens[e] > 0 and (ent2[off[e]:off[e + 1]] == NOT_FOUND).all() for e in range(n_in.size)])
    assert dropped.any() and not dropped[~outside].any()
    t_max = lens[~dropped].max()
    assert lens[dropped].min() > t_max >= 20                          # the longest lists went, by a length threshold
    kept = np.flatnonzero(~dropped)
    for e in kept[:: max(1, kept.size // 3000)]:
        assert np.array_equal(ent2[off[e]:off[e + 1]], ent[off[e]:off[e + 1]]), int(e)
    stored_units = ((lens[outside & ~dropped].astype(np.int64) * 2 + 15) // 16).sum()
    assert stored_units * 16 <= budget
    starts = orc.shuffled_starts(indptr.size - 1, 4, 2)
    ones = np.ones(indices.size, dtype=np.float32)
    for p, q in ((0.5, 2.0), (0.25, 4.0), (0.3, 1.7)):
        want, ost = orc.walks_sparse_otf(indptr, indices, ones, p, q, starts, 40, 2, return_stats=True)
        got = part.simulate("SparseOTF", p, q, False, starts, 40, seed=2)
        st = dict(part.last_stats)
        assert np.array_equal(got, want), (p, q)
        assert st["lane_kernel"] in (1, 2) and st["index_max_list"] == t_max and st["overflow_reads"] == ost.overflow_reads
        # (a job array this small runs the in-place form without a queue -- and the FLOATS form never parks: both hand
        #  such walks to the wave kernel at that step; the queueing rounds below park them for lanes_eager_kernel)
        assert st["eager_steps"] + st["redo_walks"] > 0, st
    # a larger graph, queueing rounds: the partial engine against the fully indexed one
    indptr, indices, data = rmat_csr(16, seed=1)
    full = WalkEngine.from_csr(indptr, indices, None)
    monkeypatch.setenv("PECANPY_AMD_INDEX_BUDGET", str(full.index_info()["index_bytes"] // 12))
    part = WalkEngine.from_csr(indptr, indices, None)
    monkeypatch.delenv("PECANPY_AMD_INDEX_BUDGET")
    starts = np.concatenate([np.arange(indptr.size - 1, dtype=np.uint32)] * 10)
    np.random.RandomState(0).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    a = full.simulate_device("SparseOTF", 0.5, 2, False, d_starts, 80, seed=1)
    sa = dict(full.last_stats)
    b = part.simulate_device("SparseOTF", 0.5, 2, False, d_starts, 80, seed=1)
    sb = dict(part.last_stats)
    assert sa["eager_steps"] == 0 and sa["index_max_list"] == 0xFFFFFFFF
    assert sb["eager_steps"] > 0 and sb["lane_kernel"] == 1 and sb["index_max_list"] < 0xFFFFFFFF
    assert (sa["total_steps"], sa["overflow_reads"]) == (sb["total_steps"], sb["overflow_reads"])
    assert torch.equal(a, b)
    print(f"[partial] RMAT-16: lists up to {sb['index_max_list']} entries stored, {sb['eager_steps']} of {sb['total_steps']} steps eager, "
          f"{sb['lane_rounds']} rounds, {sb['walk_kernel_ms']:.1f} ms vs {sa['walk_kernel_ms']:.1f} ms fully indexed")
