"""Host-side mirror of the reference API: graph containers, start array, thresholds, CLI."""
import os

import numpy as np
import pytest

from pecanpy_amd import cli, graph
from pecanpy_amd import pecanpy as node2vec
from pecanpy_amd.synth import csr_from_edges, rmat_csr
from pecanpy_amd.wrappers import Timer


def write_edg(tmp_path, lines):
    p = tmp_path / "g.edg"
    p.write_text("\n".join(lines) + "\n")
    return str(p)


def test_edge_list_reader_builds_sorted_csr(tmp_path):
    path = write_edg(tmp_path, ["b\ta\t2.0", "a\tc\t0.5", "d\tb\t1.5", "c\tb\t4"])
    g = graph.SparseGraph()
    g.read_edg(path, weighted=True, directed=False)
    assert g.nodes == ["b", "a", "c", "d"]          # first-appearance order
    assert g.indptr.dtype == np.uint32 and g.indices.dtype == np.uint32 and g.data.dtype == np.float32
    for i in range(g.num_nodes):
        row = g.indices[g.indptr[i]:g.indptr[i + 1]]
        assert np.all(np.diff(row.astype(np.int64)) > 0)   # ascending, duplicate free
    assert g.num_edges == 8
    dense = graph.DenseGraph()
    dense.read_edg(path, weighted=True, directed=False)
    assert dense.data.dtype == np.float64 and dense.nonzero.dtype == bool
    assert dense.data[0, 1] == 2.0 and dense.data[1, 0] == 2.0 and dense.num_edges == 8


def test_directed_unweighted_and_nonpositive_weights(tmp_path):
    path = write_edg(tmp_path, ["a\tb", "b\tc", "c\ta"])
    g = graph.SparseGraph()
    g.read_edg(path, weighted=False, directed=True)
    assert g.num_edges == 3 and np.all(g.data == 1.0)
    path = write_edg(tmp_path, ["a\tb\t0", "a\tc\t-1", "b\tc\t1"])
    with pytest.warns(RuntimeWarning):
        g.read_edg(path, weighted=True, directed=False)
    assert g.num_edges == 2
    with pytest.raises(ValueError):
        graph.AdjlstGraph._read_edge_line("a\tb", True, "\t")


def test_from_mat_npz_roundtrip_and_implicit_ids(tmp_path):
    mat = np.array([[0, 1, 0, 0], [1, 0, 0, 1], [0, 0, 0, 0], [0, 1, 1, 0]], dtype=float)
    g = graph.SparseGraph.from_mat(mat, list("abcd"))
    assert g.indptr.tolist() == [0, 1, 3, 3, 5] and g.indices.tolist() == [1, 0, 3, 1, 2]
    p = str(tmp_path / "g.csr.npz")
    g.save(p)
    h = graph.SparseGraph()
    h.read_npz(p, weighted=True)
    assert h.nodes == g.nodes and np.array_equal(h.indices, g.indices)
    np.savez(str(tmp_path / "raw.npz"), indptr=g.indptr, indices=g.indices, data=g.data)
    k = graph.SparseGraph()
    with pytest.warns(UserWarning):
        k.read_npz(str(tmp_path / "raw.npz"), weighted=False)
    assert k.nodes == ["0", "1", "2", "3"]
    k.read_npz(str(tmp_path / "raw.npz"), weighted=False, implicit_ids=True)
    d = graph.DenseGraph.from_mat(mat, list("abcd"))
    d.save(str(tmp_path / "d.npz"))
    e = graph.DenseGraph()
    e.read_npz(str(tmp_path / "d.npz"), weighted=False)
    assert np.array_equal(e.nonzero, mat != 0)
    with pytest.raises(NotImplementedError):
        graph.BaseGraph().num_edges


def test_start_array_is_the_reference_shuffle():
    g = node2vec.SparseOTF(random_state=5)
    g.set_node_ids(None, implicit_ids=True, num_nodes=11)
    starts = g._start_array(3)
    nodes = np.arange(11, dtype=np.uint32)
    want = np.concatenate([nodes] * 3)
    np.random.seed(5)
    np.random.shuffle(want)
    assert starts.dtype == np.uint32 and np.array_equal(starts, want)


def test_constructor_signature_and_mode_classes():
    for cls in (node2vec.SparseOTF, node2vec.DenseOTF, node2vec.PreComp, node2vec.PreCompFirstOrder,
                node2vec.FirstOrderUnweighted):
        g = cls(0.5, 2, 3, True, True, 0.25, 9)     # the positional order the CLI relies on
        assert (g.p, g.q, g.workers, g.verbose, g.extend, g.gamma, g.random_state) == (0.5, 2, 3, True, True, 0.25, 9)
        assert isinstance(g, node2vec.Base)
    assert hasattr(node2vec.PreComp(), "alias_indptr")


def test_noise_thresholds_follow_the_reference_expression():
    indptr, indices, data = rmat_csr(7, seed=2, weighted=True)
    keep = np.diff(indptr.astype(np.int64)) > 0
    g = node2vec.SparseOTF.from_csr(indptr, indices, data, gamma=0.5, extend=True)
    with np.errstate(all="ignore"):
        thr = g.get_noise_thresholds()
    for i in np.nonzero(keep)[0][:20]:
        row = data[indptr[i]:indptr[i + 1]]
        assert thr[i] == np.float32(max(row.mean() + 0.5 * row.std(), 0))


def test_map_walk_uses_length_cell():
    g = node2vec.SparseOTF()
    g.set_node_ids(list("abcde"))
    assert g._map_walk(np.array([2, 3, 0, 0, 2], dtype=np.uint32)) == ["c", "d"]


def test_cli_flags_defaults_and_mode_checks():
    a = cli.parse_args(["--input", "g.edg", "--output", "o.emb"])
    assert (a.mode, a.p, a.q, a.num_walks, a.walk_length, a.dimensions, a.window_size, a.epochs) == \
        ("SparseOTF", 1, 1, 10, 80, 128, 10, 1)
    assert a.workers == 0 and a.random_state is None and a.delimiter == "\t" and not a.extend
    g = node2vec.SparseOTF()
    bad = cli.parse_args(["--input", "g", "--output", "o", "--mode", "FirstOrderUnweighted", "--p", "2"])
    with pytest.raises(ValueError):
        cli.check_mode(g, bad)
    bad = cli.parse_args(["--input", "g", "--output", "o", "--mode", "PreCompFirstOrder", "--q", "2", "--weighted"])
    with pytest.raises(ValueError):
        cli.check_mode(g, bad)
    both = cli.parse_args(["--input", "g", "--output", "o", "--directed", "--extend"])
    with pytest.raises(NotImplementedError):
        cli.read_graph.__wrapped__(both) if hasattr(cli.read_graph, "__wrapped__") else cli.read_graph(both)


def test_cli_conversion_tasks(tmp_path):
    path = write_edg(tmp_path, ["a\tb", "b\tc"])
    out = str(tmp_path / "o.csr.npz")
    args = cli.parse_args(["--input", path, "--output", out, "--task", "tocsr"])
    with pytest.raises(SystemExit):
        cli.read_graph(args)
    assert set(np.load(out).files) == {"IDs", "data", "indptr", "indices"}


def test_timer_prints_stage(capsys):
    assert Timer("do thing")(lambda x: x + 1)(1) == 2
    assert "to do thing" in capsys.readouterr().out
    assert Timer("quiet", verbose=False)(len) is len


def test_synth_generators():
    indptr, indices, data = rmat_csr(8, seed=1)
    assert indptr[-1] == indices.size and np.all(data == 1)
    rows = np.repeat(np.arange(256), np.diff(indptr.astype(np.int64)))
    assert np.all(rows != indices)                                   # no self loops
    fwd = set(zip(rows.tolist(), indices.tolist()))
    assert all((b, a) in fwd for a, b in list(fwd)[:500])            # symmetric
    ip, ix, da = csr_from_edges([0, 0, 2, 0], [1, 2, 0, 1], 3)
    assert ip.tolist() == [0, 2, 2, 3] and ix.tolist() == [1, 2, 0]


def test_walk_corpus_is_lazy_and_reiterable():
    ids = ["n%d" % i for i in range(6)]
    mat = np.array([[2, 3, 0, 0, 2], [5, 0, 0, 0, 1], [1, 2, 3, 4, 4]], dtype=np.uint32)
    corpus = node2vec.WalkCorpus(mat, ids, chunk=2)
    want = [["n2", "n3"], ["n5"], ["n1", "n2", "n3", "n4"]]
    assert list(corpus) == want and list(corpus) == want and len(corpus) == 3
    assert corpus[2] == want[2]
    g = node2vec.SparseOTF()
    g.set_node_ids(ids)
    assert [g._map_walk(r) for r in mat] == want


def test_read_npz_memory_maps_uncompressed_members(tmp_path):
    """.csr.npz ingestion (reference graph.py:447-486): members stored uncompressed in the CSR dtypes are memory-mapped
    (no host copy on the way to the GPU); anything else falls back to np.load with a dtype conversion."""
    from pecanpy_amd import graph
    from pecanpy_amd.synth import rmat_csr

    ip, ix, dt = rmat_csr(9, seed=2, weighted=True)
    ids = np.arange(ip.size - 1).astype(str)
    p = str(tmp_path / "g.csr.npz")
    np.savez(p, IDs=ids, data=dt, indptr=ip, indices=ix)
    g = graph.SparseGraph()
    g.read_npz(p, True)
    assert isinstance(g.indices, np.memmap) and isinstance(g.indptr, np.memmap) and isinstance(g.data, np.memmap)
    assert np.array_equal(g.indptr, ip) and np.array_equal(g.indices, ix) and np.array_equal(g.data, dt)
    assert list(g.nodes) == list(ids)
    u = graph.SparseGraph()
    u.read_npz(p, False)                       # unweighted: all weights count as one
    assert np.array_equal(u.data, np.ones(ix.size, dtype=np.float32))
    np.savez_compressed(p, IDs=ids, data=dt.astype(np.float64), indptr=ip.astype(np.int64), indices=ix)
    c = graph.SparseGraph()
    c.read_npz(p, True)
    assert c.indptr.dtype == np.uint32 and c.data.dtype == np.float32
    assert np.array_equal(c.indptr, ip) and np.array_equal(c.indices, ix) and np.array_equal(c.data, dt)


# ---- synthetic graph families of the exactness evidence (pecanpy_amd/synth.py) --------------------------------------
def _check_undirected_csr(indptr, indices, data):
    n = indptr.size - 1
    ip = indptr.astype(np.int64)
    assert ip[0] == 0 and ip[-1] == indices.size and (np.diff(ip) >= 0).all() and (data == 1.0).all()
    rows = np.repeat(np.arange(n), np.diff(ip))
    assert (rows != indices).all()                                    # no self loops
    inner = np.ones(indices.size, dtype=bool)
    inner[ip[:-1][np.diff(ip) > 0]] = False
    assert (np.diff(indices.astype(np.int64))[inner[1:]] > 0).all()   # rows strictly ascending
    key = rows * n + indices.astype(np.int64)
    assert np.array_equal(np.sort(key), np.sort(indices.astype(np.int64) * n + rows))   # symmetric
    return np.diff(ip)


def test_ring_lattice_is_regular_and_triangle_rich():
    from pecanpy_amd.synth import ring_lattice_csr

    indptr, indices, data = ring_lattice_csr(400, 12)
    deg = _check_undirected_csr(indptr, indices, data)
    assert (deg == 24).all()
    # neighbours at ring distance t share 2k - t - 1 neighbours
    row0, row5 = set(indices[indptr[0]:indptr[1]]), set(indices[indptr[5]:indptr[6]])
    assert len(row0 & row5) == 2 * 12 - 5 - 1


def test_holme_kim_bipartite_hubs_and_gnm_are_well_formed():
    from pecanpy_amd.synth import bipartite_hubs_csr, gnm_csr, holme_kim_csr

    deg = _check_undirected_csr(*holme_kim_csr(3000, 6, 0.8, seed=1))
    assert deg.min() >= 6 and deg.max() > 60                          # grown by preferential attachment: a heavy tail
    indptr, indices, data = bipartite_hubs_csr(5, 2000, 700, seed=3)
    deg = _check_undirected_csr(indptr, indices, data)
    assert (deg[:5] == 700).all() and (indices[indptr[0]:indptr[5]] >= 5).all()   # hubs only see leaves: no triangles
    deg = _check_undirected_csr(*gnm_csr(5000, 20000, seed=2))
    assert abs(deg.mean() - 8.0) < 0.2


def test_tapered_bounds_cover_the_shard_in_decreasing_chunks():
    from pecanpy_amd.engine import tapered_bounds

    for n in (0, 1, 7, 1000, 5242880, 41943040 // 8):
        for k in (1, 2, 4, 8):
            b = tapered_bounds(n, k)
            assert len(b) == k and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(k - 1))
            sizes = [hi - lo for lo, hi in b]
            assert all(s >= 0 for s in sizes)
            if n >= 1000:
                assert all(sizes[i] >= sizes[i + 1] for i in range(k - 1))
                assert sizes[-1] <= n * 2 // (k * (k + 1)) + 1          # the tail is 2 / (k (k + 1)) of the shard


def test_device_list_specs_parse_without_a_gpu():
    """PECANPY_AMD_DEVICES forms (in-process multi-GPU, round 6): a list may name a device twice (every entry is a replica);
    a bit mask goes through the C ABI (pw_device_mask_to_list), which rejects devices that are not visible."""
    from pecanpy_amd import _lib
    from pecanpy_amd.engine import visible_devices

    assert visible_devices("0,0") == [0, 0]
    assert visible_devices("2, 0 ,1") == [2, 0, 1]
    assert visible_devices([1, 1, 0]) == [1, 1, 0]
    n = _lib.load().pw_device_count()
    assert visible_devices(None) == list(range(n)) and visible_devices("all") == list(range(n))
    assert visible_devices("mask:0x0") == []
    with pytest.raises(_lib.PwError, match="visible"):
        visible_devices("mask:0x8000000000000000")
