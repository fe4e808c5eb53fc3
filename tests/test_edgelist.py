"""Edge-list ingestion: the native reader of libpecanpy_amd and the Python
fallback against golden vectors produced by the reference's own AdjlstGraph
(tests/golden/make_golden_edgelist.py), plus native == fallback on a larger random file."""
import json
import os
import warnings

import numpy as np
import pytest

from pecanpy_amd import graph

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "edgelist_cases.json")) as _f:
    CASES = json.load(_f)


def _write(tmp_path, case):
    path = tmp_path / (case["name"] + ".edg")
    with open(path, "w", newline="") as f:
        f.write(case["text"])
    return str(path)


def _check(g, caught, case):
    assert list(g.nodes) == case["ids"]
    assert g.num_edges == case["num_edges"]
    indptr, indices, data = g.to_csr()
    assert indptr.dtype == np.uint32 and indices.dtype == np.uint32 and data.dtype == np.float32
    assert indptr.tolist() == case["indptr"] and indices.tolist() == case["indices"]
    assert data.view(np.uint32).tolist() == case["data_bits"]
    assert g.to_dense().view(np.uint64).ravel().tolist() == case["dense_bits"]
    assert [[h, t, float(w).hex()] for h, t, w in g.edges] == case["edges"]
    assert len(caught) == case["n_warnings"]


@pytest.mark.parametrize("native", [True, False], ids=["native", "python"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reader_matches_reference_golden(tmp_path, monkeypatch, case, native):
    if not native:
        monkeypatch.setattr(graph, "_native_edgelist", lambda *a: None)
    path = _write(tmp_path, case)
    g = graph.AdjlstGraph()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        if case["error"]:
            with pytest.raises({"ValueError": ValueError, "IndexError": IndexError}[case["error"]]):
                g.read(path, case["weighted"], case["directed"], case["delimiter"])
            return
        g.read(path, case["weighted"], case["directed"], case["delimiter"])
    _check(g, caught, case)


def test_native_reader_is_used_for_well_formed_files(tmp_path):
    calls = []
    real = graph._native_edgelist

    def spy(*a):
        res = real(*a)
        calls.append(res is not None)
        return res

    graph._native_edgelist = spy
    try:
        for case in CASES:
            if case["error"]:
                continue
            g = graph.AdjlstGraph()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                g.read(_write(tmp_path, case), case["weighted"], case["directed"], case["delimiter"])
    finally:
        graph._native_edgelist = real
    used = dict(zip([c["name"] for c in CASES if not c["error"]], calls))
    assert used["random_weighted"] and used["plain_unweighted"] and used["crlf"] and used["weights_formats"]
    # everything that warns or needs Python's float() goes through the statement-by-statement reader
    assert not used["duplicate_conflict"] and not used["nonpositive_weight"]
    assert not used["nan_weight"] and not used["underscore_weight"]


def test_native_equals_python_reader_on_a_larger_file(tmp_path, monkeypatch):
    rng = np.random.default_rng(11)
    n, m = 5000, 60000
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    w = {}
    path = tmp_path / "big.edg"
    with open(path, "w") as f:
        for s, d in zip(src.tolist(), dst.tolist()):
            x = w.setdefault((min(s, d), max(s, d)), round(float(rng.random()) * 5 + 0.01, 4))
            f.write(f"n{s}\tn{d}\t{x}\n")
    fast = graph.SparseGraph()
    fast.read_edg(str(path), True, False)
    monkeypatch.setattr(graph, "_native_edgelist", lambda *a: None)
    slow = graph.SparseGraph()
    slow.read_edg(str(path), True, False)
    assert fast.nodes == slow.nodes
    for a, b in ((fast.indptr, slow.indptr), (fast.indices, slow.indices), (fast.data, slow.data)):
        assert a.dtype == b.dtype and np.array_equal(a, b)


def test_graph_stays_mutable_after_a_bulk_read(tmp_path):
    case = next(c for c in CASES if c["name"] == "plain_unweighted")
    g = graph.AdjlstGraph()
    g.read(_write(tmp_path, case), False, False)
    before = g.num_edges
    g.add_edge("a", "zz", 2.5)
    assert g.num_edges == before + 2 and g.nodes[-1] == "zz"
    indptr, indices, data = g.to_csr()
    a, zz = g.nodes.index("a"), g.nodes.index("zz")
    row = indices[indptr[a]:indptr[a + 1]].tolist()
    assert zz in row and data[indptr[zz]] == np.float32(2.5)
