"""CLI end to end on the GPU (BASELINE config C1's plumbing; reference test/test_cli.py:59-87): read graph ->
preprocess -> simulate walks -> learn embeddings.  gensim is not part of the image, so the last stage writes the walks
(one per line, node IDs) -- which makes the whole chain checkable against the reference-generated goldens."""
import os
import warnings

import numpy as np
import pytest

from pecanpy_amd import cli

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = ("SparseOTF", "DenseOTF", "PreComp", "PreCompFirstOrder", "FirstOrderUnweighted")


def _write_edg(path):
    """demo/karate.edg rebuilt from the golden CSR so that the reader's first-appearance numbering
    (graph.py:270-341) reproduces the fixture's vertex indices."""
    k = np.load(os.path.join(GOLDEN, "karate_csr.npz"))
    indptr, indices, ids = k["indptr"], k["indices"], k["ids"]
    n = indptr.size - 1
    rows = [set(indices[indptr[i]:indptr[i + 1]].tolist()) for i in range(n)]
    seen, lines, used = set(), [], set()

    def emit(a, b):
        lines.append(f"{ids[a]}\t{ids[b]}\n")
        used.add((min(a, b), max(a, b)))
        seen.update((a, b))

    for v in range(n):
        if v in seen:
            continue
        earlier = sorted(u for u in rows[v] if u in seen)
        if earlier:
            emit(earlier[0], v)
        else:
            assert v + 1 in rows[v], "fixture numbering cannot be reproduced by an edge list"
            emit(v, v + 1)
    for u in range(n):
        for v in sorted(rows[u]):
            if u < v and (u, v) not in used:
                emit(u, v)
    with open(path, "w") as f:
        f.writelines(lines)
    return ids


@pytest.mark.parametrize("mode", MODES)
def test_cli_main_reproduces_the_reference_walks(mode, tmp_path, monkeypatch):
    monkeypatch.setenv("PECANPY_AMD_DUMP_WALKS", "1")      # last stage: write the walks instead of training
    edg, out = tmp_path / "karate.edg", tmp_path / "karate.walks"
    ids = _write_edg(edg)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cli.main(["--input", str(edg), "--output", str(out), "--mode", mode, "--random_state", "0",
                  "--num-walks", "10", "--walk-length", "80", "--workers", "1"])
    gold = np.load(os.path.join(GOLDEN, f"karate_{mode}_p1_q1.npz"))
    want = [" ".join(ids[row[: row[-1]]].tolist()) for row in gold["walks"]]
    got = out.read_text().splitlines()
    assert got == want


def test_cli_uses_the_named_devices_without_a_launcher(tmp_path, monkeypatch):
    """Round 6: the console script spreads the walks over the GPUs of THIS process (PECANPY_AMD_DEVICES; no torchrun) -- here the
    one device of the box named twice -- and writes the same walks as the one-device run (the reference's, SparseOTF p = q = 1)."""
    monkeypatch.setenv("PECANPY_AMD_DUMP_WALKS", "1")
    monkeypatch.setenv("PECANPY_AMD_DEVICES", "0,0")
    edg, out = tmp_path / "karate.edg", tmp_path / "karate.walks"
    ids = _write_edg(edg)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cli.main(["--input", str(edg), "--output", str(out), "--mode", "SparseOTF", "--random_state", "0",
                  "--num-walks", "10", "--walk-length", "80", "--workers", "1"])
    gold = np.load(os.path.join(GOLDEN, "karate_SparseOTF_p1_q1.npz"))
    want = [" ".join(ids[row[: row[-1]]].tolist()) for row in gold["walks"]]
    assert out.read_text().splitlines() == want


@pytest.mark.parametrize("mode", ("FirstOrderUnweighted", "PreCompFirstOrder"))
@pytest.mark.parametrize("p,q", [(2, 1), (1, 0.1), (0.1, 0.1)])
def test_cli_first_order_modes_reject_second_order_parameters(mode, p, q, tmp_path):
    edg = tmp_path / "karate.edg"
    _write_edg(edg)
    with pytest.raises(ValueError):
        cli.main(["--input", str(edg), "--output", os.devnull, "--mode", mode, "--p", str(p), "--q", str(q)])


def test_cli_from_npz(tmp_path, monkeypatch):
    """--input *.npz (CSR with IDs, graph.py:447-486) through SparseOTF with p = 0.5, q = 2."""
    monkeypatch.setenv("PECANPY_AMD_DUMP_WALKS", "1")
    out, npz = tmp_path / "w.txt", tmp_path / "karate.csr.npz"
    k = np.load(os.path.join(GOLDEN, "karate_csr.npz"))
    np.savez(npz, IDs=k["ids"], data=k["data"], indptr=k["indptr"], indices=k["indices"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cli.main(["--input", str(npz), "--output", str(out), "--mode", "SparseOTF",
                  "--p", "0.5", "--q", "2", "--random_state", "0", "--num-walks", "10", "--walk-length", "80"])
    gold = np.load(os.path.join(GOLDEN, "karate_SparseOTF_p0.5_q2.npz"))
    want = [" ".join(k["ids"][row[: row[-1]]].tolist()) for row in gold["walks"]]
    assert out.read_text().splitlines() == want


# Zachary's factions (Mr. Hi's group, 1-based IDs as in demo/karate.edg)
MR_HI = {"1", "2", "3", "4", "5", "6", "7", "8", "11", "12", "13", "14", "17", "18", "20", "22"}


def test_cli_end_to_end_embeddings_on_the_gpu(tmp_path):
    """The whole pipeline without gensim: edge list -> walks (GPU) -> skip-gram (GPU) -> word2vec text file; the
    embedding must carry the graph's structure: members of the same faction are closer than members of different ones."""
    try:
        import gensim  # noqa: F401
        pytest.skip("gensim present: the reference's trainer is used")
    except ImportError:
        pass
    edg, out = tmp_path / "karate.edg", tmp_path / "karate.emb"
    _write_edg(edg)
    cli.main(["--input", str(edg), "--output", str(out), "--mode", "SparseOTF", "--p", "1", "--q", "0.5", "--random_state", "1",
              "--num-walks", "20", "--walk-length", "40", "--dimensions", "16", "--epochs", "40", "--window-size", "5"])
    lines = out.read_text().splitlines()
    n, dim = (int(x) for x in lines[0].split())
    assert (n, dim) == (34, 16) and len(lines) == 35
    names = [ln.split()[0] for ln in lines[1:]]
    vec = np.array([[float(x) for x in ln.split()[1:]] for ln in lines[1:]])
    assert np.isfinite(vec).all() and np.abs(vec).max() > 0.1             # trained, not the initial noise (|x| < 0.032)
    unit = vec / np.linalg.norm(vec, axis=1, keepdims=True)
    sim = unit @ unit.T
    same = np.array([[(a in MR_HI) == (b in MR_HI) for b in names] for a in names])
    off = ~np.eye(34, dtype=bool)
    assert sim[same & off].mean() > sim[~same].mean() + 0.15


def test_embed_method_returns_node_ordered_vectors():
    from pecanpy_amd import pecanpy as node2vec

    k = np.load(os.path.join(GOLDEN, "karate_csr.npz"))
    g = node2vec.SparseOTF.from_csr(k["indptr"], k["indices"], k["data"], node_ids=list(k["ids"]), p=1, q=1, random_state=2)
    try:
        import gensim  # noqa: F401
        pytest.skip("gensim present")
    except ImportError:
        pass
    emb = g.embed(dim=8, num_walks=10, walk_length=20, window_size=4, epochs=3)
    assert emb.shape == (34, 8) and emb.dtype == np.float32 and np.isfinite(emb).all()
    again = g.embed(dim=8, num_walks=10, walk_length=20, window_size=4, epochs=3)
    assert again.shape == emb.shape          # (hogwild updates: repeatable in distribution, not bit for bit)
