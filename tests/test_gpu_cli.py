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
def test_cli_main_reproduces_the_reference_walks(mode, tmp_path):
    pytest.importorskip("pecanpy_amd")
    try:
        import gensim  # noqa: F401
        pytest.skip("gensim present: the CLI trains embeddings instead of writing the walks")
    except ImportError:
        pass
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


@pytest.mark.parametrize("mode", ("FirstOrderUnweighted", "PreCompFirstOrder"))
@pytest.mark.parametrize("p,q", [(2, 1), (1, 0.1), (0.1, 0.1)])
def test_cli_first_order_modes_reject_second_order_parameters(mode, p, q, tmp_path):
    edg = tmp_path / "karate.edg"
    _write_edg(edg)
    with pytest.raises(ValueError):
        cli.main(["--input", str(edg), "--output", os.devnull, "--mode", mode, "--p", str(p), "--q", str(q)])


def test_cli_from_npz(tmp_path):
    """--input *.npz (CSR with IDs, graph.py:447-486) through SparseOTF with p = 0.5, q = 2."""
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
