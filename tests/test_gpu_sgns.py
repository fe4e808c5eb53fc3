"""Skip-gram training on the GPU (Base.embed / cli.learn_embeddings, src/pecanpy/pecanpy.py:276-290,
cli.py:307-325) against the sequential CPU restatement of the same algorithm (oracle/sgns_ref.c: word2vec.c / gensim
sg=1 negative sampling).

* workers=1 -- ONE wavefront in sentence order: the same updates in the same order as the oracle, so the vectors must
  agree to float tolerance (the oracle sums its dot products in the trainer's lane order; what is left is the GPU's
  fast exp in the sigmoid);
* hogwild (default) -- same set of updates, racing: compared with the oracle through what an embedding is used for,
  the similarity structure (correlation of the cosine matrices, nearest neighbours)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from pecanpy_amd.embed import train_sgns
from pecanpy_amd.engine import PwError

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


MR_HI = {"1", "2", "3", "4", "5", "6", "7", "8", "11", "12", "13", "14", "17", "18", "20", "22"}   # Zachary's first faction


def karate_walks(num_walks=20, L=40, seed=1, p=1.0, q=0.5):
    k = np.load(os.path.join(GOLDEN, "karate_csr.npz"))
    starts = orc.shuffled_starts(34, num_walks, seed)
    return orc.walks_sparse_otf(k["indptr"], k["indices"], k["data"], p, q, starts, L, seed), 34


def faction_gap(vec):
    names = [str(x) for x in np.load(os.path.join(GOLDEN, "karate_csr.npz"))["ids"]]
    unit = vec / np.linalg.norm(vec, axis=1, keepdims=True)
    sim = unit @ unit.T
    same = np.array([[(a in MR_HI) == (b in MR_HI) for b in names] for a in names])
    off = ~np.eye(34, dtype=bool)
    return sim[same & off].mean() - sim[~same].mean()


def test_oracle_trains_a_usable_embedding():
    """(CPU) the restatement itself: the loss falls and Zachary's two factions separate."""
    walks, n = karate_walks()
    _, loss1 = orc.sgns_train(walks, n, dim=16, window=5, epochs=1, seed=3)
    vec, loss30 = orc.sgns_train(walks, n, dim=16, window=5, epochs=30, seed=3)
    assert loss30 < loss1 and np.isfinite(vec).all() and np.abs(vec).max() > 0.1   # trained, not the initial noise (|x| < 0.032)
    again, _ = orc.sgns_train(walks, n, dim=16, window=5, epochs=30, seed=3)
    assert np.array_equal(vec, again)                                       # deterministic
    assert faction_gap(vec) > 0.15


@pytest.mark.gpu
@pytest.mark.parametrize("dim,window,epochs,sample", [(16, 5, 3, 1e-3), (100, 10, 2, 1e-3), (8, 3, 4, 0.0), (128, 4, 1, 0.05)])
def test_single_wavefront_run_equals_the_sequential_restatement(dim, window, epochs, sample):
    walks, n = karate_walks(num_walks=6, L=30, seed=2)
    want, _ = orc.sgns_train(walks, n, dim=dim, window=window, epochs=epochs, sample=sample, seed=7)
    got = train_sgns(walks, n, dim=dim, window=window, epochs=epochs, sample=sample, seed=7, workers=1)
    assert got.shape == want.shape and np.isfinite(got).all()
    # tolerance: 1e-5 relative to the vectors' scale + float32 noise (the sigmoid's exp is the only differing operation)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-5 * scale + 1e-6, np.abs(got - want).max() / scale
    assert np.array_equal(got, train_sgns(walks, n, dim=dim, window=window, epochs=epochs, sample=sample, seed=7, workers=1))


@pytest.mark.gpu
def test_single_wavefront_run_on_rmat_walks():
    """A larger vocabulary with isolated vertices (never in a walk: no slot of the noise table) and dead-end rows."""
    from pecanpy_amd.synth import rmat_csr

    indptr, indices, data = rmat_csr(9, seed=4)
    n = indptr.size - 1
    starts = orc.shuffled_starts(n, 2, 5)
    walks = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 20, 5)
    want, _ = orc.sgns_train(walks, n, dim=32, window=5, epochs=2, seed=11)
    got = train_sgns(walks, n, dim=32, window=5, epochs=2, seed=11, workers=1)
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-6


@pytest.mark.gpu
def test_hogwild_run_agrees_with_the_restatement_in_similarity_structure():
    walks, n = karate_walks(num_walks=40, L=40, seed=3)
    want, _ = orc.sgns_train(walks, n, dim=16, window=5, epochs=30, seed=5)
    got = train_sgns(walks, n, dim=16, window=5, epochs=30, seed=5)

    def cos(v):
        u = v / np.linalg.norm(v, axis=1, keepdims=True)
        return u @ u.T

    a, b = cos(want), cos(got)
    off = ~np.eye(n, dtype=bool)
    assert np.corrcoef(a[off], b[off])[0, 1] > 0.85
    # nearest neighbours: the 5 most similar nodes of every node overlap by more than half on average
    na = np.argsort(-np.where(off, a, -2), axis=1)[:, :5]
    nb = np.argsort(-np.where(off, b, -2), axis=1)[:, :5]
    overlap = np.mean([len(set(x) & set(y)) / 5 for x, y in zip(na, nb)])
    assert overlap > 0.5, overlap


@pytest.mark.gpu
def test_malformed_walk_matrices_are_rejected():
    walks, n = karate_walks(num_walks=2, L=10, seed=1)
    bad = walks.copy()
    bad[3, 2] = 99                                   # node id outside the vocabulary
    with pytest.raises(PwError, match="node id"):
        train_sgns(bad, n, dim=8, window=3, epochs=1, seed=1)
    bad = walks.copy()
    bad[5, -1] = 50                                  # length cell beyond the row
    with pytest.raises(PwError, match="length cell"):
        train_sgns(bad, n, dim=8, window=3, epochs=1, seed=1)
