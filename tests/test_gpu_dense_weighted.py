"""walk_dense_weighted_kernel (csrc/walk_dense_w.hip.h: the float64-bounded decision over ONE stream of cur's compressed row)
against the complete kernel (walk_kernel<double, true, ...>: the reference's float64 chain rounding by rounding) and the oracle
(oracle/pecan_oracle.c: orc_walks_dense_otf, pinned to the dense goldens generated from rw/dense_rw.py:34-118)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from pecanpy_amd.engine import WalkEngine

pytestmark = pytest.mark.gpu


def _weighted_er(n, density, seed, symmetric=True):
    rng = np.random.default_rng(seed)
    up = np.triu(rng.random((n, n)) < density, 1)
    w = rng.random((n, n)) * 0.999 + 0.001
    if symmetric:
        mat = np.where(up, w, 0.0)
        mat = mat + mat.T
    else:
        mat = np.where(rng.random((n, n)) < density, w, 0.0)
        np.fill_diagonal(mat, 0.0)
    return mat


def _thresholds(mat, gamma):
    thr = np.zeros(mat.shape[0], dtype=np.float32)
    for i in range(mat.shape[0]):
        row = mat[i, mat[i] != 0]
        thr[i] = row.mean() + gamma * row.std() if row.size else 0.0
    return np.maximum(thr, 0)


def _run(eng, p, q, extend, starts, L, seed, env=None):
    env = env or {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        out = eng.simulate("DenseOTF", p, q, extend, starts, L, seed=seed)
        return out, dict(eng.last_stats)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("extend,gamma,p,q", [(False, 0.0, 0.5, 2.0), (False, 0.0, 0.3, 1.7), (True, 0.0, 0.5, 2.0),
                                              (True, 0.5, 1.5, 0.3), (True, 0.0, 0.25, 4.0)])
def test_bounded_kernel_equals_the_complete_kernel_and_the_oracle(extend, gamma, p, q):
    n = 3000
    mat = _weighted_er(n, 0.25, seed=11)
    mat[17, :] = 0.0   # one isolated vertex
    mat[:, 17] = 0.0
    thr = _thresholds(mat, gamma) if extend else None
    eng = WalkEngine.from_dense(mat)
    if extend:
        eng.set_thresholds(thr)
    starts = orc.shuffled_starts(n, 3, 5)
    L = 30
    fast, sf = _run(eng, p, q, extend, starts, L, 5)
    full, sc = _run(eng, p, q, extend, starts, L, 5, {"PECANPY_AMD_DENSE_NO_WFAST": "1"})
    assert np.array_equal(fast, full)
    assert sf["total_steps"] == sc["total_steps"] and sf["redo_walks"] == 0
    m = 300
    want, ost = orc.walks_dense_otf(mat, p, q, starts[:m], L, 5, thr=thr, return_stats=True)
    assert np.array_equal(fast[:m], want)
    # the in-kernel float64 chain (what decides a step whose partial sum falls inside the bound's interval: one in ~10^8) on
    # every fifth step -- same matrix, and the counter says so
    ex, sx = _run(eng, p, q, extend, starts, L, 5, {"PECANPY_AMD_DENSE_EXACT_TEST": "5"})
    assert np.array_equal(ex, fast) and sx["redo_walks"] == 0
    assert sx["ambiguous_steps"] >= sf["total_steps"] // 6 and sf["ambiguous_steps"] <= 2
    # the hand-over: every 7th walk is given to the complete kernel at its third step -- same matrix
    redo, sr = _run(eng, p, q, extend, starts, L, 5, {"PECANPY_AMD_DENSE_REDO_TEST": "7"})
    assert np.array_equal(redo, fast) and sr["redo_walks"] >= starts.size // 7 - 1 and sr["total_steps"] == sf["total_steps"]


@pytest.mark.parametrize("extend", [False, True])
def test_bounded_kernel_on_a_directed_weighted_matrix_with_dead_ends(extend):
    """Asymmetric matrix (the reference does not require symmetry: dense_rw.py:87-88 only ASSUMES it for node2vec+), rows
    without out-neighbours (walks end early, the stream addresses of the later walks move), rows wider than one block."""
    n = 1500
    mat = _weighted_er(n, 0.3, seed=4, symmetric=False)
    mat[::13, :] = 0.0   # sinks
    thr = _thresholds(mat, 0.0) if extend else None
    eng = WalkEngine.from_dense(mat)
    if extend:
        eng.set_thresholds(thr)
    starts = orc.shuffled_starts(n, 2, 9)
    want, ost = orc.walks_dense_otf(mat, 0.5, 2.0, starts, 20, 9, thr=thr, return_stats=True)
    got, st = _run(eng, 0.5, 2.0, extend, starts, 20, 9)
    assert np.array_equal(got, want)
    assert st["total_steps"] == ost.total_steps


def test_negative_weights_take_the_complete_kernel():
    """The bound needs non-negative values: a matrix with a negative entry is walked by the complete kernel alone (same walks
    as with the bounded kernel switched off)."""
    n = 400
    mat = _weighted_er(n, 0.3, seed=2)
    mat[3, 5] = mat[5, 3] = -0.25
    eng = WalkEngine.from_dense(mat)
    starts = orc.shuffled_starts(n, 1, 1)
    a, sa = _run(eng, 0.5, 2.0, False, starts, 10, 1)
    b, sb = _run(eng, 0.5, 2.0, False, starts, 10, 1, {"PECANPY_AMD_DENSE_NO_WFAST": "1"})
    assert np.array_equal(a, b)


def test_bounded_kernel_wide_rows_prefix_against_the_oracle():
    """N = 20 000, density 0.25 (rows of ~5 000 non-zeros = 20 blocks, 313 words of packed row): node2vec and node2vec+
    against the oracle on the first walks, bounded == complete on a larger sample."""
    n = 20000
    rng = np.random.default_rng(1)
    up = np.triu(rng.random((n, n), dtype=np.float32) < 0.25, 1)
    w = rng.random((n, n), dtype=np.float32).astype(np.float64) * 0.999 + 0.001
    mat = np.where(up, w, 0.0)
    del up, w
    mat = mat + mat.T
    thr = _thresholds(mat, 0.0)
    eng = WalkEngine.from_dense(mat)
    eng.set_thresholds(thr)
    starts = orc.shuffled_starts(n, 1, 3)[:4000]
    for extend, p, q in ((False, 0.5, 2.0), (True, 0.5, 2.0), (True, 0.3, 1.7)):
        fast, sf = _run(eng, p, q, extend, starts, 40, 3)
        full, sc = _run(eng, p, q, extend, starts, 40, 3, {"PECANPY_AMD_DENSE_NO_WFAST": "1"})
        assert np.array_equal(fast, full), (extend, p, q)
        assert sf["redo_walks"] == 0
        want = orc.walks_dense_otf(mat, p, q, starts[:60], 40, 3, thr=thr if extend else None)
        assert np.array_equal(fast[:60], want), (extend, p, q)
