#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the *reference itself*.

Runs ONLY in the build container (where /root/reference is mounted); the fixtures it writes are
plain arrays (inputs + expected outputs) and travel with the repo.  Recipe = SURVEY.md 8(c):
the reference's ``@njit`` bodies are valid plain Python, so four trivial stub packages
(tests/golden/_stubs) let ``from pecanpy import pecanpy`` run on NumPy in the interpreter.

Where NumPy-in-the-interpreter differs from Numba the harness patches the
*data objects* handed to the reference, never the reference code:
  * ``SeqArray``: ndarray subclass whose ``.sum()`` is a sequential same-dtype loop (Numba
    ``arr.sum()``) and whose in-place ``/=`` / ``*=`` compute in float64 then cast back
    (Numba picks the ``dd->d`` loop for float32-array op float64-scalar).
  * ``isnotin_extended`` is wrapped so ``t`` is handed on as float64 (so that
    ``alpha = 1/q + (1-1/q)*t`` is float64 as under Numba).
  * PreComp: ``alias_indptr``/``alias_dim`` cast to int64 (NumPy promotes uint64+int64 to float64).

usage:  python tests/golden/make_golden.py        (rewrites tests/golden/*.npz)
"""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.insert(1, os.path.join(REF, "src"))

import numpy as np  # noqa: E402
from numba_progress import ProgressBar  # noqa: E402  (stub)
from pecanpy import pecanpy as ref  # noqa: E402  (the reference)
from pecanpy.rw import sparse_rw as ref_sparse_rw  # noqa: E402

_spec = importlib.util.spec_from_file_location("synth", os.path.join(REPO, "pecanpy_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)


# --------------------------------------------------------------------------------------------
# Numba-semantics shims on data objects
# --------------------------------------------------------------------------------------------
class SeqArray(np.ndarray):
    def sum(self, *args, **kwargs):
        acc = self.dtype.type(0)
        for v in np.asarray(self).ravel():
            acc = self.dtype.type(acc + v)
        return acc

    def __itruediv__(self, other):
        res = np.asarray(self, dtype=np.float64) / np.asarray(other, dtype=np.float64)
        np.asarray(self)[...] = res.astype(self.dtype)
        return self

    def __imul__(self, other):
        res = np.asarray(self, dtype=np.float64) * np.asarray(other, dtype=np.float64)
        np.asarray(self)[...] = res.astype(self.dtype)
        return self


_orig_isnotin_extended = ref_sparse_rw.isnotin_extended


def _isnotin_extended_f64(*args):
    ind, t = _orig_isnotin_extended(*args)
    return ind, t.astype(np.float64)


ref_sparse_rw.isnotin_extended = _isnotin_extended_f64


def _shim(g):
    if hasattr(g, "indptr"):
        g.data = g.data.view(SeqArray)
    else:
        g._data = g._data.view(SeqArray)
    return g


def ref_walk_matrix(g, num_walks, walk_length):
    """Same statements as Base.simulate_walks up to the index matrix (pecanpy.py:133-157)."""
    g._preprocess_transition_probs()
    if isinstance(g, ref.PreComp):
        g.alias_indptr = g.alias_indptr.astype(np.int64)
        g.alias_dim = g.alias_dim.astype(np.int64)
    nodes = np.array(range(g.num_nodes), dtype=np.uint32)
    starts = np.concatenate([nodes] * num_walks)
    np.random.seed(g.random_state)
    np.random.shuffle(starts)
    move_forward = g.get_move_forward()
    has_nbrs = g.get_has_nbrs()
    with ProgressBar(total=starts.size, disable=True) as progress:
        mat = g._random_walks(starts.size, walk_length, g.random_state, starts, has_nbrs,
                              move_forward, progress)
    return starts, mat


def dense_from_csr(indptr, indices, data):
    n = indptr.size - 1
    mat = np.zeros((n, n), dtype=np.float64)
    for i in range(n):
        sl = slice(indptr[i], indptr[i + 1])
        mat[i, indices[sl]] = data[sl]
    return mat


def make_graph(mode, indptr, indices, data, **kw):
    cls = getattr(ref, mode)
    g = cls(**kw)
    ids = [str(i) for i in range(indptr.size - 1)]
    g.set_node_ids(ids)
    if mode == "DenseOTF":
        g.data = dense_from_csr(indptr, indices, data)
    else:
        g.indptr = indptr.astype(np.uint32).copy()
        g.indices = indices.astype(np.uint32).copy()
        g.data = data.astype(np.float32).copy()
    return _shim(g)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}")


def walk_case(name, mode, indptr, indices, data, p, q, extend, gamma, seed, num_walks,
              walk_length, n_prob_samples=0):
    g = make_graph(mode, indptr, indices, data, p=p, q=q, extend=extend, gamma=gamma,
                   random_state=seed)
    starts, mat = ref_walk_matrix(g, num_walks, walk_length)
    thr = g.get_noise_thresholds() if extend else np.zeros(0, np.float32)
    out = dict(mode=mode, indptr=indptr, indices=indices, data=data, p=float(p), q=float(q),
               extend=bool(extend), gamma=float(gamma), seed=int(seed), num_walks=int(num_walks),
               walk_length=int(walk_length), starts=starts, walks=mat, thr=thr)
    if n_prob_samples and mode != "DenseOTF":
        # per-step probability vectors for bitwise comparison: (cur, prev) pairs visited by walks
        rng = np.random.default_rng(123)
        pairs = []
        for row in mat[rng.permutation(mat.shape[0])[: n_prob_samples]]:
            ln = int(row[-1])
            if ln >= 3:
                j = int(rng.integers(2, ln))
                pairs.append((int(row[j - 1]), int(row[j - 2])))
        pc, pp, pv, po = [], [], [], [0]
        fn, thr_arg = g.setup_get_normalized_probs()
        for cur, prev in pairs:
            if g.indptr[cur] == g.indptr[cur + 1]:
                continue
            pr = fn(g.data, g.indices, g.indptr, g.p, g.q, cur, prev, thr_arg)
            pc.append(cur)
            pp.append(prev)
            pv.append(np.asarray(pr, dtype=np.float32))
            po.append(po[-1] + pr.size)
        out.update(prob_cur=np.array(pc, np.uint32), prob_prev=np.array(pp, np.uint32),
                   prob_vals=np.concatenate(pv) if pv else np.zeros(0, np.float32),
                   prob_off=np.array(po, np.int64))
    save(name, **out)


def main():
    # (i) the reference's own known-answer test (test/test_walk.py:10-18, 85-97): regenerate and
    #     compare with the committed table in tests/golden/ref_test_walk.py (hand-entered data)
    from ref_test_walk import IDS, MAT, WALKS

    for label, cls in [("FirstOrderUnweighted", ref.FirstOrderUnweighted), ("PreComp", ref.PreComp),
                       ("SparseOTF", ref.SparseOTF), ("DenseOTF", ref.DenseOTF)]:
        g = cls.from_mat(MAT, IDS, p=1, q=1, random_state=0)
        _, mat = ref_walk_matrix(g, 2, 3)
        assert [g._map_walk(row) for row in mat] == WALKS[label], label
    print("stub-import oracle reproduces test/test_walk.py goldens: OK")

    # (ii) karate club (BASELINE config C1 twin): CSR as produced by the reference's own reader
    k = ref.SparseOTF()
    k.read_edg(os.path.join(REF, "demo", "karate.edg"), weighted=False, directed=False)
    kip, kix, kda = k.indptr.copy(), k.indices.copy(), k.data.copy()
    ids = np.array(k.nodes)
    save("karate_csr", indptr=kip, indices=kix, data=kda, ids=ids)
    for mode in ["SparseOTF", "DenseOTF", "PreComp", "FirstOrderUnweighted", "PreCompFirstOrder"]:
        walk_case(f"karate_{mode}_p1_q1", mode, kip, kix, kda, 1, 1, False, 0, 0, 10, 80)
    for mode in ["SparseOTF", "DenseOTF", "PreComp"]:
        walk_case(f"karate_{mode}_p0.5_q2", mode, kip, kix, kda, 0.5, 2, False, 0, 0, 10, 80,
                  n_prob_samples=40)
        walk_case(f"karate_{mode}_p0.25_q4", mode, kip, kix, kda, 0.25, 4, False, 0, 7, 10, 80)
    # non power-of-two p, q: exercises fl32(f64(w)/q) (needs the SeqArray shim)
    walk_case("karate_SparseOTF_p0.3_q1.7", "SparseOTF", kip, kix, kda, 0.3, 1.7, False, 0, 3, 10, 40,
              n_prob_samples=40)
    walk_case("karate_DenseOTF_p0.3_q1.7", "DenseOTF", kip, kix, kda, 0.3, 1.7, False, 0, 3, 10, 40)

    # (iii) RMAT-10 with isolated vertices
    rip, rix, rda = synth.rmat_csr(10, seed=1)
    walk_case("rmat10_SparseOTF_p0.5_q2", "SparseOTF", rip, rix, rda, 0.5, 2, False, 0, 0, 2, 20,
              n_prob_samples=60)

    # (iv) directed graph with a sink and an isolated vertex: dead-end bookkeeping (App. A.6)
    dmat = np.array([
        [0, 1, 0, 0, 0, 0],
        [1, 0, 0, 1, 0, 0],
        [0, 0, 0, 0, 0, 0],   # sink / no out edges
        [0, 1, 1, 0, 1, 0],
        [0, 0, 0, 1, 0, 0],
        [0, 0, 0, 0, 0, 0],   # isolated
    ], dtype=float)
    dg = ref.SparseOTF.from_mat(dmat, [str(i) for i in range(6)])
    for mode in ["SparseOTF", "DenseOTF"]:
        walk_case(f"sink_{mode}_p0.5_q2", mode, dg.indptr, dg.indices, dg.data, 0.5, 2, False, 0, 5, 6, 12)

    # (v) weighted graphs: n2v and n2v+ (gamma 0 and 0.5), sparse and dense
    rng = np.random.default_rng(42)
    n = 48
    upper = np.triu(rng.random((n, n)) < 0.3, 1)
    wts_dyadic = np.triu(rng.integers(1, 9, size=(n, n)) / 4.0, 1) * upper
    wts_real = np.triu(rng.random((n, n)).astype(np.float32).astype(np.float64) + 0.05, 1) * upper
    for tag, wm in [("wdy", wts_dyadic), ("wre", wts_real)]:
        wm = (wm + wm.T).astype(np.float32).astype(np.float64)
        wg = ref.SparseOTF.from_mat(wm, [str(i) for i in range(n)])
        for mode in ["SparseOTF", "DenseOTF", "PreComp"]:
            for extend, gamma in [(False, 0.0), (True, 0.0), (True, 0.5)]:
                if mode == "PreComp" and gamma == 0.5:
                    continue
                nm = f"{tag}_{mode}_{'n2vplus' if extend else 'n2v'}_g{gamma}_p0.5_q2"
                walk_case(nm, mode, wg.indptr, wg.indices, wg.data, 0.5, 2, extend, gamma, 11, 4, 30,
                          n_prob_samples=30)
        walk_case(f"{tag}_SparseOTF_n2vplus_g0.0_p0.7_q0.4", "SparseOTF", wg.indptr, wg.indices, wg.data,
                  0.7, 0.4, True, 0.0, 13, 4, 30, n_prob_samples=30)

    # (vi) self loops (round 6; the reference accepts them, graph.py:238-268): a walker may step u -> u, and a looped prev
    #      is a member of N(prev) & N(cur) that the reference takes out again (sparse_rw.py:79-87, dense_rw.py:57-60)
    rng = np.random.default_rng(77)
    n = 150
    adj = np.triu(rng.random((n, n)) < 0.08, 1)
    adj = (adj | adj.T).astype(float)
    adj[np.arange(0, n, 4), np.arange(0, n, 4)] = 1.0          # every 4th vertex has a self loop
    hub = 7                                                     # ... and a looped hub adjacent to half of the graph
    adj[hub, ::2] = 1.0
    adj[::2, hub] = 1.0
    adj[hub, hub] = 1.0
    lg = ref.SparseOTF.from_mat(adj, [str(i) for i in range(n)])
    assert (lg.indices[lg.indptr[hub]:lg.indptr[hub + 1]] == hub).any()
    for mode in ["SparseOTF", "DenseOTF"]:
        walk_case(f"selfloop_{mode}_p0.5_q2", mode, lg.indptr, lg.indices, lg.data, 0.5, 2, False, 0, 21, 4, 30,
                  n_prob_samples=40)
    walk_case("selfloop_SparseOTF_p0.3_q1.7", "SparseOTF", lg.indptr, lg.indices, lg.data, 0.3, 1.7, False, 0, 22, 3, 30,
              n_prob_samples=40)

    # (vi-b) a sink-heavy directed graph (the dead-end bookkeeping of App. A.6 pinned at more than six nodes):
    #        400 vertices, 40 % without out-edges, so that most walks end early and shift the stream of every later walk
    rng = np.random.default_rng(78)
    n = 400
    dm = (rng.random((n, n)) < 0.02).astype(float)
    np.fill_diagonal(dm, 0.0)
    dm[rng.random(n) < 0.4, :] = 0.0
    sg = ref.SparseOTF.from_mat(dm, [str(i) for i in range(n)])
    walk_case("sinkheavy_SparseOTF_p0.5_q2", "SparseOTF", sg.indptr, sg.indices, sg.data, 0.5, 2, False, 0, 9, 4, 20)
    walk_case("sinkheavy_SparseOTF_p0.3_q1.7", "SparseOTF", sg.indptr, sg.indices, sg.data, 0.3, 1.7, False, 0, 10, 3, 20)

    # (vii) MT19937 known answers straight from NumPy's legacy generator
    offs = np.array([0, 311, 312, 623, 624, 10**6, 5 * 10**6], dtype=np.int64)
    seeds = np.array([0, 1, 12345, 2**32 - 1], dtype=np.int64)
    vals = np.zeros((seeds.size, offs.size, 4), dtype=np.float64)
    for si, s in enumerate(seeds):
        rs = np.random.RandomState(int(s))
        stream = rs.random_sample(int(offs.max()) + 4)
        for oi, o in enumerate(offs):
            vals[si, oi] = stream[o:o + 4]
    save("mt19937_known", seeds=seeds, offsets=offs, values=vals)


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    main()
