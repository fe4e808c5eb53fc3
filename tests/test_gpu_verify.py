"""Device-side, family-diverse evidence for the exactness of the lane path's interval decision (csrc/seqscan.h:
lane_tight -- the routine that settles ~11 % of the RMAT-22 steps from a bounded float32 drift instead of running the
reference's sequential chain, pecanpy.py:556-557).

PECANPY_AMD_VERIFY_TIGHT=1 makes the lane kernel record EVERY step lane_tight settles; lanes_verify_kernel then decides
each of them again with the float32 chain itself (lane_chain, one thread per step, on the device) and counts the steps
where the two disagree (pw_stats.verify_mismatch).  R-MAT rows rarely produce the arithmetic coincidences a rounding
argument can trip over, so the graph families here are chosen for them: power-of-two degrees and totals with
block-structured common neighbours (ring lattice), long rows full of common neighbours (Holme-Kim: power law + high
clustering), long rows without a single common neighbour (bipartite hubs: the closed form), a 40 000-entry hub row, and
sparse ER.  The RMAT-22 runs (1.6e9 transitions each, three (p, q)) are in tests/test_gpu_scale.py, on that module's
graph."""
import numpy as np
import pytest

from pecanpy_amd.engine import WalkEngine
from pecanpy_amd.synth import bipartite_hubs_csr, csr_from_edges, gnm_csr, holme_kim_csr, ring_lattice_csr

pytestmark = pytest.mark.gpu


def verified_run(monkeypatch, indptr, indices, p, q, num_walks, L, seed, eng=None, d_starts=None):
    """One pass with the verification on, one without: same matrix; returns the statistics of the verified pass."""
    import torch

    if eng is None:
        eng = WalkEngine.from_csr(indptr, indices, None)
    if d_starts is None:
        n = indptr.size - 1
        starts = np.concatenate([np.arange(n, dtype=np.uint32)] * num_walks)
        np.random.RandomState(seed).shuffle(starts)
        d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    monkeypatch.setenv("PECANPY_AMD_VERIFY_TIGHT", "1")
    a = eng.simulate_device("SparseOTF", p, q, False, d_starts, L, seed=seed)
    st = dict(eng.last_stats)
    monkeypatch.delenv("PECANPY_AMD_VERIFY_TIGHT")
    b = eng.simulate_device("SparseOTF", p, q, False, d_starts, L, seed=seed)
    sb = eng.last_stats                                                 # (default: the production SAMPLE, ~1/1024 of them)
    assert sb["verify_checked"] <= st["verify_checked"] // 256 + 64 and sb["verify_mismatch"] == 0 and sb["verify_dropped"] == 0, sb
    assert torch.equal(a, b)
    assert st["lane_kernel"] == 1
    assert st["verify_mismatch"] == 0, st
    assert st["verify_dropped"] == 0, st
    return st, eng, d_starts


FAMILIES = {
    # name: (graph builder, walks per vertex, walk length)
    "ring_lattice_512": (lambda: ring_lattice_csr(1 << 14, 256), 48, 80),        # 512-regular, 16384 vertices
    "ring_lattice_64": (lambda: ring_lattice_csr(1 << 17, 32), 16, 80),          # 64-regular: short tie-heavy rows
    "holme_kim": (lambda: holme_kim_csr(1 << 17, 12, 0.8, seed=5), 32, 80),
    "bipartite_hubs": (lambda: bipartite_hubs_csr(48, 150000, 30000, seed=2), 24, 80),
    "gnm_sparse": (lambda: gnm_csr(1 << 19, 1 << 22, seed=4), 10, 80),
}


_cache = {}   # the engine of the family tested last (the index of a family is built once for its three (p, q))


@pytest.mark.parametrize("p,q", [(0.5, 2.0), (4.0, 0.25), (1.0, 1.0)])
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_interval_decision_verified_by_the_float_chain(family, p, q, monkeypatch):
    build, num_walks, L = FAMILIES[family]
    if _cache.get("family") != family:
        _cache.clear()
        indptr, indices, _ = build()
        _cache.update(family=family, indptr=indptr, indices=indices, eng=None, d_starts=None)
    st, _cache["eng"], _cache["d_starts"] = verified_run(monkeypatch, _cache["indptr"], _cache["indices"], p, q, num_walks, L,
                                                          seed=11, eng=_cache["eng"], d_starts=_cache["d_starts"])
    if family in ("ring_lattice_512", "holme_kim", "bipartite_hubs"):
        assert st["verify_checked"] > 1000, st                          # the family does exercise the interval decision
    print(f"[verify] {family} p={p} q={q}: {st['total_steps']} transitions, {st['ambiguous_steps']} ambiguous, "
          f"{st['verify_checked']} settled by the interval decision and re-decided by the chain, "
          f"{st['wave_chain_steps']} float chains, {st['verify_ties']} chains declined (ties)")


def test_interval_decision_verified_on_the_hub_graph(monkeypatch):
    """A 40 000-entry hub row with thousands of common neighbours per edge (the longest lists / bisections)."""
    rng = np.random.default_rng(13)
    n, hub_deg = 60000, 40000
    hub = np.arange(1, hub_deg + 1)
    src = [np.zeros(hub.size, dtype=np.int64), rng.integers(1, n, 300000), np.full(3000, 7, dtype=np.int64)]
    dst = [hub, rng.integers(1, n, 300000), rng.integers(1, n, 3000)]
    s, d = np.concatenate(src), np.concatenate(dst)
    keep = s != d
    s, d = s[keep], d[keep]
    indptr, indices, _ = csr_from_edges(np.concatenate([s, d]), np.concatenate([d, s]), n)
    for p, q in ((0.5, 2.0), (0.25, 4.0), (2.0, 0.5)):
        st, _, _ = verified_run(monkeypatch, indptr, indices, p, q, 40, 80, seed=3)
        assert st["verify_checked"] > 1000, st


def test_verification_is_live(monkeypatch):
    """The check itself: (a) PECANPY_AMD_VERIFY_TIGHT=poison falsifies every 1024th RECORD (the walks are untouched) and
    the chain must flag exactly those; (b) with the record buffer capped the excess is counted as dropped, not ignored."""
    import torch

    indptr, indices, _ = ring_lattice_csr(1 << 12, 256)
    eng = WalkEngine.from_csr(indptr, indices, None)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 32)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    clean = eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 80, seed=1)
    monkeypatch.setenv("PECANPY_AMD_VERIFY_TIGHT", "poison")
    out = eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 80, seed=1)
    st = dict(eng.last_stats)
    assert torch.equal(out, clean)
    assert st["verify_checked"] > 4096
    lo = st["verify_checked"] // 1024                                   # one per 1024 records of each round (+ slot 0 of each round)
    assert lo - st["verify_ties"] <= st["verify_mismatch"] <= lo + st["lane_rounds"], st
    assert st["verify_mismatch"] > 0
    monkeypatch.setenv("PECANPY_AMD_VERIFY_TIGHT", "1")
    monkeypatch.setenv("PECANPY_AMD_VERIFY_CAP", "1000")
    eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 80, seed=1)
    st = eng.last_stats
    assert st["verify_checked"] <= 1000 * st["lane_rounds"]
    assert st["verify_dropped"] > 0 and st["verify_mismatch"] == 0


def test_sampled_verification_in_production(monkeypatch):
    """Round 6: WITHOUT any switch, ~1/1024 of the steps the interval decision settles are decided
    again by the float chain inside the same call (pw_stats.verify_checked); a mismatch sends the walk to the complete
    kernel and is reported.  PECANPY_AMD_VERIFY_SAMPLE_POISON falsifies every 1024th record (the first included): the check must
    see it, the walk is redone (same result: the lane kernel's walk was right), and the host layer warns."""
    import torch

    from pecanpy_amd import pecanpy as node2vec

    indptr, indices, _ = ring_lattice_csr(1 << 12, 256)
    eng = WalkEngine.from_csr(indptr, indices, None)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 32)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    monkeypatch.setenv("PECANPY_AMD_VERIFY_TIGHT", "1")
    full = eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 80, seed=1)
    settled = eng.last_stats["verify_checked"]
    monkeypatch.delenv("PECANPY_AMD_VERIFY_TIGHT")
    clean = eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 80, seed=1)
    st = dict(eng.last_stats)
    assert torch.equal(full, clean)
    assert settled // 2048 <= st["verify_checked"] <= settled // 512 + 8, (settled, st)     # the sample: ~1/1024
    assert st["verify_mismatch"] == 0 and st["verify_dropped"] == 0 and st["redo_walks"] == 0
    for form_env in ({}, {"PECANPY_AMD_LANE_CHAINS": "1"}, {"PECANPY_AMD_NO_CHAIN_QUEUE": "1"}):      # rounds / CHAINS / in place
        for k, v in form_env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("PECANPY_AMD_VERIFY_SAMPLE_POISON", "1")
        out = eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 80, seed=1)
        sp = dict(eng.last_stats)
        monkeypatch.delenv("PECANPY_AMD_VERIFY_SAMPLE_POISON")
        assert sp["verify_checked"] > 0 and 1 <= sp["verify_mismatch"] <= sp["verify_checked"] // 1024 + 1, (form_env, sp)
        assert torch.equal(out, clean)                                  # (the redone walks equal the lane kernel's)
        assert sp["total_steps"] == st["total_steps"]                   # ... and are not counted twice
        out = eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 80, seed=1)
        assert eng.last_stats["verify_mismatch"] == 0 and torch.equal(out, clean)
        for k in form_env:
            monkeypatch.delenv(k)
    monkeypatch.setenv("PECANPY_AMD_VERIFY_SAMPLE", "0")              # off
    eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 80, seed=1)
    assert eng.last_stats["verify_checked"] == 0
    monkeypatch.delenv("PECANPY_AMD_VERIFY_SAMPLE")
    # the host layer says so
    g = node2vec.SparseOTF.from_csr(indptr, indices, None, p=0.5, q=2, random_state=1)
    monkeypatch.setenv("PECANPY_AMD_VERIFY_SAMPLE_POISON", "1")
    with pytest.warns(RuntimeWarning, match="sampled interval decisions"):
        g.simulate_walks_array(32, 80)


@pytest.mark.parametrize("p,q", [(0.3, 1.7), (3.0, 0.37)])
@pytest.mark.parametrize("family", ["holme_kim", "bipartite_hubs", "ring_lattice_512"])
def test_float_form_interval_decision_verified_by_the_float_chain(family, p, q, monkeypatch):
    """Round 6: the FLOATS form (1/p or 1/q not a power of two) runs the interval decision too (lane_tight_values, in front of its
    float chains).  PECANPY_AMD_VERIFY_TIGHT=1 records every step it settles and the float32 chain over the whole row decides
    each of them again on the device: none may differ.  Without the switch a 1/1024 sample is checked the same way."""
    import torch

    build, num_walks, L = FAMILIES[family]
    indptr, indices, _ = build()
    n = indptr.size - 1
    num_walks = max(2, num_walks // 4)
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * num_walks)
    np.random.RandomState(5).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    eng = WalkEngine.from_csr(indptr, indices, None)
    monkeypatch.setenv("PECANPY_AMD_VERIFY_TIGHT", "1")
    a = eng.simulate_device("SparseOTF", p, q, False, d_starts, L, seed=2)
    st = dict(eng.last_stats)
    monkeypatch.delenv("PECANPY_AMD_VERIFY_TIGHT")
    b = eng.simulate_device("SparseOTF", p, q, False, d_starts, L, seed=2)
    sb = dict(eng.last_stats)
    assert torch.equal(a, b)
    assert st["lane_kernel"] == 2 and st["verify_mismatch"] == 0 and st["verify_dropped"] == 0, st
    assert st["verify_checked"] > 1000, st
    assert sb["verify_mismatch"] == 0 and sb["verify_checked"] <= st["verify_checked"] // 256 + 64, sb
    print(f"[verify] FLOATS {family} p={p} q={q}: {st['total_steps']} transitions, {st['ambiguous_steps']} left open by the bound, "
          f"{st['verify_checked']} settled by the interval decision and re-decided by the chain, {st['wave_chain_steps']} float chains")
    if family == "holme_kim" and p == 0.3:     # the check is live here too
        monkeypatch.setenv("PECANPY_AMD_VERIFY_TIGHT", "poison")
        c = eng.simulate_device("SparseOTF", p, q, False, d_starts, L, seed=2)
        sp = eng.last_stats
        assert torch.equal(a, c) and sp["verify_mismatch"] >= max(1, sp["verify_checked"] // 1024 - sp["verify_ties"])
