"""Full-size checks of the HIP path through size-independent properties (no oracle at this size):
determinism, every sampled transition is a CSR edge (up to the counted overflow reads), the
bookkeeping cells, and shard invariance of the single-stream addressing."""
import os

import numpy as np
import pytest

from pecanpy_amd.engine import WalkEngine
from pecanpy_amd.synth import rmat_csr

pytestmark = pytest.mark.gpu
SCALE, W, L, SEED = int(os.environ.get("PECANPY_TEST_SCALE", "22")), 10, 80, 0   # BASELINE size


@pytest.fixture(scope="module")
def run():
    import torch

    indptr, indices, data = rmat_csr(SCALE, seed=1)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * W)
    np.random.RandomState(SEED).shuffle(starts)
    eng = WalkEngine.from_csr(indptr, indices, data)
    dev = torch.device("cuda", 0)
    d_starts = torch.from_numpy(starts.view(np.int32)).to(dev)
    out = eng.simulate_device("SparseOTF", 0.5, 2, False, d_starts, L, seed=SEED)
    stats = dict(eng.last_stats)
    return dict(indptr=indptr, indices=indices, starts=starts, eng=eng, d_starts=d_starts, out=out,
                stats=stats, dev=dev)


def test_full_size_determinism_and_bookkeeping(run):
    import torch

    out = run["out"]
    again = run["eng"].simulate_device("SparseOTF", 0.5, 2, False, run["d_starts"], L, seed=SEED)
    assert torch.equal(out, again)
    w = out.long() & 0xFFFFFFFF
    deg = torch.from_numpy(np.diff(run["indptr"].astype(np.int64))).to(run["dev"])
    st = torch.from_numpy(run["starts"].astype(np.int64)).to(run["dev"])
    assert torch.equal(w[:, 0], st)
    isolated = deg[st] == 0
    assert torch.all(w[isolated, L + 1] == 1) and torch.all(w[~isolated, L + 1] == L + 1)  # undirected: no dead ends
    assert torch.all(w[isolated, 1:L + 1] == 0)
    assert int((w[:, L + 1] - 1).sum().item()) == run["stats"]["total_steps"]


def test_full_size_every_transition_is_an_edge(run):
    import torch

    dev = run["dev"]
    n = run["indptr"].size - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(run["indptr"].astype(np.int64)))
    keys = torch.from_numpy(rows * n + run["indices"].astype(np.int64)).to(dev)  # sorted (CSR order)
    w = run["out"].long() & 0xFFFFFFFF
    bad = 0
    chunk = 1 << 19
    for lo in range(0, w.shape[0], chunk):
        blk = w[lo:lo + chunk]
        full = blk[:, L + 1] == L + 1
        blk = blk[full]
        q = (blk[:, :L] * n + blk[:, 1:L + 1]).reshape(-1)
        pos = torch.searchsorted(keys, q)
        pos = pos.clamp(max=keys.numel() - 1)
        bad += int((keys[pos] != q).sum().item())
    # the only non-edges are the mirrored "choice == degree" reads of the reference (App. D quirk 1)
    assert bad <= run["stats"]["overflow_reads"]
    assert run["stats"]["overflow_reads"] < 1e-4 * run["stats"]["total_steps"]


def test_full_size_shard_invariance(run):
    import torch

    starts = run["starts"]
    n_jobs = starts.size
    lo, hi = (3 * n_jobs) // 8, (4 * n_jobs) // 8
    skip = run["eng"].count_stream_draws(starts[:lo], L)
    shard = run["eng"].simulate_device("SparseOTF", 0.5, 2, False, run["d_starts"][lo:hi].contiguous(), L,
                                       seed=SEED, stream_skip=skip)
    assert torch.equal(shard, run["out"][lo:hi])


def test_full_size_lazy_step_equals_eager_step(run, monkeypatch):
    """The lazy step (per-edge common-neighbour counts, progressive membership, exact-arithmetic
    decision of the CDF search) must reproduce the eager float-chain step draw for draw: 1.6e9
    transitions at RMAT-22, far more boundary cases than the oracle-sized parity tests can reach."""
    import torch

    monkeypatch.setenv("PECANPY_AMD_NO_LAZY", "1")   # read by pw_csr_create: no `tri`, eager step only
    eager = WalkEngine.from_csr(run["indptr"], run["indices"], None)
    monkeypatch.delenv("PECANPY_AMD_NO_LAZY")
    out = eager.simulate_device("SparseOTF", 0.5, 2, False, run["d_starts"], L, seed=SEED)
    assert eager.last_stats["total_steps"] == run["stats"]["total_steps"]
    assert eager.last_stats["overflow_reads"] == run["stats"]["overflow_reads"]
    assert torch.equal(out, run["out"])
    for p, q in ((0.25, 4.0), (2.0, 0.5), (1.0, 1.0), (1.0, 0.25), (4.0, 0.125), (0.125, 8.0)):
        a = run["eng"].simulate_device("SparseOTF", p, q, False, run["d_starts"][: 1 << 21].contiguous(), L, seed=3)
        b = eager.simulate_device("SparseOTF", p, q, False, run["d_starts"][: 1 << 21].contiguous(), L, seed=3)
        assert torch.equal(a, b), (p, q)


@pytest.mark.parametrize("p,q", [(0.5, 2.0), (0.25, 4.0), (2.0, 0.5)])
def test_full_size_interval_decision_verified_by_the_float_chain(run, p, q, monkeypatch):
    """PECANPY_AMD_VERIFY_TIGHT=1 at BASELINE size: every step the interval decision (lane_tight) settles -- a tenth of
    the 1.6e9 transitions -- is decided again on the device by the sequential float32 chain; no step may differ
    (tests/test_gpu_verify.py runs the same check on graph families built to provoke rounding coincidences)."""
    import torch

    monkeypatch.setenv("PECANPY_AMD_VERIFY_TIGHT", "1")
    out = run["eng"].simulate_device("SparseOTF", p, q, False, run["d_starts"], L, seed=SEED)
    st = dict(run["eng"].last_stats)
    assert st["lane_kernel"] == 1
    assert st["verify_mismatch"] == 0 and st["verify_dropped"] == 0, st
    assert st["verify_checked"] > 0.02 * st["total_steps"], st
    assert st["verify_checked"] + st["wave_chain_steps"] >= st["ambiguous_steps"] - st["redo_walks"] * L
    if (p, q) == (0.5, 2.0):
        assert torch.equal(out, run["out"])
    print(f"[verify] RMAT-{SCALE} p={p} q={q}: {st['total_steps']} transitions, {st['verify_checked']} re-decided by the chain, "
          f"{st['verify_ties']} declined")


@pytest.mark.parametrize("p,q", [(0.3, 1.7), (3.0, 0.37)])
def test_full_size_float_form_interval_decision_verified(run, p, q, monkeypatch):
    """... and for the FLOATS form (round 6: lane_tight_values in front of the float chains), at BASELINE size: every step
    the interval decision settles -- a tenth of the 1.6e9 transitions -- decided again by the float32 chain over the whole row."""
    monkeypatch.setenv("PECANPY_AMD_VERIFY_TIGHT", "1")
    run["eng"].simulate_device("SparseOTF", p, q, False, run["d_starts"], L, seed=SEED)
    st = dict(run["eng"].last_stats)
    assert st["lane_kernel"] == 2
    assert st["verify_mismatch"] == 0 and st["verify_dropped"] == 0, st
    assert st["verify_checked"] > 0.02 * st["total_steps"], st
    print(f"[verify] FLOATS RMAT-{SCALE} p={p} q={q}: {st['total_steps']} transitions, {st['ambiguous_steps']} left open by the bound, "
          f"{st['verify_checked']} settled by the interval decision and re-decided by the chain, {st['wave_chain_steps']} float chains, "
          f"{st['verify_ties']} declined")


@pytest.mark.parametrize("p,q", [(0.5, 2.0), (0.25, 4.0)])
def test_full_size_oracle_prefix(run, p, q):
    """BASELINE size against the oracle itself: the first 20 000 jobs of the shuffled job array (stream offset 0,
    so they are exactly the head of the whole run) -- bit-exact walks and the same mirrored overflow reads."""
    import torch

    from oracle import pyoracle as orc

    n = 20000
    data = np.ones(run["indices"].size, dtype=np.float32)
    want, ost = orc.walks_sparse_otf(run["indptr"], run["indices"], data, p, q, run["starts"][:n], L, SEED, return_stats=True)
    got = run["eng"].simulate_device("SparseOTF", p, q, False, run["d_starts"][:n].contiguous(), L, seed=SEED)
    st = dict(run["eng"].last_stats)
    got = got.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    assert st["total_steps"] == ost.total_steps and st["overflow_reads"] == ost.overflow_reads
    if (p, q) == (0.5, 2.0):
        assert np.array_equal(run["out"][:n].cpu().numpy().view(np.uint32), want)
    else:
        # BASELINE C3 as it is run: the WHOLE 41.9 M-job array (the queueing / deferred form with its rounds -- the 20 000-job
        # call above is one in-place launch, a different instantiation of the kernel) laid against the same oracle prefix
        whole = run["eng"].simulate_device("SparseOTF", p, q, False, run["d_starts"], L, seed=SEED)
        ws = dict(run["eng"].last_stats)
        assert ws["lane_kernel"] == 1 and ws["lane_rounds"] > 1, ws
        assert np.array_equal(whole[:n].cpu().numpy().view(np.uint32), want)
        del whole


def test_c2_full_size_oracle_prefix():
    """BASELINE C2 as named (RMAT-18, 10 x 80, p = 0.5, q = 2): the whole job array in the form the engine picks for it -- ONE
    launch of the CHAINS form (round 5: 2.6 M jobs = 13 per resident lane; float chains run from the pool inside the launch;
    round 4: one in-place launch of the TAILS instantiation) -- against the oracle on its first 20 000 jobs."""
    import torch

    from oracle import pyoracle as orc

    indptr, indices, data = rmat_csr(18, seed=1)
    n_nodes = indptr.size - 1
    starts = np.concatenate([np.arange(n_nodes, dtype=np.uint32)] * W)
    np.random.RandomState(SEED).shuffle(starts)
    eng = WalkEngine.from_csr(indptr, indices, data)
    out = eng.simulate_device("SparseOTF", 0.5, 2, False, torch.from_numpy(starts.view(np.int32)).cuda(), L, seed=SEED)
    st = dict(eng.last_stats)
    assert st["lane_kernel"] == 1 and st["lane_rounds"] == 1 and st["redo_walks"] == 0, st
    assert "PECANPY_AMD_LANE_CHAINS" not in os.environ and "PECANPY_AMD_LANE_TAILS" not in os.environ   # (the engine's own choice of form)
    n = 20000
    want, ost = orc.walks_sparse_otf(indptr, indices, np.ones(indices.size, dtype=np.float32), 0.5, 2, starts[:n], L, SEED,
                                     return_stats=True)
    assert np.array_equal(out[:n].cpu().numpy().view(np.uint32), want)
    head = eng.simulate_device("SparseOTF", 0.5, 2, False, torch.from_numpy(starts[:n].view(np.int32)).cuda(), L, seed=SEED)
    hs = dict(eng.last_stats)
    assert np.array_equal(head.cpu().numpy().view(np.uint32), want)
    assert (hs["total_steps"], hs["overflow_reads"]) == (ost.total_steps, ost.overflow_reads)


@pytest.mark.parametrize("p,q", [(0.3, 1.7), (3.0, 0.37)])
def test_full_size_float_lane_form_equals_the_wave_kernel(run, p, q, monkeypatch):
    """The FLOATS form of the lane kernel (unit weights, 1/p or 1/q NOT a power of two: two closed-form float32 chains
    per lane and step, lane_chain<true>) at BASELINE size against the wave-per-walk kernel's eager step -- the complete
    float32 scan -- on 2^21 jobs (80 M transitions), and for one pair against the oracle on the first 20 000 jobs."""
    import torch

    from oracle import pyoracle as orc

    m = 1 << 21
    d = run["d_starts"][:m].contiguous()
    lane = run["eng"].simulate_device("SparseOTF", p, q, False, d, L, seed=3)
    st = dict(run["eng"].last_stats)
    assert st["lane_kernel"] == 2, st
    monkeypatch.setenv("PECANPY_AMD_NO_LANES", "1")
    wave = run["eng"].simulate_device("SparseOTF", p, q, False, d, L, seed=3)
    sw = dict(run["eng"].last_stats)
    monkeypatch.delenv("PECANPY_AMD_NO_LANES")
    assert sw["lane_kernel"] == 0, sw
    assert (st["total_steps"], st["overflow_reads"]) == (sw["total_steps"], sw["overflow_reads"])
    assert torch.equal(lane, wave), (p, q)
    print(f"[floats] RMAT-{SCALE} p={p} q={q}: {st['total_steps']} transitions, lane == wave; {st['redo_walks']} redo walks, "
          f"{st['overflow_reads']} overflow reads")
    if (p, q) == (0.3, 1.7):
        n = 20000
        data = np.ones(run["indices"].size, dtype=np.float32)
        want, ost = orc.walks_sparse_otf(run["indptr"], run["indices"], data, p, q, run["starts"][:n], L, 3, return_stats=True)
        assert np.array_equal(lane[:n].cpu().numpy().view(np.uint32), want)
        head = run["eng"].simulate_device("SparseOTF", p, q, False, run["d_starts"][:n].contiguous(), L, seed=3)
        hs = dict(run["eng"].last_stats)
        assert hs["lane_kernel"] == 2 and np.array_equal(head.cpu().numpy().view(np.uint32), want)
        assert (hs["total_steps"], hs["overflow_reads"]) == (ost.total_steps, ost.overflow_reads)


def test_full_size_oracle_slice_deep_in_the_stream(run):
    """The last shard of an 8-GPU run: 5 000 jobs starting at 7/8 of the shuffled job array, i.e. ~1.4e9 doubles into the
    seed's stream (the device's 45-entry jump tree at that depth), bit-exact against the sequential oracle, which skips
    the same number of draws one by one."""
    from oracle import pyoracle as orc

    starts = run["starts"]
    lo = (7 * starts.size) // 8
    n = 5000
    skip = run["eng"].count_stream_draws(starts[:lo], L)
    deg = np.diff(run["indptr"].astype(np.int64))
    assert skip == int((deg[starts[:lo]] > 0).sum()) * L and skip > 1.3e9
    data = np.ones(run["indices"].size, dtype=np.float32)
    want, ost = orc.walks_sparse_otf(run["indptr"], run["indices"], data, 0.5, 2, starts[lo:lo + n], L, SEED,
                                     stream_skip=skip, return_stats=True)
    got = run["eng"].simulate_device("SparseOTF", 0.5, 2, False, run["d_starts"][lo:lo + n].contiguous(), L, seed=SEED,
                                     stream_skip=skip)
    st = dict(run["eng"].last_stats)
    got = got.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, want)
    assert (st["total_steps"], st["overflow_reads"]) == (ost.total_steps, ost.overflow_reads)
    assert np.array_equal(run["out"][lo:lo + n].cpu().numpy().view(np.uint32), want)   # ... and the whole-array run


# ---- BASELINE C5: weighted RMAT-20, node2vec+ ------------------------------------------------------------------------
C5_SCALE = int(os.environ.get("PECANPY_TEST_C5_SCALE", "20"))


@pytest.fixture(scope="module")
def c5():
    import torch

    from pecanpy_amd import pecanpy as node2vec

    indptr, indices, data = rmat_csr(C5_SCALE, seed=1, weighted=True)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * W)
    np.random.RandomState(SEED).shuffle(starts)
    g = node2vec.SparseOTF.from_csr(indptr, indices, data, extend=True, gamma=0)
    with np.errstate(all="ignore"):
        thr = np.nan_to_num(g.get_noise_thresholds(), nan=0.0)
    eng = WalkEngine.from_csr(indptr, indices, data)
    eng.set_thresholds(thr)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    out = eng.simulate_device("SparseOTF", 0.5, 2, True, d_starts, L, seed=SEED)
    return dict(indptr=indptr, indices=indices, data=data, thr=thr, starts=starts, eng=eng, d_starts=d_starts, out=out,
                stats=dict(eng.last_stats))


def test_c5_full_size_properties_and_oracle_prefix(c5):
    """Weighted RMAT-20 with --extend at its named size: determinism, bookkeeping, every transition an edge, shard
    invariance, and the first 5 000 jobs bit-exact against the oracle."""
    import torch

    from oracle import pyoracle as orc

    eng, out = c5["eng"], c5["out"]
    again = eng.simulate_device("SparseOTF", 0.5, 2, True, c5["d_starts"], L, seed=SEED)
    assert torch.equal(out, again)
    w = out.long() & 0xFFFFFFFF
    n = c5["indptr"].size - 1
    deg = torch.from_numpy(np.diff(c5["indptr"].astype(np.int64))).cuda()
    st = torch.from_numpy(c5["starts"].astype(np.int64)).cuda()
    isolated = deg[st] == 0
    assert torch.equal(w[:, 0], st)
    assert torch.all(w[isolated, L + 1] == 1) and torch.all(w[~isolated, L + 1] == L + 1)
    assert int((w[:, L + 1] - 1).sum().item()) == c5["stats"]["total_steps"]
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(c5["indptr"].astype(np.int64)))
    keys = torch.from_numpy(rows * n + c5["indices"].astype(np.int64)).cuda()
    bad = 0
    for lo in range(0, w.shape[0], 1 << 19):
        blk = w[lo:lo + (1 << 19)]
        blk = blk[blk[:, L + 1] == L + 1]
        qk = (blk[:, :L] * n + blk[:, 1:L + 1]).reshape(-1)
        pos = torch.searchsorted(keys, qk).clamp(max=keys.numel() - 1)
        bad += int((keys[pos] != qk).sum().item())
    assert bad <= c5["stats"]["overflow_reads"]
    n_jobs = c5["starts"].size
    lo, hi = (5 * n_jobs) // 8, (6 * n_jobs) // 8
    skip = eng.count_stream_draws(c5["starts"][:lo], L)
    shard = eng.simulate_device("SparseOTF", 0.5, 2, True, c5["d_starts"][lo:hi].contiguous(), L, seed=SEED, stream_skip=skip)
    assert torch.equal(shard, out[lo:hi])
    m = 5000
    want = orc.walks_sparse_otf(c5["indptr"], c5["indices"], c5["data"], 0.5, 2, c5["starts"][:m], L, SEED, thr=c5["thr"])
    assert np.array_equal(out[:m].cpu().numpy().view(np.uint32), want)


# ---- BASELINE C4: Erdos-Renyi N = 100 000, density 0.25, DenseOTF --------------------------------------------------
def _er_bits(n, density, seed=1):
    import torch

    sys_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys

    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from bench import er_bits_gpu

    return er_bits_gpu(n, density, torch.device("cuda", 0), seed=seed)


def test_c4_full_size_properties():
    """ER-100k DenseOTF at its named size through the packed-bits kernel: determinism, bookkeeping, every
    transition a set adjacency bit, shard invariance."""
    import torch

    n = int(os.environ.get("PECANPY_TEST_C4_NODES", "100000"))
    bits, deg = _er_bits(n, 0.25)
    eng = WalkEngine.from_dense_bits(bits, n)
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * W)
    np.random.RandomState(SEED).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    out = eng.simulate_device("DenseOTF", 0.5, 2, False, d_starts, L, seed=SEED)
    stats = dict(eng.last_stats)
    again = eng.simulate_device("DenseOTF", 0.5, 2, False, d_starts, L, seed=SEED)
    assert torch.equal(out, again)
    w = out.long() & 0xFFFFFFFF
    assert torch.equal(w[:, 0], torch.from_numpy(starts.astype(np.int64)).cuda())
    assert torch.all(w[:, L + 1] == L + 1) and int(deg.min().item()) > 0
    assert stats["total_steps"] == starts.size * L
    bad = 0
    for lo in range(0, w.shape[0], 1 << 18):
        blk = w[lo:lo + (1 << 18)]
        u, v = blk[:, :L].reshape(-1), blk[:, 1:L + 1].reshape(-1)
        word = bits.view(-1)[u * bits.shape[1] + (v >> 6)]
        bad += int((((word >> (v & 63)) & 1) == 0).sum().item())
    assert bad <= stats["overflow_reads"] + stats["clamped_reads"]
    lo, hi = (2 * starts.size) // 8, (3 * starts.size) // 8
    shard = eng.simulate_device("DenseOTF", 0.5, 2, False, d_starts[lo:hi].contiguous(), L, seed=SEED, stream_skip=lo * L)
    assert torch.equal(shard, out[lo:hi])


def test_c4_full_size_fast_kernel_equals_the_complete_kernel_and_the_oracle():
    """BASELINE C4 at its real shape (N = 100 000, density 0.25: WPL = 25, ~25 000 set bits per row): the register-only
    walk_dense_fast_kernel against walk_dense_bits_kernel (the same decision + the float64 chain) on all 80 M steps,
    and against the packed-bits oracle (oracle/pecan_oracle.c: orc_walks_dense_otf_bits, pinned to the dense goldens)
    on the first 100 jobs x 80 steps."""
    import torch

    from oracle import pyoracle as orc

    n = int(os.environ.get("PECANPY_TEST_C4_NODES", "100000"))
    bits, deg = _er_bits(n, 0.25)
    eng = WalkEngine.from_dense_bits(bits, n)
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * W)
    np.random.RandomState(SEED).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    fast = eng.simulate_device("DenseOTF", 0.5, 2, False, d_starts, L, seed=SEED)
    st = dict(eng.last_stats)
    os.environ["PECANPY_AMD_DENSE_NO_FAST"] = "1"
    try:
        full = eng.simulate_device("DenseOTF", 0.5, 2, False, d_starts, L, seed=SEED)
        sf = dict(eng.last_stats)
    finally:
        del os.environ["PECANPY_AMD_DENSE_NO_FAST"]
    assert st["total_steps"] == sf["total_steps"] == starts.size * L
    assert st["redo_walks"] <= 16 and sf["redo_walks"] == 0, (st, sf)
    assert torch.equal(fast, full)
    m = 100
    hb = bits.cpu().numpy().view(np.uint64).reshape(n, -1)
    want, ost = orc.walks_dense_otf_bits(hb, n, 0.5, 2, starts[:m], L, SEED, return_stats=True)
    assert np.array_equal(fast[:m].cpu().numpy().view(np.uint32), want)
    assert ost.overflow_reads == 0
    # a non-dyadic pair takes the complete kernel alone (no exact decision without dyadic weights in float64? it has one:
    # the chain is the fallback) -- the same oracle
    want2 = orc.walks_dense_otf_bits(hb, n, 0.3, 1.7, starts[:40], L, 2)
    got2 = eng.simulate_device("DenseOTF", 0.3, 1.7, False, d_starts[:40].contiguous(), L, seed=2)
    assert np.array_equal(got2.cpu().numpy().view(np.uint32), want2)


def test_c4_density_oracle_prefix_on_a_20k_slice():
    """The same generator and kernel at N = 20 000, density 0.25 (rows of ~5 000 neighbours): the first 200 jobs
    bit-exact against the dense oracle on the unpacked float64 matrix."""
    import torch

    from oracle import pyoracle as orc

    n = 20000
    bits, _ = _er_bits(n, 0.25, seed=3)
    eng = WalkEngine.from_dense_bits(bits, n)
    hb = bits.cpu().numpy().view(np.uint64)
    adj = np.unpackbits(hb.view(np.uint8), bitorder="little").reshape(n, -1)[:, :n]
    mat = adj.astype(np.float64)
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 2)
    np.random.RandomState(5).shuffle(starts)
    starts = starts[:200]
    want = orc.walks_dense_otf(mat, 0.5, 2, starts, 40, 5, nonzero=adj.astype(bool))
    got = eng.simulate("DenseOTF", 0.5, 2, False, starts, 40, seed=5)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,density", [(9000, 0.05), (40000, 0.02), (70000, 0.01), (110000, 0.006)])
def test_dense_fast_kernel_equals_the_complete_kernel(n, density, monkeypatch):
    """walk_dense_fast_kernel (dyadic 1/p, 1/q: the exact decision alone, in registers) against walk_dense_bits_kernel
    (the same decision + the float64 chain behind it) on rows of 141 / 625 / 1094 / 1719 words -- the four register
    widths the kernel is instantiated for -- at three (p, q); then with every 7th walk handed over to the complete
    kernel at its third step (PECANPY_AMD_DENSE_REDO_TEST): the hand-over rewrites the whole row."""
    import torch

    bits, deg = _er_bits(n, density, seed=4)
    eng = WalkEngine.from_dense_bits(bits, n)
    starts = np.arange(n, dtype=np.uint32)[: 30000]
    np.random.RandomState(2).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    # (round 6: 1/p or 1/q not a power of two -- the last three pairs -- run the same kernel with float64 masses and the
    #  float64-bounded decision, BOUNDED; a partial sum inside the bound's interval, ~2 d^2 u of the steps, hands the walk over)
    for p, q in ((0.3, 1.7), (3.0, 0.37), (1.0, 0.9), (0.5, 2.0), (4.0, 0.25), (1.0, 1.0)):
        fast = eng.simulate_device("DenseOTF", p, q, False, d_starts, 40, seed=3)
        st = dict(eng.last_stats)
        monkeypatch.setenv("PECANPY_AMD_DENSE_NO_FAST", "1")
        full = eng.simulate_device("DenseOTF", p, q, False, d_starts, 40, seed=3)
        assert eng.last_stats["total_steps"] == st["total_steps"]
        monkeypatch.delenv("PECANPY_AMD_DENSE_NO_FAST")
        assert torch.equal(fast, full), (n, p, q)
        assert st["redo_walks"] <= 2, st                              # (the decision is decisive for all but ~1e-8 of the steps)
    monkeypatch.setenv("PECANPY_AMD_DENSE_REDO_TEST", "7")
    mixed = eng.simulate_device("DenseOTF", 1.0, 1.0, False, d_starts, 40, seed=3)
    assert eng.last_stats["redo_walks"] >= starts.size // 7
    assert eng.last_stats["total_steps"] == st["total_steps"]
    assert torch.equal(mixed, full)


def test_host_call_through_the_ring_at_scale(monkeypatch):
    """pw_simulate at a size where the ring of part buffers is the default (>= 800 MB of output: 8 parts, three buffers, a
    stream expansion per part): RMAT-20, 10 x 80 -- equal to the device-resident call row for row, and to the whole-matrix form."""
    import torch

    indptr, indices, data = rmat_csr(20, seed=1)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 10)
    np.random.RandomState(0).shuffle(starts)
    eng = WalkEngine.from_csr(indptr, indices, None)
    d = eng.simulate_device("SparseOTF", 0.5, 2.0, False, torch.from_numpy(starts.view(np.int32)).cuda(), 80, seed=3)
    want = d.cpu().numpy().view(np.uint32)
    del d
    got = eng.simulate("SparseOTF", 0.5, 2.0, False, starts, 80, seed=3)
    assert got.nbytes >= 800 << 20 and np.array_equal(got, want)
    monkeypatch.setenv("PECANPY_AMD_NO_RING", "1")
    assert np.array_equal(eng.simulate("SparseOTF", 0.5, 2.0, False, starts, 80, seed=3), want)
