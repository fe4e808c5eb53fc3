"""Lane kernel (one walk per lane, csrc/walk_lanes.hip.h) and the input checks of pw_csr_create.

The lane kernel serves unit-weight CSR graphs with power-of-two 1/p, 1/q; every other GPU parity test with such
parameters already runs on it (the default).  Here: that it IS the kernel that ran, that it equals the
wave-per-walk kernel and the oracle on cases built to hit its branches (hub rows beyond the LDS window, directed
graphs with dead ends and missing reverse edges, first steps, overflow reads handed back), and the C-ABI checks."""
import os
import socket
import subprocess
import sys
import json

import numpy as np
import pytest

from oracle import pyoracle as orc
from pecanpy_amd.engine import PwError, WalkEngine
from pecanpy_amd.synth import csr_from_edges, rmat_csr

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wave_engine(indptr, indices, data, monkeypatch):
    monkeypatch.setenv("PECANPY_AMD_NO_LANES", "1")
    eng = WalkEngine.from_csr(indptr, indices, data)
    monkeypatch.delenv("PECANPY_AMD_NO_LANES")
    return eng


@pytest.mark.parametrize("scale,p,q", [(12, 0.5, 2), (13, 0.25, 4), (12, 2, 0.5), (12, 1, 1), (11, 4, 0.125), (12, 1, 0.25)])
def test_lane_kernel_runs_and_equals_oracle_and_wave_kernel(scale, p, q, monkeypatch):
    indptr, indices, data = rmat_csr(scale, seed=scale + 20)
    starts = orc.shuffled_starts(indptr.size - 1, 3, 5)
    want, ost = orc.walks_sparse_otf(indptr, indices, data, p, q, starts, 40, 5, return_stats=True)
    eng = WalkEngine.from_csr(indptr, indices, data)
    assert eng.index_info()["lane_list_entries"] > 0
    got = eng.simulate("SparseOTF", p, q, False, starts, 40, seed=5)
    st = dict(eng.last_stats)
    assert st["lane_kernel"] == 1
    assert np.array_equal(got, want)
    assert st["total_steps"] == ost.total_steps and st["overflow_reads"] == ost.overflow_reads
    # (mirrored overflow reads are stepped by the lane kernel itself, through the vertex's overflow line: no redo needed)
    wave = _wave_engine(indptr, indices, data, monkeypatch)
    got_w = wave.simulate("SparseOTF", p, q, False, starts, 40, seed=5)
    assert wave.last_stats["lane_kernel"] == 0
    assert np.array_equal(got_w, want)


def test_lane_kernel_not_used_outside_its_regime():
    indptr, indices, data = rmat_csr(10, seed=3)
    starts = orc.shuffled_starts(indptr.size - 1, 2, 1)
    _, _, wdata = rmat_csr(10, seed=3, weighted=True)
    weng = WalkEngine.from_csr(indptr, indices, wdata)
    weng.simulate("SparseOTF", 0.5, 2, False, starts, 20, seed=1)
    assert weng.last_stats["lane_kernel"] == 0 and weng.index_info()["lane_list_entries"] > 0   # (lists: yes -- the wave kernel scatters its masks from them)


@pytest.mark.parametrize("p,q", [(0.3, 1.7), (3.0, 0.37), (1.0, 1.3), (0.7, 1.0)])
def test_float_chain_lane_kernel_for_non_dyadic_p_q(p, q, monkeypatch):
    """1/p or 1/q not a power of two: no exact integer decision exists; the lane kernel's FLOATS form evaluates the
    reference's two float32 chains (row total, CDF search) per lane, every step.  Oracle, wave kernel and lane kernel
    agree on R-MAT graphs and on the hub graph (40 000-entry row, thousands of common neighbours per list)."""
    for graph in ("rmat12", "rmat13", "hub"):
        if graph == "hub":
            indptr, indices, data = _hub_graph(np.random.default_rng(21))
            n = indptr.size - 1
            starts = np.concatenate([np.zeros(100, dtype=np.uint32), np.random.default_rng(2).integers(0, n, 3000).astype(np.uint32)])
        else:
            indptr, indices, data = rmat_csr(int(graph[4:]), seed=31)
            starts = orc.shuffled_starts(indptr.size - 1, 3, 5)
        want, ost = orc.walks_sparse_otf(indptr, indices, data, p, q, starts, 30, 5, return_stats=True)
        eng = WalkEngine.from_csr(indptr, indices, data)
        got = eng.simulate("SparseOTF", p, q, False, starts, 30, seed=5)
        st = dict(eng.last_stats)
        assert st["lane_kernel"] == 2, graph
        assert np.array_equal(got, want), graph
        assert st["total_steps"] == ost.total_steps and st["overflow_reads"] == ost.overflow_reads
        wave = _wave_engine(indptr, indices, data, monkeypatch)
        assert np.array_equal(wave.simulate("SparseOTF", p, q, False, starts, 30, seed=5), want)
        assert wave.last_stats["lane_kernel"] == 0


def test_lane_kernel_hub_rows_beyond_the_lds_window(monkeypatch):
    """A 40k-degree hub with thousands of common neighbours: long bisections, ambiguous steps whose float chain
    slides the 16384-position LDS window, first steps on the hub."""
    rng = np.random.default_rng(3)
    n = 60000
    hub = np.arange(1, 40001)
    src = [np.zeros(hub.size, dtype=np.int64), rng.integers(1, n, 300000), np.full(3000, 7, dtype=np.int64)]
    dst = [hub, rng.integers(1, n, 300000), rng.integers(1, n, 3000)]
    s, d = np.concatenate(src), np.concatenate(dst)
    keep = s != d
    s, d = s[keep], d[keep]
    indptr, indices, data = csr_from_edges(np.concatenate([s, d]), np.concatenate([d, s]), n)
    starts = np.concatenate([np.zeros(64, dtype=np.uint32), rng.integers(0, n, 2000).astype(np.uint32)])
    want, ost = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 30, 9, return_stats=True)
    eng = WalkEngine.from_csr(indptr, indices, data)
    got = eng.simulate("SparseOTF", 0.5, 2, False, starts, 30, seed=9)
    assert eng.last_stats["lane_kernel"] == 1 and eng.last_stats["ambiguous_steps"] > 0
    assert np.array_equal(got, want)
    assert eng.last_stats["overflow_reads"] == ost.overflow_reads


def test_lane_kernel_directed_graph_with_dead_ends(monkeypatch):
    """Directed: reverse edges mostly missing (rev_pos = not found), sinks end walks early, the stream is
    re-addressed in repair passes that run the lane kernel on job lists."""
    rng = np.random.default_rng(8)
    n = 3000
    src = rng.integers(0, n, 24000)
    dst = rng.integers(0, n, 24000)
    keep = (src != dst) & (src % 50 != 0)            # 2 % of the vertices have no out-edges
    indptr, indices, data = csr_from_edges(src[keep], dst[keep], n)
    starts = orc.shuffled_starts(n, 2, 3)
    want, ost = orc.walks_sparse_otf(indptr, indices, data, 0.25, 4, starts, 12, 3, return_stats=True)
    eng = WalkEngine.from_csr(indptr, indices, data)
    got = eng.simulate("SparseOTF", 0.25, 4, False, starts, 12, seed=3)
    st = eng.last_stats
    assert st["lane_kernel"] == 1 and st["dead_end_walks"] > 0 and st["stream_addressing"] == 0
    assert np.array_equal(got, want)
    assert st["total_steps"] == ost.total_steps
    wave = _wave_engine(indptr, indices, data, monkeypatch)
    got_w = wave.simulate("SparseOTF", 0.25, 4, False, starts, 12, seed=3)
    assert wave.last_stats["stream_addressing"] == st["stream_addressing"]
    assert np.array_equal(got, got_w)


def test_lane_kernel_equals_wave_kernel_at_rmat18(monkeypatch):
    import torch

    indptr, indices, data = rmat_csr(18, seed=1)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 4)
    np.random.RandomState(1).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    lanes = WalkEngine.from_csr(indptr, indices, None)
    wave = _wave_engine(indptr, indices, None, monkeypatch)
    for p, q in ((0.5, 2.0), (4.0, 0.25)):
        a = lanes.simulate_device("SparseOTF", p, q, False, d_starts, 80, seed=2)
        sa = dict(lanes.last_stats)
        b = wave.simulate_device("SparseOTF", p, q, False, d_starts, 80, seed=2)
        assert sa["lane_kernel"] == 1 and wave.last_stats["lane_kernel"] == 0
        assert torch.equal(a, b)
        assert sa["total_steps"] == wave.last_stats["total_steps"]
        assert sa["overflow_reads"] == wave.last_stats["overflow_reads"]
        # a job array of a few jobs per resident lane runs in one in-place launch; with the queue rule forced the same
        # call takes rounds (parked walks, lanes_chain_kernel) and returns the same matrix
        assert sa["lane_rounds"] == 1
        monkeypatch.setenv("PECANPY_AMD_CHAIN_TAIL", "100000")
        c = lanes.simulate_device("SparseOTF", p, q, False, d_starts, 80, seed=2)
        monkeypatch.delenv("PECANPY_AMD_CHAIN_TAIL")
        assert lanes.last_stats["lane_rounds"] > 1 and torch.equal(a, c)


# ---- input validation at the C ABI --------------------------------------------------------------
def test_csr_create_rejects_unsorted_duplicate_and_out_of_range_rows():
    indptr = np.array([0, 3, 5, 6], dtype=np.uint32)
    good = np.array([0, 1, 2, 0, 2, 1], dtype=np.uint32)
    WalkEngine.from_csr(indptr, good, None).close()
    unsorted = good.copy()
    unsorted[[0, 1]] = unsorted[[1, 0]]
    with pytest.raises(PwError, match="strictly ascending"):
        WalkEngine.from_csr(indptr, unsorted, None)
    dup = good.copy()
    dup[4] = 0                                            # row 1 = [0, 0]
    with pytest.raises(PwError, match="row 1"):
        WalkEngine.from_csr(indptr, dup, None)
    oob = good.copy()
    oob[5] = 3                                            # == n_nodes
    with pytest.raises(PwError, match="column index >= n_nodes"):
        WalkEngine.from_csr(indptr, oob, None)


def test_simulate_rejects_start_vertices_out_of_range():
    indptr, indices, data = rmat_csr(8, seed=1)
    eng = WalkEngine.from_csr(indptr, indices, data)
    starts = np.array([0, 5, indptr.size - 1], dtype=np.uint32)       # last one == n_nodes
    with pytest.raises(PwError, match="start vertex"):
        eng.simulate("SparseOTF", 0.5, 2, False, starts, 10, seed=0)


# ---- bench.py launches its own ranks --------------------------------------------------------------------------
def test_bench_self_launches_two_ranks_and_gathers():
    """`python bench.py --gpus 2` without a launcher (the driver's command line): two ranks are started, walk
    their shards on the one GPU of the box (gloo for the gather), and the gathered matrix equals a whole-array
    run (PECANPY_BENCH_VERIFY)."""
    env = dict(os.environ, PECANPY_BENCH_BACKEND="gloo", PECANPY_BENCH_ONE_GPU="1", PECANPY_BENCH_VERIFY="1")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--scale", "16", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "gathered shards verified" in res.stderr
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["gather_on_rank0"] is True
    assert len(rec["config"]["per_rank_walk_kernel_ms"]) == 2
    assert rec["roofline"]["frac"] <= 1.0


def test_weighted_normaliser_table_equals_the_two_pass_step(monkeypatch):
    """Weighted graphs: the per-edge normaliser table (built by the walk step's own pass 1) must not change a
    single transition -- node2vec and node2vec+, incl. rows longer than the mask segment."""
    import torch

    from pecanpy_amd import pecanpy as node2vec

    indptr, indices, data = rmat_csr(15, seed=4, weighted=True)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 3)
    np.random.RandomState(2).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    g = node2vec.SparseOTF.from_csr(indptr, indices, data, extend=True, gamma=0)
    with np.errstate(all="ignore"):
        thr = np.nan_to_num(g.get_noise_thresholds(), nan=0.0)
    table = WalkEngine.from_csr(indptr, indices, data)
    table.set_thresholds(thr)
    monkeypatch.setenv("PECANPY_AMD_NO_TOT", "1")
    plain = WalkEngine.from_csr(indptr, indices, data)
    plain.set_thresholds(thr)
    for extend, p, q in ((False, 0.5, 2.0), (True, 0.5, 2.0), (False, 0.3, 1.7), (True, 3.0, 0.4)):
        monkeypatch.setenv("PECANPY_AMD_NO_TOT", "1")
        b = plain.simulate_device("SparseOTF", p, q, extend, d_starts, 40, seed=7)
        assert plain.last_stats["param_index_ms"] == 0
        monkeypatch.delenv("PECANPY_AMD_NO_TOT")
        a = table.simulate_device("SparseOTF", p, q, extend, d_starts, 40, seed=7)
        assert table.last_stats["param_index_ms"] > 0                      # built for these parameters ...
        a2 = table.simulate_device("SparseOTF", p, q, extend, d_starts, 40, seed=7)
        assert table.last_stats["param_index_ms"] == 0                     # ... and cached
        assert torch.equal(a, b) and torch.equal(a, a2), (extend, p, q)
        assert table.last_stats["total_steps"] == plain.last_stats["total_steps"]
        assert table.last_stats["overflow_reads"] == plain.last_stats["overflow_reads"]


def _hub_graph(rng, n=60000, hub_deg=40000):
    hub = np.arange(1, hub_deg + 1)
    src = [np.zeros(hub.size, dtype=np.int64), rng.integers(1, n, 300000), np.full(3000, 7, dtype=np.int64)]
    dst = [hub, rng.integers(1, n, 300000), rng.integers(1, n, 3000)]
    s, d = np.concatenate(src), np.concatenate(dst)
    keep = s != d
    s, d = s[keep], d[keep]
    return csr_from_edges(np.concatenate([s, d]), np.concatenate([d, s]), n)


@pytest.mark.parametrize("p,q", [(0.5, 2), (4, 0.25)])
def test_parked_walks_and_chain_kernel_equal_in_place_chains_and_oracle(p, q, monkeypatch):
    """Steps that need the float32 chain: (a) the walk is parked, the chains of the whole queue run in one launch
    (lanes_chain_kernel) and the next round resumes the walks -- forced here for every round (PECANPY_AMD_CHAIN_TAIL=0);
    (b) the chain runs in place, inside the lane kernel (PECANPY_AMD_NO_CHAIN_QUEUE=1); (c) the default mix.  All three
    give the oracle's walks and count the same chain steps."""
    rng = np.random.default_rng(13)
    indptr, indices, data = _hub_graph(rng)
    n = indptr.size - 1
    starts = np.concatenate([np.zeros(200, dtype=np.uint32), rng.integers(0, n, 6000).astype(np.uint32)])
    want, ost = orc.walks_sparse_otf(indptr, indices, data, p, q, starts, 24, 4, return_stats=True)
    eng = WalkEngine.from_csr(indptr, indices, data)
    runs = {}
    for name, env in (("queue", {"PECANPY_AMD_CHAIN_TAIL": "0"}), ("in_place", {"PECANPY_AMD_NO_CHAIN_QUEUE": "1"}), ("default", {})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = eng.simulate("SparseOTF", p, q, False, starts, 24, seed=4)
        for k in env:
            monkeypatch.delenv(k)
        st = dict(eng.last_stats)
        assert st["lane_kernel"] == 1
        assert np.array_equal(got, want), name
        assert st["total_steps"] == ost.total_steps and st["overflow_reads"] == ost.overflow_reads
        runs[name] = st
    assert runs["queue"]["lane_rounds"] > 1 and runs["in_place"]["lane_rounds"] == 1
    # the queueing form defers the interval decision through a pool in LDS; a step that finds the pool full is parked
    # UNDECIDED and settled by the chain kernel (round 4), so it may run a few more chains than the in-place form -- never
    # fewer, and the ambiguous steps are the same set
    assert runs["in_place"]["wave_chain_steps"] > 0
    assert runs["in_place"]["wave_chain_steps"] <= runs["queue"]["wave_chain_steps"] <= 2 * runs["in_place"]["wave_chain_steps"]
    assert runs["queue"]["ambiguous_steps"] == runs["in_place"]["ambiguous_steps"] > runs["queue"]["wave_chain_steps"]


@pytest.mark.parametrize("p,q", [(0.5, 2), (0.25, 4)])
def test_chains_form_equals_the_rounds_and_the_oracle(p, q, monkeypatch):
    """Round 5, CHAINS form: job arrays of 1..16 jobs per resident lane (rounds 5-6: 32) run in ONE launch whose wavefronts run the float chains
    themselves, from the pool (no parking, no chain kernel, no rounds).  Same walks as the queueing form with its rounds, as the
    plain in-place launch, and as the oracle on a prefix -- on an R-MAT graph large enough for the form to be picked by the
    engine's own rule, and on the hub graph forced into it (chains on long rows, overflow lists searched sector by sector)."""
    import torch

    indptr, indices, data = rmat_csr(17, seed=5)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 10)
    np.random.RandomState(1).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    eng = WalkEngine.from_csr(indptr, indices, data)
    runs = {}
    for name, env in (("chains", {"PECANPY_AMD_LANE_CHAINS": "1"}), ("rounds", {"PECANPY_AMD_LANE_CHAINS": "0", "PECANPY_AMD_CHAIN_TAIL": "65536"}),
                      ("in_place", {"PECANPY_AMD_LANE_CHAINS": "0", "PECANPY_AMD_NO_CHAIN_QUEUE": "1"}), ("default", {})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = eng.simulate_device("SparseOTF", p, q, False, d_starts, 80, seed=7)
        for k in env:
            monkeypatch.delenv(k)
        runs[name] = (out, dict(eng.last_stats))
    want, ost = orc.walks_sparse_otf(indptr, indices, data, p, q, starts[:8000], 80, 7, return_stats=True)
    for name, (out, st) in runs.items():
        assert st["lane_kernel"] == 1 and st["redo_walks"] == 0, (name, st)
        assert torch.equal(out, runs["chains"][0]), name
        assert st["total_steps"] == runs["chains"][1]["total_steps"]
    assert np.array_equal(runs["chains"][0][:8000].cpu().numpy().view(np.uint32), want)
    assert runs["chains"][1]["lane_rounds"] == 1 and runs["rounds"][1]["lane_rounds"] > 1
    assert runs["chains"][1]["wave_chain_steps"] > 0
    assert runs["default"][1]["lane_rounds"] == 1          # (1.3 M jobs: the engine's rule picks the CHAINS form)
    # the hub graph: long rows, most steps ambiguous, many chains per wavefront
    rng = np.random.default_rng(13)
    indptr, indices, data = _hub_graph(rng)
    hn = indptr.size - 1
    hstarts = np.concatenate([np.zeros(300, dtype=np.uint32), rng.integers(0, hn, 30000).astype(np.uint32)])
    hwant = orc.walks_sparse_otf(indptr, indices, data, p, q, hstarts, 24, 4)
    heng = WalkEngine.from_csr(indptr, indices, data)
    monkeypatch.setenv("PECANPY_AMD_LANE_CHAINS", "1")
    got = heng.simulate("SparseOTF", p, q, False, hstarts, 24, seed=4)
    monkeypatch.delenv("PECANPY_AMD_LANE_CHAINS")
    assert heng.last_stats["lane_kernel"] == 1 and heng.last_stats["wave_chain_steps"] > 0
    assert np.array_equal(got, hwant)


def test_late_rounds_in_the_chains_form_equal_the_plain_rounds_and_the_oracle(monkeypatch):
    """Round 6: a round that resumes few walks per resident lane runs in the CHAINS form (its float chains inside the launch), so
    the tail of a pass is one launch instead of lane round / chain launch / lane round ...  Same walks as the plain rounds
    (PECANPY_AMD_LATE_CHAINS=0) and as the oracle on a prefix, in fewer rounds -- on a job array large enough for the rounds."""
    import torch

    indptr, indices, data = rmat_csr(19, seed=3)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 20)          # 10.5 M jobs: beyond the CHAINS form's own range
    np.random.RandomState(2).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    eng = WalkEngine.from_csr(indptr, indices, data)
    runs = {}
    for name, late in (("plain", "0"), ("late", "64"), ("default", None)):
        if late is not None:
            monkeypatch.setenv("PECANPY_AMD_LATE_CHAINS", late)
        out = eng.simulate_device("SparseOTF", 0.5, 2, False, d_starts, 80, seed=11)
        if late is not None:
            monkeypatch.delenv("PECANPY_AMD_LATE_CHAINS")
        runs[name] = (out, dict(eng.last_stats))
    for name, (out, st) in runs.items():
        assert st["lane_kernel"] == 1 and st["redo_walks"] == 0, (name, st)
        assert torch.equal(out, runs["plain"][0]), name
        assert st["total_steps"] == runs["plain"][1]["total_steps"]
    assert runs["plain"][1]["lane_rounds"] >= 3
    assert 1 < runs["late"][1]["lane_rounds"] < runs["plain"][1]["lane_rounds"]
    assert runs["default"][1]["lane_rounds"] <= runs["plain"][1]["lane_rounds"]
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts[:6000], 80, 11)
    assert np.array_equal(runs["late"][0][:6000].cpu().numpy().view(np.uint32), want)


def test_parked_walks_in_repair_passes_on_job_lists(monkeypatch):
    """Directed graph with sinks: the repair passes run the lane kernel on job lists; with every chain step parked the
    rounds resume walks of a job list."""
    rng = np.random.default_rng(8)
    n = 3000
    src = rng.integers(0, n, 24000)
    dst = rng.integers(0, n, 24000)
    keep = (src != dst) & (src % 50 != 0)
    indptr, indices, data = csr_from_edges(src[keep], dst[keep], n)
    starts = orc.shuffled_starts(n, 2, 3)
    eng = WalkEngine.from_csr(indptr, indices, data)
    ref = eng.simulate("SparseOTF", 0.25, 4, False, starts, 12, seed=3)
    st0 = dict(eng.last_stats)
    monkeypatch.setenv("PECANPY_AMD_CHAIN_TAIL", "0")
    got = eng.simulate("SparseOTF", 0.25, 4, False, starts, 12, seed=3)
    monkeypatch.delenv("PECANPY_AMD_CHAIN_TAIL")
    assert np.array_equal(got, ref)
    assert eng.last_stats["total_steps"] == st0["total_steps"] and eng.last_stats["repair_rounds"] == st0["repair_rounds"]
    assert st0["stream_addressing"] == 0
    want = orc.walks_sparse_otf(indptr, indices, data, 0.25, 4, starts, 12, 3)
    assert np.array_equal(got, want)


def test_library_warmup_entry_point():
    """pw_warmup: the library's one-time start-up as a call of its own (round 6; Base.__init__ runs it on a helper thread beside
    the graph read).  It succeeds on a visible device, reports its wall clock, rejects a device that is not there, and the host
    layer's helper thread ends with the time recorded."""
    import ctypes as C

    from pecanpy_amd import _lib

    lib = _lib.load()
    ms = C.c_double(-1.0)
    assert lib.pw_warmup(C.c_int(0), C.byref(ms)) == 0 and ms.value >= 0.0
    assert lib.pw_warmup(C.c_int(0), None) == 0
    assert lib.pw_warmup(C.c_int(lib.pw_device_count()), C.byref(ms)) != 0
    _lib.warmup_async(0)
    t = _lib._warm["thread"]
    assert t is not None
    t.join(timeout=60.0)
    assert not t.is_alive() and _lib.warmup_ms() is not None


def test_graph_handle_releases_its_device_memory(monkeypatch):
    """pw_graph_destroy frees everything a handle allocated on the way -- index, stream, redo list, the queues of parked
    walks: creating, walking and closing handles in a loop must not eat device memory."""
    import torch

    indptr, indices, data = rmat_csr(15, seed=2)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 4)
    monkeypatch.setenv("PECANPY_AMD_CHAIN_TAIL", "0")       # the queues are allocated
    torch.cuda.synchronize()
    free = []
    for _ in range(4):
        eng = WalkEngine.from_csr(indptr, indices, None)
        eng.simulate("SparseOTF", 0.5, 2, False, starts, 40, seed=1)
        assert eng.last_stats["lane_kernel"] == 1 and eng.last_stats["lane_rounds"] > 1
        eng.close()
        free.append(torch.cuda.mem_get_info()[0])
    assert free[0] - free[-1] < (32 << 20), free


def test_overflow_reads_step_through_the_vertex_overflow_line(monkeypatch):
    """choice == degree (the float32 CDF falls short of r): the reference reads the first neighbour of the next non-empty
    row (App. D quirk 1) -- a vertex that depends on cur only, so the pair (cur, that vertex) has a line of its own in the
    lane index (lines[nnz + cur]) and the lane kernel steps through the read itself.  Same walks as the oracle and as a
    handle without those lines (PECANPY_AMD_NO_VLINES: the wave kernel finishes such walks); overflow reads on hub rows,
    on first steps, and the clamped read at the end of the index array (always the wave kernel's)."""
    rng = np.random.default_rng(5)
    indptr, indices, data = _hub_graph(rng, n=70000, hub_deg=60000)
    n = indptr.size - 1
    starts = np.concatenate([np.zeros(4000, dtype=np.uint32), rng.integers(0, n, 20000).astype(np.uint32)])
    seen = 0
    for p, q in ((0.5, 2.0), (0.25, 4.0), (1.0, 0.25)):
        want, ost = orc.walks_sparse_otf(indptr, indices, data, p, q, starts, 60, 7, return_stats=True)
        seen += ost.overflow_reads                           # (how often a row's float32 CDF falls short depends on its total)
        eng = WalkEngine.from_csr(indptr, indices, data)
        got = eng.simulate("SparseOTF", p, q, False, starts, 60, seed=7)
        st = dict(eng.last_stats)
        assert np.array_equal(got, want)
        assert st["overflow_reads"] == ost.overflow_reads and st["total_steps"] == ost.total_steps
        assert st["redo_walks"] < st["overflow_reads"] or ost.overflow_reads == 0   # stepped in the lane kernel, not handed over
        monkeypatch.setenv("PECANPY_AMD_NO_VLINES", "1")
        plain = WalkEngine.from_csr(indptr, indices, data)
        monkeypatch.delenv("PECANPY_AMD_NO_VLINES")
        got2 = plain.simulate("SparseOTF", p, q, False, starts, 60, seed=7)
        assert np.array_equal(got2, want)
        assert plain.last_stats["redo_walks"] >= plain.last_stats["overflow_reads"] == ost.overflow_reads
    assert seen > 20


@pytest.mark.parametrize("extend,gamma,p,q", [(False, 0, 0.5, 2), (False, 0, 0.3, 1.7), (True, 0, 0.5, 2), (True, 0.5, 1.5, 0.3),
                                               (True, 0, 4, 0.25), (False, 0, 1, 1)])
def test_weighted_lane_form_equals_the_oracle_and_the_wave_kernel(extend, gamma, p, q, monkeypatch):
    """The WEIGHTED form of the lane kernel (round 4: float64 prefix sums + a rigorous bound on the float32 chain, the rest
    parked for the wave-per-walk scan): weighted R-MAT graphs, node2vec and node2vec+ (gamma 0 / 0.5), dyadic and
    non-dyadic p, q -- bit-exact against the oracle and against the wave-per-walk kernel, most steps decided by the lane."""
    from pecanpy_amd import pecanpy as node2vec

    indptr, indices, data = rmat_csr(12, seed=5, weighted=True)
    n = indptr.size - 1
    thr = None
    if extend:
        g = node2vec.SparseOTF.from_csr(indptr, indices, data, extend=True, gamma=gamma)
        with np.errstate(all="ignore"):
            thr = np.nan_to_num(g.get_noise_thresholds(), nan=0.0)
    starts = orc.shuffled_starts(n, 10, 2)
    want, ost = orc.walks_sparse_otf(indptr, indices, data, p, q, starts, 40, 2, thr=thr, return_stats=True)
    eng = WalkEngine.from_csr(indptr, indices, data)
    if extend:
        eng.set_thresholds(thr)
    monkeypatch.setenv("PECANPY_AMD_CHAIN_TAIL", "0")          # queueing rounds whatever the size of the job array
    got = eng.simulate("SparseOTF", p, q, extend, starts, 40, seed=2)
    st = dict(eng.last_stats)
    assert st["lane_kernel"] == 3, st
    assert np.array_equal(got, want), (extend, p, q)
    assert (st["total_steps"], st["overflow_reads"]) == (ost.total_steps, ost.overflow_reads)
    assert 0 < st["eager_steps"] < 0.5 * st["total_steps"], st          # (first steps + what the bound leaves open)
    monkeypatch.setenv("PECANPY_AMD_NO_WLANES", "1")
    wave = eng.simulate("SparseOTF", p, q, extend, starts, 40, seed=2)
    assert eng.last_stats["lane_kernel"] == 0 and np.array_equal(wave, got)
    monkeypatch.delenv("PECANPY_AMD_NO_WLANES")
    monkeypatch.delenv("PECANPY_AMD_CHAIN_TAIL")
    small = eng.simulate("SparseOTF", p, q, extend, starts[:300], 40, seed=2)      # no queue: such walks go to walk_kernel
    assert np.array_equal(small, want[:300])


def test_negative_or_non_finite_weights_are_refused():
    """A weight that is negative, NaN or infinite makes the reference's probabilities w / w.sum() meaningless (it walks on without
    complaint: cumsum + searchsorted over whatever comes out).  The exact scans of this library assume weights >= 0 (ADVICE r05:
    WeightedRow::margin takes the sign of a common neighbour's delta from q alone; the wave kernel's partial sums are monotone), so
    pw_csr_create refuses such a graph loudly instead of walking it differently; zero weights are fine."""
    from pecanpy_amd import _lib

    indptr, indices, data = rmat_csr(10, seed=6, weighted=True)
    for bad in (-0.25, float("nan"), float("inf")):
        d = data.copy()
        d[37] = bad
        with pytest.raises(_lib.PwError, match="finite and >= 0"):
            WalkEngine.from_csr(indptr, indices, d)
    d = data.copy()
    d[37] = 0.0
    eng = WalkEngine.from_csr(indptr, indices, d)
    starts = orc.shuffled_starts(indptr.size - 1, 2, 3)
    with np.errstate(all="ignore"):
        want = orc.walks_sparse_otf(indptr, indices, d, 0.5, 2.0, starts, 20, 5)
    assert np.array_equal(eng.simulate("SparseOTF", 0.5, 2.0, False, starts, 20, seed=5), want)


def test_weighted_lane_form_on_a_directed_graph_with_dead_ends(monkeypatch):
    """Weighted DIRECTED graphs through the weighted lane form: entries without a reverse edge (prev is not in cur's row),
    dead ends that shorten walks and shift the stream addresses (repair passes: job lists, wave kernel).  A graph with a
    single rarely-reached sink and a sink-heavy one (block-wise repair): the oracle's walks, draw for draw, and equal to the
    wave-per-walk kernel."""
    rng = np.random.default_rng(21)
    m = 4000
    monkeypatch.setenv("PECANPY_AMD_CHAIN_TAIL", "0")
    monkeypatch.setenv("PECANPY_AMD_FORCE_TOT", "1")
    for sinks in ("one", "many"):
        src, dst = rng.integers(0, m, 90000), rng.integers(0, m, 90000)
        keep = (src != dst) & ((src != 1234) if sinks == "one" else (src % 97 != 0))
        if sinks == "one":
            keep &= ~((dst == 1234) & (rng.random(dst.size) < 0.8))      # ... with few in-edges
        indptr, indices, _ = csr_from_edges(src[keep], dst[keep], m)
        data = (rng.random(indices.size) * 0.999 + 0.001).astype(np.float32)
        starts = orc.shuffled_starts(m, 6, 5)
        L = 10 if sinks == "one" else 30
        eng = WalkEngine.from_csr(indptr, indices, data)
        for p, q in ((0.5, 2.0), (1.3, 0.7)):
            want, ost = orc.walks_sparse_otf(indptr, indices, data, p, q, starts, L, 5, return_stats=True)
            got = eng.simulate("SparseOTF", p, q, False, starts, L, seed=5)
            st = dict(eng.last_stats)
            assert st["lane_kernel"] == 3 and st["dead_end_walks"] > 0, st
            assert st["stream_addressing"] == 0, st          # (exact whatever the number of sinks: block-wise repair)
            assert np.array_equal(got, want), (sinks, p, q)
            assert st["total_steps"] == ost.total_steps
            monkeypatch.setenv("PECANPY_AMD_NO_WLANES", "1")
            wave = eng.simulate("SparseOTF", p, q, False, starts, L, seed=5)
            assert eng.last_stats["lane_kernel"] == 0 and eng.last_stats["stream_addressing"] == st["stream_addressing"]
            monkeypatch.delenv("PECANPY_AMD_NO_WLANES")
            assert np.array_equal(got, wave), (sinks, p, q)


def test_weighted_halves_on_twin_contexts_equal_one_context(monkeypatch):
    """Round 6: weighted job arrays of a million walks or more are walked as two halves on two call contexts of the same
    handle (graph, index and per-(p, q) tables shared), so that one half's eager kernel runs beside the other half's lane
    round.  Same walks as one context (PECANPY_AMD_NO_TWIN=1) and as the oracle -- node2vec and node2vec+, and on a DIRECTED
    weighted graph whose dead ends in the first half move the second half's place in the stream."""
    import torch

    from oracle import pyoracle as orc
    from pecanpy_amd import pecanpy as node2vec
    from pecanpy_amd.synth import csr_from_edges, hash_edge_weights

    indptr, indices, data = rmat_csr(16, seed=6, weighted=True)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 17)           # 1.1 M jobs
    np.random.RandomState(3).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    g = node2vec.SparseOTF.from_csr(indptr, indices, data, extend=True, gamma=0)
    with np.errstate(all="ignore"):
        thr = np.nan_to_num(g.get_noise_thresholds(), nan=0.0)
    eng = WalkEngine.from_csr(indptr, indices, data)
    eng.set_thresholds(thr)
    for extend in (False, True):
        a = eng.simulate_device("SparseOTF", 0.5, 2.0, extend, d_starts, 40, seed=4)
        sa = dict(eng.last_stats)
        monkeypatch.setenv("PECANPY_AMD_NO_TWIN", "1")
        b = eng.simulate_device("SparseOTF", 0.5, 2.0, extend, d_starts, 40, seed=4)
        sb = dict(eng.last_stats)
        monkeypatch.delenv("PECANPY_AMD_NO_TWIN")
        assert torch.equal(a, b), extend
        assert sa["lane_kernel"] == 3 and sb["lane_kernel"] == 3
        assert (sa["total_steps"], sa["overflow_reads"], sa["eager_steps"]) == (sb["total_steps"], sb["overflow_reads"], sb["eager_steps"])
        want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2.0, starts[:1500], 40, 4, thr=thr if extend else None)
        assert np.array_equal(a[:1500].cpu().numpy().view(np.uint32), want), extend
    # the last rows (second half, second context) against the oracle at their stream offset
    full_steps = (a[:, -1].long() - 1).cumsum(0)
    k = starts.size - 800
    skip = int(full_steps[k - 1].item())
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2.0, starts[k:], 40, 4, thr=thr, stream_skip=skip)
    assert np.array_equal(a[k:].cpu().numpy().view(np.uint32), want)
    # directed + weighted, sinks
    rng = np.random.default_rng(2)
    m = 1 << 16
    src, dst = rng.integers(0, m, 1 << 20), rng.integers(0, m, 1 << 20)
    keep = (src != dst) & (src % 4096 != 0)          # (16 sinks: a few hundred dead ends -- every repair round re-walks on the wave kernel)
    ip, ix, _ = csr_from_edges(src[keep], dst[keep], m)
    w = hash_edge_weights(ip, ix, 5)
    st2 = np.concatenate([np.arange(m, dtype=np.uint32)] * 17)
    np.random.RandomState(1).shuffle(st2)
    d2 = torch.from_numpy(st2.view(np.int32)).cuda()
    e2 = WalkEngine.from_csr(ip, ix, w)
    a = e2.simulate_device("SparseOTF", 0.5, 2.0, False, d2, 20, seed=7)
    sa = dict(e2.last_stats)
    monkeypatch.setenv("PECANPY_AMD_NO_TWIN", "1")
    b = e2.simulate_device("SparseOTF", 0.5, 2.0, False, d2, 20, seed=7)
    monkeypatch.delenv("PECANPY_AMD_NO_TWIN")
    assert torch.equal(a, b) and sa["dead_end_walks"] > 0 and sa["stream_addressing"] == 0
    assert sa["total_steps"] == e2.last_stats["total_steps"]
    want = orc.walks_sparse_otf(ip, ix, w, 0.5, 2.0, st2[:1500], 20, 7)
    assert np.array_equal(a[:1500].cpu().numpy().view(np.uint32), want)
