"""In-process multi-device entry points of the C ABI (round 6): pw_csr_create_multi / pw_graph_replicate /
pw_simulate_multi -- one host thread per device inside one call, the index built once and copied device to device, shards
addressed into the ONE random stream (the reference is one process, src/pecanpy/cli.py:340-351).

The GPU boxes have ONE device, so the device list names device 0 two or three times: every entry is a replica with its own
streams and scratch, the shards run side by side on the same GPU, and the assembled matrix must equal the one-handle run bit
for bit.  What a real multi-GPU node adds -- the xGMI hop of the replication and of the peer copies -- is the same
hipMemcpyPeer call with two different device indices; it has NOT been measured on hardware (no multi-GPU node in any round)."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from pecanpy_amd.engine import MultiWalkEngine, WalkEngine, visible_devices
from pecanpy_amd.synth import csr_from_edges, hash_edge_weights, rmat_csr

pytestmark = pytest.mark.gpu


def test_two_replicas_on_one_device_equal_one_handle():
    indptr, indices, data = rmat_csr(14, seed=2)
    n = indptr.size - 1
    starts = orc.shuffled_starts(n, 10, 0)
    one = WalkEngine.from_csr(indptr, indices, data)
    multi = MultiWalkEngine.from_csr(indptr, indices, data, devices="0,0")
    assert multi.devices == [0, 0]
    for p, q, lane in ((0.5, 2.0, 1), (0.3, 1.7, 2)):
        want = one.simulate("SparseOTF", p, q, False, starts, 40, seed=5)
        st1 = dict(one.last_stats)
        got = multi.simulate("SparseOTF", p, q, False, starts, 40, seed=5)
        st2 = dict(multi.last_stats)
        assert np.array_equal(got, want)
        assert st2["lane_kernel"] == lane and st2["total_steps"] == st1["total_steps"] and st2["overflow_reads"] == st1["overflow_reads"]
    # ... with a stream offset (a shard of a larger job), and against the oracle on a prefix
    want = one.simulate("SparseOTF", 0.5, 2, False, starts[1000:], 40, seed=5, stream_skip=123457)
    got = multi.simulate("SparseOTF", 0.5, 2, False, starts[1000:], 40, seed=5, stream_skip=123457)
    assert np.array_equal(got, want)
    ref = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts[:3000], 40, 5)
    assert np.array_equal(multi.simulate("SparseOTF", 0.5, 2, False, starts, 40, seed=5)[:3000], ref)
    # the replica carries the same index (copied, not rebuilt)
    a, b = one.lane_index(), multi.engines[1].lane_index()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert multi.engines[1].index_info()["build_ms"] == 0.0 and multi.engines[0].index_info()["build_ms"] > 0.0


def test_three_replicas_device_output_and_uneven_shards():
    import torch

    indptr, indices, data = rmat_csr(13, seed=7)
    n = indptr.size - 1
    starts = orc.shuffled_starts(n, 5, 3)[: 5 * n - 11]          # (not divisible by three)
    one = WalkEngine.from_csr(indptr, indices, data)
    want = one.simulate("SparseOTF", 0.25, 4, False, starts, 30, seed=9)
    multi = MultiWalkEngine.from_engine(WalkEngine.from_csr(indptr, indices, data), [0, 0, 0])
    got = multi.simulate_to_device("SparseOTF", 0.25, 4, False, starts, 30, seed=9)
    assert got.is_cuda and np.array_equal(got.cpu().numpy().view(np.uint32), want)
    assert multi.last_stats["total_steps"] == one.last_stats["total_steps"]
    # a job array too small to shard runs on the first handle alone
    small = starts[:50]
    assert np.array_equal(multi.simulate("SparseOTF", 0.25, 4, False, small, 30, seed=9),
                          one.simulate("SparseOTF", 0.25, 4, False, small, 30, seed=9))
    # no seed: one seed is drawn for all shards (walks are valid; nothing to compare with)
    mat = multi.simulate("SparseOTF", 0.25, 4, False, starts, 30, seed=None)
    assert mat.shape == want.shape and (mat[:, 0] == starts).all()
    torch.cuda.synchronize()


def test_peer_copy_path_in_tapered_chunks(monkeypatch):
    """The path a replica on ANOTHER GPU takes (walk into local memory in 3 : 2 : 1 chunks, chunk c copied to the first device
    while chunk c + 1 is walked, chunk c + 1 addressed by the draws chunk c consumed), forced for replicas on the one device a
    test box has (PECANPY_AMD_MULTI_FORCE_PEER=1: hipMemcpyPeerAsync device 0 -> device 0) -- on a directed graph with sinks."""
    rng = np.random.default_rng(6)
    m = 6000
    src, dst = rng.integers(0, m, 80000), rng.integers(0, m, 80000)
    keep = (src != dst) & (src % 30 != 0)
    indptr, indices, data = csr_from_edges(src[keep], dst[keep], m)
    starts = orc.shuffled_starts(m, 6, 2)
    want = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 25, 2)
    multi = MultiWalkEngine.from_csr(indptr, indices, data, devices=[0, 0, 0])
    monkeypatch.setenv("PECANPY_AMD_MULTI_FORCE_PEER", "1")
    for chunks in ("1", "3", "5"):
        monkeypatch.setenv("PECANPY_AMD_MULTI_CHUNKS", chunks)
        got = multi.simulate_to_device("SparseOTF", 0.5, 2, False, starts, 25, seed=2)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want), chunks
        assert multi.last_stats["dead_end_walks"] > 0


def test_dead_ends_shift_the_later_shards():
    """Directed graph with sinks: a shard consumes fewer draws than it announced, so the shards behind it are walked again
    from the draws actually consumed -- the matrix is the oracle's (one sequential stream, pecanpy.py:198-206)."""
    rng = np.random.default_rng(5)
    m = 4000
    src, dst = rng.integers(0, m, 50000), rng.integers(0, m, 50000)
    keep = (src != dst) & (src % 25 != 0)          # every 25th vertex has no out-edges
    indptr, indices, data = csr_from_edges(src[keep], dst[keep], m)
    starts = orc.shuffled_starts(m, 4, 1)
    want, ost = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts, 30, 1, return_stats=True)
    multi = MultiWalkEngine.from_csr(indptr, indices, data, devices=[0, 0, 0])
    got = multi.simulate("SparseOTF", 0.5, 2, False, starts, 30, seed=1)
    assert np.array_equal(got, want)
    st = multi.last_stats
    assert st["dead_end_walks"] > 0 and st["stream_addressing"] == 0 and st["total_steps"] == ost.total_steps


def test_weighted_node2vec_plus_and_alias_modes():
    from pecanpy_amd import pecanpy as node2vec

    indptr, indices, _ = rmat_csr(12, seed=4)
    data = hash_edge_weights(indptr, indices, 2)
    g = node2vec.SparseOTF.from_csr(indptr, indices, data, extend=True, gamma=0)
    with np.errstate(all="ignore"):
        thr = np.nan_to_num(g.get_noise_thresholds(), nan=0.0)
    starts = orc.shuffled_starts(indptr.size - 1, 6, 2)
    one = WalkEngine.from_csr(indptr, indices, data)
    one.set_thresholds(thr)
    want = one.simulate("SparseOTF", 0.5, 2, True, starts, 30, seed=2)
    multi = MultiWalkEngine.from_engine(one, [0, 0])          # (thresholds set BEFORE replication travel with the replica)
    got = multi.simulate("SparseOTF", 0.5, 2, True, starts, 30, seed=2)
    assert np.array_equal(got, want)
    ref = orc.walks_sparse_otf(indptr, indices, data, 0.5, 2, starts[:2000], 30, 2, thr=thr)
    assert np.array_equal(got[:2000], ref)
    # alias modes draw a variable number of words per step: the first handle walks the whole array
    small = starts[:400]
    assert np.array_equal(multi.simulate("PreComp", 0.5, 2, False, small, 10, seed=3),
                          one.simulate("PreComp", 0.5, 2, False, small, 10, seed=3))


def test_host_api_spreads_over_the_named_devices(monkeypatch):
    """The drop-in classes: PECANPY_AMD_DEVICES names the devices of THIS process (no launcher); same walks as one device."""
    from pecanpy_amd import pecanpy as node2vec

    indptr, indices, data = rmat_csr(12, seed=9)
    g1 = node2vec.SparseOTF.from_csr(indptr, indices, data, p=0.5, q=2, random_state=4)
    want = g1.simulate_walks_array(4, 25)
    assert g1._multi is None
    monkeypatch.setenv("PECANPY_AMD_DEVICES", "0,0")
    g2 = node2vec.SparseOTF.from_csr(indptr, indices, data, p=0.5, q=2, random_state=4)
    got = g2.simulate_walks_array(4, 25)
    assert g2._multi is not None and g2._multi.devices == [0, 0]
    assert np.array_equal(got, want)
    assert g2.simulate_walks(1, 5)[:3] == g1.simulate_walks(1, 5)[:3]
    monkeypatch.setenv("PECANPY_AMD_DEVICES", "1")            # one device: the plain path
    g3 = node2vec.SparseOTF.from_csr(indptr, indices, data, p=0.5, q=2, random_state=4)
    assert np.array_equal(g3.simulate_walks_array(4, 25), want) and g3._multi is None


def test_device_lists():
    assert visible_devices("0,0") == [0, 0]
    assert visible_devices("mask:0x1") == [0]
    assert visible_devices(1) == [0]
    assert visible_devices(None)[0] == 0
    with pytest.raises(Exception):
        visible_devices("mask:0x8000000000000000")           # (device 63 is not visible)
