"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import
this module, and only as the checker.  The product (``pecanpy_amd``) never imports it.

Each wrapper names the reference code its C body restates (see ``oracle/pecan_oracle.c``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_BASE = None


class Stats(C.Structure):
    _fields_ = [
        ("overflow_reads", C.c_uint64),
        ("clamped_reads", C.c_uint64),
        ("total_steps", C.c_uint64),
    ]


def build(force=False):
    """Compile liboracle.so / libcpu_baseline.so with gcc (no GPU needed)."""
    targets = ["liboracle.so"]
    if os.path.exists(os.path.join(_HERE, "cpu_baseline.c")):
        targets.append("libcpu_baseline.so")
    if os.path.exists(os.path.join(_HERE, "sgns_ref.c")):
        targets.append("libsgns_ref.so")
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE] + targets, stdout=subprocess.DEVNULL)


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_sparse_probs.restype = C.c_uint32
        _LIB.orc_dense_probs.restype = C.c_uint32
    return _LIB


def _csr(indptr, indices, data):
    return (
        np.ascontiguousarray(indptr, dtype=np.uint32),
        np.ascontiguousarray(indices, dtype=np.uint32),
        np.ascontiguousarray(data, dtype=np.float32),
    )


def _thr(thr):
    return None if thr is None else np.ascontiguousarray(thr, dtype=np.float32)


def sparse_probs(indptr, indices, data, p, q, cur, prev=None, thr=None):
    """get_normalized_probs / get_extended_normalized_probs (sparse_rw.py:51-130)."""
    indptr, indices, data = _csr(indptr, indices, data)
    thr = _thr(thr)
    d = int(indptr[cur + 1] - indptr[cur])
    out = np.zeros(d, dtype=np.float32)
    lib().orc_sparse_probs(
        _ptr(indptr, C.c_uint32), _ptr(indices, C.c_uint32), _ptr(data, C.c_float),
        C.c_double(p), C.c_double(q), C.c_uint32(cur), C.c_int(prev is not None),
        C.c_uint32(0 if prev is None else prev), _ptr(thr, C.c_float), _ptr(out, C.c_float),
    )
    return out


def walks_sparse_otf(indptr, indices, data, p, q, starts, walk_length, seed, thr=None,
                     stream_skip=0, return_stats=False):
    """SparseOTF: Base._random_walks + SparseOTF.move_forward (pecanpy.py:164-210, 543-559)."""
    indptr, indices, data = _csr(indptr, indices, data)
    thr = _thr(thr)
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    out = np.zeros((starts.size, walk_length + 2), dtype=np.uint32)
    st = Stats()
    lib().orc_walks_sparse_otf(
        _ptr(indptr, C.c_uint32), _ptr(indices, C.c_uint32), _ptr(data, C.c_float),
        C.c_uint32(indptr.size - 1), C.c_double(p), C.c_double(q), _ptr(thr, C.c_float),
        _ptr(starts, C.c_uint32), C.c_uint64(starts.size), C.c_uint32(walk_length),
        C.c_uint32(seed), C.c_uint64(stream_skip), _ptr(out, C.c_uint32), C.byref(st),
    )
    return (out, st) if return_stats else out


def _dense(data, nonzero=None):
    data = np.ascontiguousarray(data, dtype=np.float64)
    if nonzero is None:
        nonzero = data != 0
    nonzero = np.ascontiguousarray(nonzero, dtype=np.uint8)
    return data, nonzero


def dense_probs(data, p, q, cur, prev=None, thr=None, nonzero=None):
    """DenseRWGraph.get_normalized_probs / get_extended_normalized_probs (dense_rw.py:34-118)."""
    data, nonzero = _dense(data, nonzero)
    thr = _thr(thr)
    n = data.shape[0]
    pr = np.zeros(n, dtype=np.float64)
    cols = np.zeros(n, dtype=np.uint32)
    d = lib().orc_dense_probs(
        _ptr(data, C.c_double), _ptr(nonzero, C.c_uint8), C.c_uint32(n), C.c_double(p),
        C.c_double(q), C.c_uint32(cur), C.c_int(prev is not None),
        C.c_uint32(0 if prev is None else prev), _ptr(thr, C.c_float), _ptr(pr, C.c_double),
        _ptr(cols, C.c_uint32),
    )
    return pr[:d].copy(), cols[:d].copy()


def walks_dense_otf(data, p, q, starts, walk_length, seed, thr=None, nonzero=None,
                    stream_skip=0, return_stats=False):
    """DenseOTF: pecanpy.py:597-612 + dense_rw.py."""
    data, nonzero = _dense(data, nonzero)
    thr = _thr(thr)
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    out = np.zeros((starts.size, walk_length + 2), dtype=np.uint32)
    st = Stats()
    lib().orc_walks_dense_otf(
        _ptr(data, C.c_double), _ptr(nonzero, C.c_uint8), C.c_uint32(data.shape[0]),
        C.c_double(p), C.c_double(q), _ptr(thr, C.c_float), _ptr(starts, C.c_uint32),
        C.c_uint64(starts.size), C.c_uint32(walk_length), C.c_uint32(seed),
        C.c_uint64(stream_skip), _ptr(out, C.c_uint32), C.byref(st),
    )
    return (out, st) if return_stats else out


def pack_adjacency(nonzero):
    """bool[N, N] -> uint64[N, ceil(N / 64)], bit x of word x // 64 = column x (little-endian bit order)."""
    nz = np.ascontiguousarray(nonzero, dtype=bool)
    n = nz.shape[0]
    wpr = (n + 63) // 64
    pad = np.zeros((n, wpr * 64), dtype=np.uint8)
    pad[:, :n] = nz
    return np.packbits(pad, axis=1, bitorder="little").view(np.uint64).reshape(n, wpr)


def walks_dense_otf_bits(bits, n, p, q, starts, walk_length, seed, stream_skip=0, return_stats=False):
    """DenseOTF on an UNWEIGHTED dense graph held as bit-packed adjacency rows (pecanpy.py:597-612 +
    dense_rw.py:34-72 with every stored value 1.0): the BASELINE C4 shape without the 80 GB float64 matrix."""
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    assert bits.ndim == 2 and bits.shape[0] == n and bits.shape[1] * 64 >= n
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    out = np.zeros((starts.size, walk_length + 2), dtype=np.uint32)
    st = Stats()
    lib().orc_walks_dense_otf_bits(
        _ptr(bits, C.c_uint64), C.c_uint32(n), C.c_uint32(bits.shape[1]), C.c_double(p), C.c_double(q),
        _ptr(starts, C.c_uint32), C.c_uint64(starts.size), C.c_uint32(walk_length), C.c_uint32(seed),
        C.c_uint64(stream_skip), _ptr(out, C.c_uint32), C.byref(st),
    )
    return (out, st) if return_stats else out


def precomp_tables(indptr, indices, data, p, q, thr=None):
    """PreComp.preprocess_transition_probs (pecanpy.py:442-507)."""
    indptr, indices, data = _csr(indptr, indices, data)
    thr = _thr(thr)
    deg = (indptr[1:] - indptr[:-1]).astype(np.uint64)
    alias_indptr = np.zeros(indptr.size, dtype=np.uint64)
    alias_indptr[1:] = np.cumsum(deg * deg)
    n_alias = int(alias_indptr[-1])
    alias_j = np.zeros(n_alias, dtype=np.uint32)
    alias_q = np.zeros(n_alias, dtype=np.float32)
    lib().orc_precomp_tables(
        _ptr(indptr, C.c_uint32), _ptr(indices, C.c_uint32), _ptr(data, C.c_float),
        C.c_uint32(indptr.size - 1), C.c_double(p), C.c_double(q), _ptr(thr, C.c_float),
        _ptr(alias_indptr, C.c_uint64), _ptr(alias_j, C.c_uint32), _ptr(alias_q, C.c_float),
    )
    return alias_indptr, alias_j, alias_q


def walks_precomp(indptr, indices, data, p, q, starts, walk_length, seed, thr=None, tables=None):
    """PreComp.move_forward (pecanpy.py:409-438)."""
    indptr, indices, data = _csr(indptr, indices, data)
    if tables is None:
        tables = precomp_tables(indptr, indices, data, p, q, thr)
    alias_indptr, alias_j, alias_q = tables
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    out = np.zeros((starts.size, walk_length + 2), dtype=np.uint32)
    st = Stats()
    lib().orc_walks_precomp(
        _ptr(indptr, C.c_uint32), _ptr(indices, C.c_uint32), _ptr(data, C.c_float),
        C.c_uint32(indptr.size - 1), C.c_double(p), C.c_double(q),
        _ptr(alias_indptr, C.c_uint64), _ptr(alias_j, C.c_uint32), _ptr(alias_q, C.c_float),
        _ptr(starts, C.c_uint32), C.c_uint64(starts.size), C.c_uint32(walk_length),
        C.c_uint32(seed), _ptr(out, C.c_uint32), C.byref(st),
    )
    return out


def first_order_tables(indptr, data):
    """PreCompFirstOrder.preprocess_transition_probs (pecanpy.py:336-361)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
    data = np.ascontiguousarray(data, dtype=np.float32)
    alias_j = np.zeros(data.size, dtype=np.uint32)
    alias_q = np.zeros(data.size, dtype=np.float32)
    lib().orc_first_order_tables(
        _ptr(indptr, C.c_uint32), _ptr(data, C.c_float), C.c_uint32(indptr.size - 1),
        _ptr(alias_j, C.c_uint32), _ptr(alias_q, C.c_float),
    )
    return alias_j, alias_q


def walks_first_order(indptr, indices, data, starts, walk_length, seed, precomp=False):
    """FirstOrderUnweighted (pecanpy.py:299-309) / PreCompFirstOrder (:319-334)."""
    indptr, indices, data = _csr(indptr, indices, data)
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    out = np.zeros((starts.size, walk_length + 2), dtype=np.uint32)
    aj = aq = None
    if precomp:
        aj, aq = first_order_tables(indptr, data)
    st = Stats()
    lib().orc_walks_first_order(
        _ptr(indptr, C.c_uint32), _ptr(indices, C.c_uint32), C.c_uint32(indptr.size - 1),
        C.c_int(1 if precomp else 0), _ptr(aj, C.c_uint32), _ptr(aq, C.c_float),
        _ptr(starts, C.c_uint32), C.c_uint64(starts.size), C.c_uint32(walk_length),
        C.c_uint32(seed), _ptr(out, C.c_uint32), C.byref(st),
    )
    return out


def random_sample(seed, offset, n):
    """RandomState(seed).random_sample doubles #offset..."""
    out = np.zeros(n, dtype=np.float64)
    lib().orc_random_sample(C.c_uint32(seed), C.c_uint64(offset), C.c_uint64(n), _ptr(out, C.c_double))
    return out


def random_words(seed, offset, n):
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_random_words(C.c_uint32(seed), C.c_uint64(offset), C.c_uint64(n), _ptr(out, C.c_uint32))
    return out


def shuffled_starts(num_nodes, num_walks, seed):
    """Start array of Base.simulate_walks (pecanpy.py:135-141): NumPy legacy seed + shuffle."""
    nodes = np.arange(num_nodes, dtype=np.uint32)
    starts = np.concatenate([nodes] * num_walks)
    rs = np.random.RandomState(seed)
    rs.shuffle(starts)
    return starts


def baseline_lib():
    global _BASE
    if _BASE is None:
        path = os.path.join(_HERE, "libcpu_baseline.so")
        if not os.path.exists(path):
            build()
        _BASE = C.CDLL(path)
    return _BASE


def cpu_baseline_walks(indptr, indices, data, p, q, starts, walk_length, seed, n_threads=0,
                       faithful=True, want_walks=False):
    """OpenMP port of the Numba-parallel SparseOTF path (oracle/cpu_baseline.c).
    Returns (steps, walks-or-None).  n_threads=0 -> all cores."""
    indptr, indices, data = _csr(indptr, indices, data)
    starts = np.ascontiguousarray(starts, dtype=np.uint32)
    out = np.zeros((starts.size, walk_length + 2), dtype=np.uint32) if want_walks else None
    steps = C.c_uint64(0)
    baseline_lib().cpub_walks_sparse(
        _ptr(indptr, C.c_uint32), _ptr(indices, C.c_uint32), _ptr(data, C.c_float),
        C.c_uint32(indptr.size - 1), C.c_double(p), C.c_double(q), _ptr(starts, C.c_uint32),
        C.c_uint64(starts.size), C.c_uint32(walk_length), C.c_uint32(seed), C.c_int(n_threads),
        C.c_int(1 if faithful else 0), _ptr(out, C.c_uint32), C.byref(steps))
    return int(steps.value), out


def cpu_baseline_max_threads():
    return int(baseline_lib().cpub_max_threads())


_SGNS = None


def sgns_train(walk_matrix, num_nodes, dim=128, window=10, epochs=1, negative=5, alpha=0.025, min_alpha=1e-4,
               sample=1e-3, seed=0):
    """Sequential skip-gram with negative sampling over a walk matrix (oracle/sgns_ref.c: word2vec.c / gensim sg=1,
    the model of Base.embed, src/pecanpy/pecanpy.py:276-290).  Returns ``(vectors float32[num_nodes, dim], mean loss of
    the last epoch)``."""
    global _SGNS
    if _SGNS is None:
        path = os.path.join(_HERE, "libsgns_ref.so")
        if not os.path.exists(path):
            build()
        _SGNS = C.CDLL(path)
    mat = np.ascontiguousarray(walk_matrix, dtype=np.uint32)
    out = np.zeros((int(num_nodes), int(dim)), dtype=np.float32)
    loss = C.c_double(0)
    rc = _SGNS.sgns_ref_train(mat.ctypes.data_as(C.c_void_p), C.c_uint64(mat.shape[0]), C.c_uint32(mat.shape[1] - 2),
                              C.c_uint32(int(num_nodes)), C.c_uint32(int(dim)), C.c_uint32(int(window)), C.c_uint32(int(negative)),
                              C.c_uint32(int(epochs)), C.c_float(alpha), C.c_float(min_alpha), C.c_float(sample),
                              C.c_uint32(int(seed) & 0xFFFFFFFF), out.ctypes.data_as(C.c_void_p), C.byref(loss))
    if rc != 0:
        raise ValueError("sgns_ref_train: malformed walk matrix")
    return out, float(loss.value)
