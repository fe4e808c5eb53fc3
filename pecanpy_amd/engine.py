"""Device-side walk engine: Python face of the C ABI (include/pecanpy_amd.h).

``WalkEngine`` owns one device-resident graph handle (``pw_csr_create`` / ``pw_dense_create``) and
exposes the walk operator that replaces the reference's ``Base._random_walks`` + ``has_nbrs`` +
``move_forward`` (reference src/pecanpy/pecanpy.py:164-210).  NumPy arrays go through
``pw_simulate`` (host pointers); torch CUDA tensors go through ``pw_simulate_device`` (nothing
crosses PCIe).  ``simulate_sharded`` is the multi-GPU path: one process per GPU, the shuffled job
array split into contiguous ranges, graph replicated, one gather of the walk shards at the end.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MODE_IDS, PwError, PwStats

__all__ = ["WalkEngine", "MultiWalkEngine", "visible_devices", "shard_bounds", "auto_rank0_share", "tapered_bounds", "PwError"]


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def shard_bounds(n_jobs, world_size, rank0_share=None):
    """Contiguous job ranges [lo, hi) per rank.

    ``rank0_share`` (default 1.0 = uniform): rank 0's shard as a fraction of a uniform one.  Rank 0 is where the walk
    matrix is assembled -- it also writes the rows nobody sends and scatters what arrives -- so giving it fewer jobs to
    walk takes that work off the pass's critical path (``auto_rank0_share``); the other ranks share the rest evenly."""
    if rank0_share is None or world_size <= 1 or rank0_share == 1.0:
        return [((r * n_jobs) // world_size, ((r + 1) * n_jobs) // world_size) for r in range(world_size)]
    share = min(max(float(rank0_share), 0.0), float(world_size))
    n0 = min(n_jobs, int(round(share * n_jobs / world_size)))
    rest = n_jobs - n0
    cuts = [0, n0] + [n0 + ((r * rest) // (world_size - 1)) for r in range(1, world_size)]
    return [(cuts[r], cuts[r + 1]) for r in range(world_size)]


def auto_rank0_share(world_size, gather=True):
    """Rank 0's share of a uniform shard when it assembles the matrix (model of DESIGN.md section 6: at 8 GPUs its
    prefill of the other shards' isolated rows, the receive kernels and the scatter of 7/8 of the rows cost about half
    of a 1/8 shard's walk time; nothing extra at 1 GPU, proportionally less in between)."""
    if not gather or world_size <= 1:
        return 1.0
    return max(0.4, 1.0 - 0.5 * (min(world_size, 8) - 1) / 7.0)


def tapered_bounds(n_jobs, n_chunks):
    """Contiguous ranges [lo, hi) of decreasing size (weights n_chunks, n_chunks - 1, ..., 1): a shard walked in such
    chunks, each travelling to rank 0 while the next is walked, leaves only its SMALLEST chunk's transfer exposed at the
    end of the pass (4 chunks: 10 % of the shard instead of 25 %)."""
    n_chunks = max(1, int(n_chunks))
    total = n_chunks * (n_chunks + 1) // 2
    cuts = [0]
    acc = 0
    for c in range(n_chunks):
        acc += n_chunks - c
        cuts.append((acc * n_jobs) // total)
    return [(cuts[c], cuts[c + 1]) for c in range(n_chunks)]


class WalkEngine:
    def __init__(self, handle, lib, kind, n_nodes, device):
        self._h = handle
        self._lib = lib
        self.kind = kind
        self.n_nodes = n_nodes
        self.device = device
        self.last_stats = None
        self._max_degree = n_nodes   # upper bound; from_csr / from_dense narrow it
        self._nnz = 0                # CSR entries (from_csr)

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def from_csr(cls, indptr, indices, data=None, device=0):
        lib = _lib.load()
        indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        if data is not None:
            data = np.ascontiguousarray(data, dtype=np.float32)
        h = C.c_void_p()
        _lib.check(lib.pw_csr_create(_np_ptr(indptr), _np_ptr(indices), _np_ptr(data),
                                     indptr.size - 1, indices.size, int(device), C.byref(h)))
        eng = cls(h, lib, "csr", indptr.size - 1, int(device))
        eng._max_degree = int(np.diff(indptr.astype(np.int64)).max()) if indptr.size > 1 else 0
        eng._nnz = int(indices.size)
        return eng

    @classmethod
    def from_dense(cls, data, device=0):
        lib = _lib.load()
        data = np.ascontiguousarray(data, dtype=np.float64)
        if data.ndim != 2 or data.shape[0] != data.shape[1]:
            raise ValueError("dense adjacency must be a square matrix")
        h = C.c_void_p()
        _lib.check(lib.pw_dense_create(_np_ptr(data), data.shape[0], int(device), C.byref(h)))
        return cls(h, lib, "dense", data.shape[0], int(device))

    @classmethod
    def from_dense_bits(cls, bits, n_nodes, device=0):
        """Unweighted dense graph from packed adjacency rows: ``bits`` is ``uint64[n, ceil(n/64)]`` as a
        NumPy array (host) or an int64 torch CUDA tensor (device, same bit pattern)."""
        lib = _lib.load()
        wpr = (int(n_nodes) + 63) // 64
        h = C.c_void_p()
        if isinstance(bits, np.ndarray):
            bits = np.ascontiguousarray(bits, dtype=np.uint64)
            if bits.size != int(n_nodes) * wpr:
                raise ValueError("bits must hold n * ceil(n/64) words")
            _lib.check(lib.pw_dense_create_bits(_np_ptr(bits), int(n_nodes), 0, int(device), C.byref(h)))
        else:  # torch CUDA tensor
            if not bits.is_cuda or not bits.is_contiguous() or bits.numel() != int(n_nodes) * wpr:
                raise ValueError("bits must be a contiguous CUDA tensor of n * ceil(n/64) 64-bit words")
            import torch

            torch.cuda.current_stream(bits.device).synchronize()
            _lib.check(lib.pw_dense_create_bits(C.c_void_p(bits.data_ptr()), int(n_nodes), 1, int(device), C.byref(h)))
        return cls(h, lib, "dense", int(n_nodes), int(device))

    def set_thresholds(self, thr):
        thr = np.ascontiguousarray(thr, dtype=np.float32)
        if thr.size != self.n_nodes:
            raise ValueError("threshold array must have one entry per node")
        _lib.check(self._lib.pw_graph_set_thresholds(self._h, _np_ptr(thr)))

    def index_info(self):
        """Device time (ms) and bytes of the per-graph index built at creation, and the number of entries of the
        lane kernel's common-neighbour lists (0 when that index was not built)."""
        ms, nbytes, entries = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        _lib.check(self._lib.pw_graph_index_info(self._h, C.byref(ms), C.byref(nbytes), C.byref(entries)))
        return {"build_ms": float(ms.value), "index_bytes": int(nbytes.value), "lane_list_entries": int(entries.value)}

    def lane_index(self):
        """Test hook: ``(n_in, rev_pos, offsets, entries)`` of the lane index (see ``pw_lane_index_export``)."""
        n = max(self._nnz, 1)   # (dense handles / no lane index: the library's PW_ERR_UNSUPPORTED surfaces below)
        n_in = np.zeros(n, dtype=np.uint32)
        rev = np.zeros(n, dtype=np.uint32)
        _lib.check(self._lib.pw_lane_index_export(self._h, _np_ptr(n_in), _np_ptr(rev), None))   # counts first
        n_in, rev = n_in[: self._nnz], rev[: self._nnz]
        total = int(n_in.sum(dtype=np.int64))
        if total != self.index_info()["lane_list_entries"]:
            raise _lib.PwError("lane index: the per-entry counts do not add up to the number of list entries")
        entries = np.zeros(max(total, 1), dtype=np.uint32)
        _lib.check(self._lib.pw_lane_index_export(self._h, None, None, _np_ptr(entries)))
        off = np.concatenate([[0], np.cumsum(n_in, dtype=np.int64)])
        return n_in, rev, off, entries[:total]

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.pw_graph_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    # ---- the walk operator ----------------------------------------------------------------
    def simulate(self, mode, p, q, extend, starts, walk_length, seed=None, stream_skip=0):
        """Host-buffer variant: returns ``uint32[n_jobs, walk_length + 2]`` (NumPy)."""
        starts = np.ascontiguousarray(starts, dtype=np.uint32)
        out = np.empty((starts.size, walk_length + 2), dtype=np.uint32)
        st = PwStats()
        _lib.check(self._lib.pw_simulate(
            self._h, MODE_IDS[mode], float(p), float(q), int(bool(extend)), _np_ptr(starts),
            starts.size, int(walk_length), int(seed is not None), int(seed or 0) & 0xFFFFFFFF,
            int(stream_skip), _np_ptr(out), C.byref(st)))
        self.last_stats = st.as_dict()
        return out

    def simulate_device(self, mode, p, q, extend, d_starts, walk_length, seed=None, stream_skip=0,
                        out=None):
        """Device-buffer variant on torch CUDA tensors (int32 storage viewed as uint32)."""
        import torch

        if not d_starts.is_cuda or d_starts.dtype != torch.int32 or not d_starts.is_contiguous():
            raise ValueError("d_starts must be a contiguous int32 CUDA tensor")
        n = d_starts.numel()
        if out is None:
            out = torch.empty((n, walk_length + 2), dtype=torch.int32, device=d_starts.device)
        torch.cuda.current_stream(d_starts.device).synchronize()  # inputs were produced on torch's stream
        st = PwStats()
        _lib.check(self._lib.pw_simulate_device(
            self._h, MODE_IDS[mode], float(p), float(q), int(bool(extend)),
            C.c_void_p(d_starts.data_ptr()), n, int(walk_length), int(seed is not None),
            int(seed or 0) & 0xFFFFFFFF, int(stream_skip), C.c_void_p(out.data_ptr()), C.byref(st)))
        self.last_stats = st.as_dict()
        return out

    # ---- single transitions (the reference's move_forward / get_normalized_probs callbacks) ---------------
    def step(self, mode, p, q, extend, cur, prev=None, r=None):
        """``move_forward(cur, prev)`` on the device with the uniform draw ``r`` (default: ``np.random.random()``,
        as the reference draws it); returns the next vertex index."""
        if r is None:
            r = np.random.random()
        nxt, pos = C.c_uint32(0), C.c_uint32(0)
        _lib.check(self._lib.pw_step(self._h, MODE_IDS[mode], float(p), float(q), int(bool(extend)), int(cur),
                                     int(prev is not None), int(prev or 0), float(r), C.byref(nxt), C.byref(pos)))
        return int(nxt.value)

    def probs(self, mode, p, q, extend, cur, prev=None):
        """``get_normalized_probs(cur, prev)`` computed by the walk kernels' own step code: float32 (CSR) or
        float64 (dense) vector over ``cur``'s neighbours."""
        dt = np.float32 if self.kind == "csr" else np.float64
        buf = np.zeros(self.max_degree() + 1, dtype=dt)
        n = C.c_uint32(0)
        _lib.check(self._lib.pw_probs(self._h, MODE_IDS[mode], float(p), float(q), int(bool(extend)), int(cur),
                                      int(prev is not None), int(prev or 0), _np_ptr(buf), C.byref(n)))
        return buf[: int(n.value)].copy()

    def max_degree(self):
        return int(self._max_degree)

    def precomp_build(self, p, q, extend, first_order):
        """Alias tables on the device (PreComp / PreCompFirstOrder preprocessing)."""
        _lib.check(self._lib.pw_precomp_build(self._h, float(p), float(q), int(bool(extend)),
                                              int(bool(first_order))))

    def precomp_export(self, first_order):
        """Host copies ``(alias_indptr, alias_j, alias_q)`` of the tables built last."""
        n = C.c_uint64(0)
        _lib.check(self._lib.pw_precomp_export(self._h, None, None, None, C.byref(n)))
        alias_indptr = np.zeros(self.n_nodes + 1, dtype=np.uint64)
        alias_j = np.zeros(int(n.value), dtype=np.uint32)
        alias_q = np.zeros(int(n.value), dtype=np.float32)
        _lib.check(self._lib.pw_precomp_export(self._h, _np_ptr(alias_indptr), _np_ptr(alias_j),
                                               _np_ptr(alias_q), C.byref(n)))
        return alias_indptr, alias_j, alias_q

    def stream_sample(self, seed, offset, n):
        """Test hook: doubles ``#offset .. #offset + n`` of ``RandomState(seed).random_sample`` as the device's jump-ahead
        tree and expansion kernels produce them (``pw_stream_sample_device``)."""
        out = np.zeros(int(n), dtype=np.float64)
        _lib.check(self._lib.pw_stream_sample_device(self._h, int(seed) & 0xFFFFFFFF, int(offset), int(n), _np_ptr(out)))
        return out

    def stream_hold(self, seed, stream_skip, n_draws):
        """Expand the draws ``[stream_skip, stream_skip + n_draws)`` of ``seed``'s stream once; ``simulate_device`` calls inside
        that range (same seed) use them in place until ``stream_release`` -- one jump-ahead tree for a shard walked in chunks."""
        _lib.check(self._lib.pw_stream_hold(self._h, int(seed) & 0xFFFFFFFF, int(stream_skip), int(n_draws)))

    def stream_release(self):
        _lib.check(self._lib.pw_stream_release(self._h))

    def count_stream_draws(self, starts, walk_length):
        starts = np.ascontiguousarray(starts, dtype=np.uint32)
        n = C.c_uint64(0)
        _lib.check(self._lib.pw_count_stream_draws(self._h, _np_ptr(starts), starts.size,
                                                   int(walk_length), C.byref(n)))
        return int(n.value)


def visible_devices(spec=None):
    """Device list from a spec: ``None`` / ``"all"`` = every visible GPU, an int = that many (0 = all), a bit mask given as
    ``"mask:0x0f"``, or a comma list ``"0,1,2"`` (a device may be named twice: every entry is a replica)."""
    lib = _lib.load()
    n = int(lib.pw_device_count())
    if spec is None or spec == "all" or spec == 0:
        return list(range(n))
    if isinstance(spec, int):
        return list(range(min(spec, n)))
    if isinstance(spec, str) and spec.startswith("mask:"):
        buf = (C.c_int * 64)()
        k = lib.pw_device_mask_to_list(int(spec[5:], 0), buf, 64)
        if k < 0:
            _lib.check(k)
        return [int(buf[i]) for i in range(k)]
    if isinstance(spec, str):
        return [int(t) for t in spec.split(",") if t.strip() != ""]
    return [int(d) for d in spec]


class MultiWalkEngine:
    """Replicas of one graph on several GPUs, driven from THIS process by one host thread per device inside one C-ABI call
    (``pw_csr_create_multi`` / ``pw_simulate_multi``): what the reference's single process with its Numba thread pool is to
    the CPU (src/pecanpy/cli.py:340-351, pecanpy.py:165-189).  The index is built once and copied device to device.  The
    walk matrix equals a one-device run bit for bit (one random stream, shards addressed by the draws of the earlier ones)."""

    def __init__(self, engines):
        self.engines = list(engines)
        self._lib = self.engines[0]._lib
        self.kind = self.engines[0].kind
        self.n_nodes = self.engines[0].n_nodes
        self.devices = [e.device for e in self.engines]
        self.last_stats = None

    @classmethod
    def from_csr(cls, indptr, indices, data=None, devices=None):
        lib = _lib.load()
        devices = visible_devices(devices)
        if not devices:
            raise PwError("no HIP device visible (libpecanpy_amd needs a GPU; there is no CPU fallback)")
        indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        if data is not None:
            data = np.ascontiguousarray(data, dtype=np.float32)
        dev = (C.c_int * len(devices))(*devices)
        hs = (C.c_void_p * len(devices))()
        _lib.check(lib.pw_csr_create_multi(_np_ptr(indptr), _np_ptr(indices), _np_ptr(data), indptr.size - 1, indices.size,
                                           dev, len(devices), hs))
        engines = []
        for h, d in zip(hs, devices):
            eng = WalkEngine(C.c_void_p(h), lib, "csr", indptr.size - 1, int(d))
            eng._max_degree = int(np.diff(indptr.astype(np.int64)).max()) if indptr.size > 1 else 0
            eng._nnz = int(indices.size)
            engines.append(eng)
        return cls(engines)

    @classmethod
    def from_engine(cls, engine, devices):
        """Replicate an existing one-device engine onto ``devices`` (entries equal to its own device become replicas too)."""
        lib = engine._lib
        engines = [engine]
        for d in list(devices)[1:]:
            h = C.c_void_p()
            _lib.check(lib.pw_graph_replicate(engine._h, int(d), C.byref(h)))
            rep = WalkEngine(h, lib, engine.kind, engine.n_nodes, int(d))
            rep._max_degree, rep._nnz = engine._max_degree, engine._nnz
            engines.append(rep)
        return cls(engines)

    def set_thresholds(self, thr):
        for e in self.engines:
            e.set_thresholds(thr)

    def index_info(self):
        return self.engines[0].index_info()

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []

    def _handles(self):
        return (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])

    def simulate(self, mode, p, q, extend, starts, walk_length, seed=None, stream_skip=0):
        """Host-buffer variant: ``uint32[n_jobs, walk_length + 2]`` (NumPy), every device writing its own rows."""
        starts = np.ascontiguousarray(starts, dtype=np.uint32)
        out = np.empty((starts.size, walk_length + 2), dtype=np.uint32)
        st = PwStats()
        _lib.check(self._lib.pw_simulate_multi(
            self._handles(), len(self.engines), MODE_IDS[mode], float(p), float(q), int(bool(extend)), _np_ptr(starts),
            starts.size, int(walk_length), int(seed is not None), int(seed or 0) & 0xFFFFFFFF, int(stream_skip),
            _np_ptr(out), 0, C.byref(st)))
        self.last_stats = st.as_dict()
        return out

    def simulate_to_device(self, mode, p, q, extend, starts, walk_length, seed=None, stream_skip=0, out=None):
        """The matrix assembled in the memory of the FIRST device (torch int32 tensor): the other devices' rows arrive by
        peer copies over xGMI ("gathered once at the end", BASELINE north star) -- without RCCL, from one process."""
        import torch

        starts = np.ascontiguousarray(starts, dtype=np.uint32)
        dev = torch.device("cuda", self.devices[0])
        if out is None:
            out = torch.empty((starts.size, walk_length + 2), dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        st = PwStats()
        _lib.check(self._lib.pw_simulate_multi(
            self._handles(), len(self.engines), MODE_IDS[mode], float(p), float(q), int(bool(extend)), _np_ptr(starts),
            starts.size, int(walk_length), int(seed is not None), int(seed or 0) & 0xFFFFFFFF, int(stream_skip),
            C.c_void_p(out.data_ptr()), 1, C.byref(st)))
        self.last_stats = st.as_dict()
        return out
