"""node2vec walk strategies: the reference's mode classes on the MI355X walk engine.

Drop-in mirror of ``pecanpy.pecanpy`` (reference src/pecanpy/pecanpy.py): same class names,
constructor signature ``(p, q, workers, verbose, extend, gamma, random_state)`` (:83-101) and
methods ``simulate_walks`` (:116-162), ``preprocess_transition_probs`` (:231-238), ``embed``
(:240-290).  What differs is where the walks run: ``Base._random_walks`` (the Numba ``prange``
kernel, :164-210) and the per-mode ``move_forward`` closures are replaced by one call into
``libpecanpy_amd.so`` (HIP kernels for gfx950).  Seeded runs reproduce the reference's
*single-thread* walks bit for bit, independent of how many GPUs execute them.
"""
import numpy as np

from .engine import WalkEngine
from .graph import BaseGraph, DenseGraph, SparseGraph
from .wrappers import Timer

__all__ = ["Base", "FirstOrderUnweighted", "PreCompFirstOrder", "PreComp", "SparseOTF", "DenseOTF",
           "WalkCorpus"]


class WalkCorpus:
    """Re-iterable view of a walk index matrix as sentences of node IDs.

    ``simulate_walks`` has to materialise ``List[List[str]]`` (reference API, pecanpy.py:160); at
    RMAT-22 scale that is ~3.4 G Python objects.  ``WalkCorpus`` keeps the ``uint32[n_jobs, L+2]``
    matrix the GPU produced and maps rows to ID lists lazily, chunk by chunk, so gensim's
    ``Word2Vec(corpus)`` (which iterates the corpus once per epoch) can stream it
    (SURVEY.md section 8(f) rank 1).
    """

    def __init__(self, walk_idx_mat, node_ids, chunk=65536):
        self.matrix = walk_idx_mat
        self._ids = np.asarray(node_ids, dtype=object)
        self._chunk = int(chunk)

    def __len__(self):
        return int(self.matrix.shape[0])

    def __iter__(self):
        mat, ids = self.matrix, self._ids
        last = mat.shape[1] - 1
        for lo in range(0, mat.shape[0], self._chunk):
            block = mat[lo:lo + self._chunk]
            names = ids[block[:, :last]]           # vectorised index -> ID for the whole chunk
            for row, n in zip(names, block[:, last]):
                yield row[:n].tolist()

    def __getitem__(self, i):
        row = self.matrix[i]
        return self._ids[row[: row[-1]]].tolist()


class Base(BaseGraph):
    """Skeleton shared by all walk modes (reference ``Base``, pecanpy.py:27-290).

    Args mirror the reference: ``p`` return parameter, ``q`` in-out parameter, ``workers`` (used
    for Word2Vec only), ``verbose``, ``extend`` (node2vec+), ``gamma`` (noise-threshold factor),
    ``random_state`` (seed; ``None`` = entropy from the OS).
    """

    _mode = None  # name understood by the C ABI (pw_mode)

    def __init__(self, p=1, q=1, workers=1, verbose=False, extend=False, gamma=0, random_state=None):
        super().__init__()
        self.p = p
        self.q = q
        self.workers = workers
        self.verbose = verbose
        self.extend = extend
        self.gamma = gamma
        self.random_state = random_state
        self._preprocessed = False
        self._engine = None
        self._multi = None       # replicas on the other GPUs of this process (_multi_engine)
        self._engine_key = None
        self._thr_key = None
        self._run_seed = None
        self.device = None  # GPU index; None -> LOCAL_RANK / 0
        self.last_stats = None
        # the library's one-time start-up (~140 ms for the first stream it creates) runs on a helper thread beside what comes
        # next in the reference's flow -- reading the graph (cli.py:328-337) -- instead of in front of the first walk
        import os

        if os.environ.get("PECANPY_AMD_NO_WARMUP") is None:
            from . import _lib

            _lib.warmup_async(int(os.environ.get("LOCAL_RANK", "0")))

    # ---- engine plumbing -------------------------------------------------------------------
    def _device_index(self):
        if self.device is not None:
            return int(self.device)
        import os

        return int(os.environ.get("LOCAL_RANK", "0"))

    def _graph_key(self):
        raise NotImplementedError

    def _make_engine(self, device):
        raise NotImplementedError

    def _get_engine(self):
        key = (self._graph_key(), self._device_index())
        if self._engine is None or self._engine_key != key:
            if self._multi is not None:
                for rep in self._multi.engines[1:]:
                    rep.close()
                self._multi = None
            if self._engine is not None:
                self._engine.close()
            self._engine = self._make_engine(self._device_index())
            self._engine_key = key
            self._thr_key = None
        if self.extend and self._thr_key != (self.gamma,):   # gamma may change between calls
            thr = self.get_noise_thresholds()
            self._engine.set_thresholds(thr)
            if self._multi is not None:
                for rep in self._multi.engines[1:]:
                    rep.set_thresholds(thr)
            self._thr_key = (self.gamma,)
        return self._engine

    #: nominal steps (jobs x walk_length) from which a call spreads over every visible GPU when PECANPY_AMD_DEVICES is unset
    MULTI_DEVICE_MIN_STEPS = 200_000_000

    def _multi_engine(self, n_jobs, walk_length):
        """In-process multi-GPU (round 6): replicas of the engine's graph on the other visible devices, driven by one host
        thread per device inside ONE C-ABI call (``pw_simulate_multi``) -- what the reference's single process with its Numba
        thread pool is on the CPU (cli.py:340-351).  ``PECANPY_AMD_DEVICES`` = ``all`` | a count | ``0,1,2`` | ``mask:0x0f``;
        unset: every visible GPU once the call is large enough to pay for the replication.  ``None``: one device."""
        import os

        if self._mode not in ("SparseOTF", "DenseOTF") or self.device is not None or "LOCAL_RANK" in os.environ:
            return None
        if self._engine is None or self._engine.kind != "csr":
            return None
        spec = os.environ.get("PECANPY_AMD_DEVICES")
        from .engine import MultiWalkEngine, visible_devices

        if spec is None:
            if n_jobs * walk_length < self.MULTI_DEVICE_MIN_STEPS:
                return None
            devices = visible_devices(None)
        else:
            devices = visible_devices(int(spec) if spec.strip().isdigit() else spec.strip())
        if len(devices) < 2:
            return None
        if devices[0] != self._engine.device:
            devices = [self._engine.device] + [d for d in devices if d != self._engine.device]
        if self._multi is None or self._multi.devices != devices:
            if self._multi is not None:
                for rep in self._multi.engines[1:]:
                    rep.close()
            self._multi = MultiWalkEngine.from_engine(self._engine, devices)
        return self._multi

    # ---- reference API ---------------------------------------------------------------------
    def _map_walk(self, walk_idx_ary):
        """Index row -> ID list; the last cell is the effective length (pecanpy.py:103-114)."""
        n = int(walk_idx_ary[-1])
        ids = self.nodes
        return [ids[i] for i in walk_idx_ary[:n].tolist()]

    @staticmethod
    def _dist():
        """``torch.distributed`` when this process is one rank of a multi-rank job, else ``None`` (torch is only
        needed for the multi-GPU path)."""
        try:
            import torch.distributed as dist
        except ImportError:
            return None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist
        return None

    def _call_seed(self):
        """Seed of this ``simulate_walks`` call.  ``random_state=None`` means entropy from the OS (reference:
        ``np.random.seed(None)``); across the ranks of a multi-GPU job that entropy must be the same, or every rank
        would shuffle -- and then shard -- a different job array: rank 0 draws it and broadcasts it."""
        if self.random_state is not None:
            return self.random_state
        dist = self._dist()
        if dist is None:
            return None
        box = [int(np.random.SeedSequence().generate_state(1)[0])]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def _start_array(self, num_walks, seed=None):
        """Each node ``num_walks`` times, then NumPy's legacy seeded shuffle (pecanpy.py:135-141)."""
        nodes = np.arange(self.num_nodes, dtype=np.uint32)
        starts = np.concatenate([nodes] * num_walks)
        np.random.seed(self.random_state if seed is None else seed)
        np.random.shuffle(starts)
        return starts

    def simulate_walks_array(self, num_walks, walk_length, gather=True):
        """The walk index matrix ``uint32[n_jobs, walk_length + 2]`` (what ``_random_walks``
        returns in the reference); ``simulate_walks`` maps it to ID lists.  Under
        ``torch.distributed`` with ``gather=False`` every rank gets ``(rows, (lo, hi))``: its own
        slice [lo, hi) of that matrix, without the final collective."""
        self._preprocess_transition_probs()
        self._run_seed = self._call_seed()
        starts = self._start_array(num_walks, self._run_seed)
        return self._random_walks(starts, walk_length, gather=gather)

    def simulate_walks_corpus(self, num_walks, walk_length):
        """Like ``simulate_walks`` but returns a lazy, re-iterable :class:`WalkCorpus`."""
        return WalkCorpus(self.simulate_walks_array(num_walks, walk_length), self.nodes)

    def simulate_walks(self, num_walks, walk_length):
        """Generate ``num_walks`` walks from every node; returns ``List[List[str]]``."""
        mat = self.simulate_walks_array(num_walks, walk_length)
        return [self._map_walk(row) for row in mat]

    def _note_stats(self, stats):
        self.last_stats = stats
        if stats and stats.get("verify_mismatch", 0) > 0:
            import os
            import warnings

            if not os.environ.get("PECANPY_AMD_VERIFY_TIGHT"):
                warnings.warn(
                    f"{stats['verify_mismatch']} of {stats['verify_checked']} sampled interval decisions of the lane kernel disagreed "
                    "with the sequential float32 chain; the affected walks were generated again by the complete kernel.  This has "
                    "never been observed (DESIGN.md section 3) -- please report the graph and parameters.  "
                    "PECANPY_AMD_NO_LANES=1 avoids the lane kernel altogether.", RuntimeWarning, stacklevel=3)
        if stats and stats.get("stream_addressing") == 1:
            import warnings

            warnings.warn(
                "PECANPY_AMD_NOMINAL_STREAM is set: walks on this sink-heavy directed graph were generated with NOMINAL "
                "stream addressing (one fixed slot of walk_length draws per walk) after 32 re-addressing passes: "
                "reproducible under the seed, but not the reference's draw-for-draw assignment (unset it for the exact, "
                "block-wise repair; see DESIGN.md section 3)", RuntimeWarning, stacklevel=3)

    def _random_walks(self, starts, walk_length, gather=True):
        """GPU replacement of the reference's njit ``_random_walks`` (pecanpy.py:164-210)."""
        eng = self._get_engine()
        seed = self._run_seed if self.random_state is None else self.random_state
        if self._dist() is not None:
            if self._mode not in ("SparseOTF", "DenseOTF"):
                raise NotImplementedError(
                    f"{self._mode} draws a variable number of random words per step; its seeded "
                    "stream cannot be sharded across GPUs -- run it in a single process")
            return self._random_walks_sharded(eng, starts, walk_length, seed, gather)
        multi = self._multi_engine(starts.size, walk_length)
        if multi is not None:
            eng = multi
        mat = eng.simulate(self._mode, self.p, self.q, self.extend, starts, walk_length, seed=seed)
        self._note_stats(eng.last_stats)
        return mat

    def _random_walks_sharded(self, eng, starts, walk_length, seed, gather=True):
        import torch
        import torch.distributed as dist

        from .sharding import sharded_walk_matrix, to_uint32_numpy

        dev = torch.device("cuda", eng.device)
        host_comm = dist.get_backend() == "gloo"  # gloo moves CPU tensors; nccl (= RCCL) device tensors

        def run_shard(sl, skip):
            d_starts = torch.from_numpy(np.ascontiguousarray(sl).view(np.int32)).to(dev)
            out = eng.simulate_device(self._mode, self.p, self.q, self.extend, d_starts,
                                      walk_length, seed=seed, stream_skip=skip)
            steps = eng.last_stats["total_steps"] if d_starts.numel() else 0
            if d_starts.numel() and eng.last_stats["stream_addressing"] == 1:
                # nominal slots: the shard owns walk_length draws per walk with neighbours, used or not
                steps = eng.count_stream_draws(sl, walk_length)
            return (out.cpu() if host_comm else out), steps

        full = sharded_walk_matrix(run_shard, lambda sl: eng.count_stream_draws(sl, walk_length),
                                   starts, walk_length, gather=gather)
        self._note_stats(eng.last_stats)
        if not gather:
            rows, bounds = full
            return to_uint32_numpy(rows), bounds
        return to_uint32_numpy(full)

    def get_move_forward(self):
        """``move_forward(cur_idx, prev_idx=None) -> next_idx`` of the on-the-fly modes (pecanpy.py:522-561, 576-614),
        evaluated on the GPU by the walk kernels' own step code (``pw_step``); the uniform draw comes from
        ``np.random.random()`` as in the reference.  One kernel launch per call: for API compatibility and
        inspection -- ``simulate_walks`` is the throughput path."""
        if self._mode not in ("SparseOTF", "DenseOTF"):
            raise NotImplementedError(f"{self._mode}: single steps are provided for the on-the-fly modes")
        eng = self._get_engine()
        mode, p, q, extend = self._mode, self.p, self.q, self.extend

        def move_forward(cur_idx, prev_idx=None):
            return eng.step(mode, p, q, extend, cur_idx, prev_idx)

        return move_forward

    def setup_get_normalized_probs(self):
        """``(get_normalized_probs, noise_thresholds)`` as the reference returns them (pecanpy.py:212-229).  The
        callable keeps the reference's signature ``(data, indices, indptr, p, q, cur_idx, prev_idx=None,
        average_weight_ary=None)`` but computes on the GPU from the graph this object holds (``pw_probs``: the
        probabilities the walk kernels sample from, bit for bit)."""
        thr = self.get_noise_thresholds() if self.extend else None
        if self._mode == "DenseOTF":
            mode = "DenseOTF"
        else:
            mode = "SparseOTF"      # the alias modes precompute exactly these vectors (pecanpy.py:442-507)
        eng = self._get_engine()
        extend = self.extend

        def get_normalized_probs(data, indices, indptr, p, q, cur_idx, prev_idx=None, average_weight_ary=None):
            return eng.probs(mode, p, q, extend, cur_idx, prev_idx)

        return get_normalized_probs, thr

    def preprocess_transition_probs(self):
        """No-op for on-the-fly modes (pecanpy.py:231-233)."""

    def _preprocess_transition_probs(self):
        if not self._preprocessed:
            self.preprocess_transition_probs()
            self._preprocessed = True

    def embed(self, dim=128, num_walks=10, walk_length=80, window_size=10, epochs=1, verbose=False):
        """``simulate_walks`` + skip-gram (pecanpy.py:240-290): returns ``float32[num_nodes, dim]`` in node order.

        With gensim installed the reference's call is made (``Word2Vec(walks, sg=1, min_count=0, ...)``); without
        it the GPU trainer of this package runs on the walk matrix directly (``pecanpy_amd.embed.train_sgns``: same
        model and defaults, no string corpus)."""
        try:
            from gensim.models import Word2Vec
        except ImportError:
            Word2Vec = None
        if Word2Vec is None:
            from .embed import train_sgns

            mat = Timer("generate walks", verbose)(self.simulate_walks_array)(num_walks, walk_length)
            return Timer("train embeddings", verbose)(train_sgns)(
                mat, self.num_nodes, dim=dim, window=window_size, epochs=epochs, seed=self.random_state,
                device=self._device_index())
        walks = Timer("generate walks", verbose)(self.simulate_walks)(num_walks, walk_length)
        w2v = Timer("train embeddings", verbose)(Word2Vec)(
            walks, vector_size=dim, window=window_size, sg=1, min_count=0, workers=self.workers,
            epochs=epochs, seed=self.random_state)
        return w2v.wv[self.nodes]


class _SparseBase(Base, SparseGraph):
    """CSR-backed modes (reference ``SparseRWGraph`` mixin, rw/sparse_rw.py:9-35)."""

    def __init__(self, *args, **kwargs):
        Base.__init__(self, *args, **kwargs)
        self.data = None
        self.indptr = None
        self.indices = None

    def _graph_key(self):
        return (id(self.indptr), id(self.indices), id(self.data))

    def _make_engine(self, device):
        return WalkEngine.from_csr(self.indptr, self.indices, self.data, device=device)

    def get_has_nbrs(self):
        """``has_nbrs(idx)`` callback (sparse_rw.py:12-20); host-side helper, not used by the GPU path."""
        indptr = self.indptr
        return lambda idx: indptr[idx] != indptr[idx + 1]

    def get_noise_thresholds(self):
        """Per-node noisy-edge threshold ``max(mean + gamma * std, 0)`` (sparse_rw.py:22-35), bit for bit
        what the reference's row-by-row NumPy expression gives: evaluated by the native restatement of
        NumPy's pairwise reductions (``pw_noise_thresholds_csr``), by the NumPy loop itself when the
        library is not built."""
        data = np.ascontiguousarray(self.data, dtype=np.float32)
        indptr = np.ascontiguousarray(self.indptr, dtype=np.uint32)
        n = self.num_nodes
        thr = np.zeros(n, dtype=np.float32)
        try:
            from . import _lib

            lib = _lib.load()
        except Exception:  # library not built
            lib = None
        if lib is not None:
            # the promotion rule of the installed NumPy (NEP 50 from 2.0 on), so that the native path equals the
            # row-by-row NumPy expression below in the same environment
            fn = lib.pw_noise_thresholds_csr if int(np.__version__.split(".")[0]) >= 2 else lib.pw_noise_thresholds_csr_numpy1
            _lib.check(fn(indptr.ctypes.data, data.ctypes.data, n, float(self.gamma), thr.ctypes.data))
            return thr
        for i in range(n):
            row = data[indptr[i]:indptr[i + 1]]
            thr[i] = row.mean() + self.gamma * row.std()
        return np.maximum(thr, 0)


class SparseOTF(_SparseBase):
    """Sparse graph, transition probabilities on the fly (reference pecanpy.py:510-561)."""

    _mode = "SparseOTF"


class FirstOrderUnweighted(_SparseBase):
    """Uniform neighbour pick, p = q = 1, unweighted (reference pecanpy.py:293-309)."""

    _mode = "FirstOrderUnweighted"


class PreCompFirstOrder(_SparseBase):
    """First-order alias tables (reference pecanpy.py:312-361)."""

    _mode = "PreCompFirstOrder"

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.alias_j = self.alias_q = None

    def preprocess_transition_probs(self):
        """Build the per-node alias tables on the GPU (reference pecanpy.py:336-361)."""
        eng = self._get_engine()
        eng.precomp_build(1, 1, False, True)
        _, self.alias_j, self.alias_q = eng.precomp_export(True)


class PreComp(_SparseBase):
    """Second-order alias tables (reference pecanpy.py:364-507)."""

    _mode = "PreComp"

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.alias_dim = None
        self.alias_j = None
        self.alias_q = None
        self.alias_indptr = None

    def preprocess_transition_probs(self):
        """Build the sum(deg^2) second-order alias tables on the GPU (reference pecanpy.py:442-507)
        and mirror them into ``alias_dim / alias_indptr / alias_j / alias_q``."""
        eng = self._get_engine()
        eng.precomp_build(self.p, self.q, self.extend, False)
        self.alias_indptr, self.alias_j, self.alias_q = eng.precomp_export(False)
        self.alias_dim = self.indptr[1:] - self.indptr[:-1]


class DenseOTF(Base, DenseGraph):
    """Dense graph, transition probabilities on the fly (reference pecanpy.py:564-614)."""

    _mode = "DenseOTF"

    def __init__(self, *args, **kwargs):
        Base.__init__(self, *args, **kwargs)
        self._data = None
        self._nonzero = None

    def _graph_key(self):
        return (id(self._data),)

    def _make_engine(self, device):
        return WalkEngine.from_dense(self.data, device=device)

    def get_has_nbrs(self):
        nonzero = self.nonzero
        return lambda idx: bool(nonzero[idx].any())

    def get_noise_thresholds(self):
        """Dense variant (rw/dense_rw.py:11-19): float64 rows, non-zero entries only; native restatement
        of NumPy's reductions (``pw_noise_thresholds_dense``) with the NumPy loop as fallback."""
        n = self.num_nodes
        thr = np.zeros(n, dtype=np.float32)
        try:
            from . import _lib

            lib = _lib.load()
        except Exception:  # library not built
            lib = None
        if lib is not None:
            mat = np.ascontiguousarray(self.data, dtype=np.float64)
            _lib.check(lib.pw_noise_thresholds_dense(mat.ctypes.data, n, float(self.gamma), thr.ctypes.data))
            return thr
        for i in range(n):
            w = self.data[i, self.nonzero[i]]
            thr[i] = w.mean() + self.gamma * w.std()
        return np.maximum(thr, 0)
