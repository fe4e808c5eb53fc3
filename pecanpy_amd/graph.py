"""Host-side graph containers: the *input layout contract* of the walk engine.

Public surface of the reference's ``pecanpy.graph`` (src/pecanpy/graph.py): ``BaseGraph`` (IDs,
:19-105), ``AdjlstGraph`` (edge-list reader / writer, :108-386), ``SparseGraph`` (CSR
``indptr:uint32[N+1]``, ``indices:uint32[nnz]`` ascending per row, ``data:float32[nnz]``, :389-528) and
``DenseGraph`` (``data:float64[N,N]`` + ``nonzero:bool[N,N]``, :531-657).

The implementation is this package's own: edges live in one flat ``{(src << 32) | dst: weight}`` map (or,
after a bulk read, directly in CSR arrays), conversions are vectorised NumPy, and well-formed edge-list
files are parsed by the native reader in libpecanpy_amd (``pw_edgelist_read``) with the
statement-by-statement reader as the fallback for everything that warns or raises in the reference.
"""
import ctypes
import warnings

import numpy as np

__all__ = ["BaseGraph", "AdjlstGraph", "SparseGraph", "DenseGraph"]

_SHIFT = 32
_LOW = (1 << _SHIFT) - 1


class BaseGraph:
    """Vertex-ID bookkeeping shared by every graph flavour."""

    def __init__(self):
        self._node_ids = []
        self._node_idmap = {}

    nodes = property(lambda self: self._node_ids, doc="vertex IDs; position = vertex index")
    num_nodes = property(lambda self: len(self._node_ids))

    @property
    def num_edges(self):
        raise NotImplementedError(
            f"{type(self).__name__} does not have num_edges, use the derived classes like "
            "SparseGraph and DenseGraph instead.")

    @property
    def density(self):
        n = self.num_nodes
        return self.num_edges / n / (n - 1)

    def set_node_ids(self, node_ids, implicit_ids=False, num_nodes=None):
        """Install the ID list, or the canonical IDs ``"0" .. "N-1"`` when the file carried none."""
        if node_ids is None or implicit_ids:
            if num_nodes is None:
                raise ValueError("Need to specify `num_nodes` when setting implicit node IDs.")
            if not implicit_ids:
                warnings.warn(
                    "WARNING: Implicitly set node IDs to the canonical node ordering due to missing IDs "
                    "field in the raw CSR npz file. This warning message can be suppressed by setting "
                    "implicit_ids to True in the read_npz function call, or by setting the "
                    "--implicit_ids flag in the CLI", stacklevel=2)
            node_ids = map(str, range(num_nodes))
        self._node_ids = list(node_ids)
        self._node_idmap = dict(zip(self._node_ids, range(len(self._node_ids))))

    def get_has_nbrs(self):
        raise NotImplementedError

    def get_move_forward(self):
        raise NotImplementedError


def _csr_from_flat(keys, weights, n):
    """Sorted CSR (uint32 / uint32 / float32) from ``src << 32 | dst`` keys of distinct edges."""
    keys = np.asarray(keys, dtype=np.uint64)
    order = np.argsort(keys, kind="stable")
    keys = keys[order]
    counts = np.bincount((keys >> np.uint64(_SHIFT)).astype(np.int64), minlength=n)
    indptr = np.zeros(n + 1, dtype=np.uint32)
    indptr[1:] = np.cumsum(counts)
    w64 = np.asarray(weights, dtype=np.float64)[order]
    return indptr, (keys & np.uint64(_LOW)).astype(np.uint32), w64.astype(np.float32), w64


def _native_edgelist(path, weighted, directed, delimiter):
    """``(indptr, indices, data32, data64, ids, insertions)`` from the native reader, or ``None`` when
    the file (or the build) needs the Python reader.  Host-only code of the C-ABI library; no GPU involved."""
    try:
        from . import _lib

        lib = _lib.load()
    except Exception:  # library not built: the Python reader still works
        return None
    handle = ctypes.c_void_p()
    rc = lib.pw_edgelist_read(str(path).encode(), int(bool(weighted)), int(bool(directed)),
                              delimiter.encode("utf-8", "surrogateescape"), ctypes.byref(handle))
    if rc != 0:
        return None
    try:
        dims = [ctypes.c_uint64() for _ in range(4)]
        lib.pw_edgelist_shape(handle, *[ctypes.byref(d) for d in dims])
        n, nnz, insertions, id_bytes = (int(d.value) for d in dims)
        indptr = np.empty(n + 1, dtype=np.uint32)
        indices = np.empty(nnz, dtype=np.uint32)
        data = np.empty(nnz, dtype=np.float32)
        data64 = np.empty(nnz, dtype=np.float64)
        offs = np.empty(n + 1, dtype=np.uint64)
        chars = np.empty(max(id_bytes, 1), dtype=np.uint8)
        lib.pw_edgelist_export(handle, *(a.ctypes.data_as(ctypes.c_void_p)
                                         for a in (indptr, indices, data, data64, offs, chars)))
    finally:
        lib.pw_edgelist_destroy(handle)
    blob = chars[:id_bytes].tobytes().decode("ascii")
    cuts = offs.tolist()
    ids = [blob[a:b] for a, b in zip(cuts, cuts[1:])]
    return indptr, indices, data, data64, ids, insertions


class AdjlstGraph(BaseGraph):
    """Mutable graph used to read / write edge-list files and to build the array layouts.

    Vertices are numbered in order of first appearance.  Edges are held either in a flat map keyed by
    ``src << 32 | dst`` (incremental ``add_edge``) or, after a bulk ``read``, as CSR arrays that are
    expanded into the map only if the graph is modified afterwards.
    """

    def __init__(self):
        super().__init__()
        self._w = {}          # (src << 32 | dst) -> weight (Python float, as given)
        self._csr = None      # (indptr, indices, data32, data64) of a bulk read, valid while `_w` is empty
        self._num_edges = 0   # counts insertions, like the reference (graph.py:240-243)

    # ---- storage helpers -------------------------------------------------------------------------
    def _thaw(self):
        """Expand the arrays of a bulk read into the edge map (before any modification)."""
        if self._csr is not None:
            indptr, indices, _, w64 = self._csr
            src = np.repeat(np.arange(indptr.size - 1, dtype=np.uint64), np.diff(indptr.astype(np.int64)))
            keys = (src << np.uint64(_SHIFT)) | indices.astype(np.uint64)
            self._w = dict(zip(keys.tolist(), w64.tolist()))
            self._csr = None

    def _arrays(self):
        """``(indptr, indices, data float32, data float64)``, rows in vertex order, tails ascending."""
        if self._csr is not None:
            return self._csr
        count = len(self._w)
        return _csr_from_flat(np.fromiter(self._w.keys(), dtype=np.uint64, count=count),
                              np.fromiter(self._w.values(), dtype=np.float64, count=count), self.num_nodes)

    # ---- reference surface -------------------------------------------------------------------------
    @property
    def edges_iter(self):
        """``(head, tail, weight)`` in row order, tails ascending."""
        indptr, indices, _, w64 = self._arrays()
        bounds = indptr.tolist()
        tails, weights = indices.tolist(), w64.tolist()
        for head in range(self.num_nodes):
            for e in range(bounds[head], bounds[head + 1]):
                yield head, tails[e], weights[e]

    @property
    def edges(self):
        return list(self.edges_iter)

    @property
    def num_edges(self):
        return self._num_edges

    def add_node(self, node_id):
        """Register ``node_id`` (no-op if known)."""
        if node_id not in self._node_idmap:
            self._thaw()
            self._node_idmap[node_id] = len(self._node_ids)
            self._node_ids.append(node_id)

    def get_node_idx(self, node_id):
        self.add_node(node_id)
        return self._node_idmap[node_id]

    def _put(self, src, dst, weight):
        self._w[(int(src) << _SHIFT) | int(dst)] = weight
        self._num_edges += 1

    # name kept: SparseGraph.from_mat-style callers of the reference use it
    _add_edge_from_idx = _put

    def add_edge(self, id1, id2, weight=1.0, directed=False):
        """Insert ``id1 -> id2`` (and the reverse unless ``directed``); non-positive weights are dropped
        with a warning, a changed weight of an existing edge is reported and overwritten."""
        if weight <= 0:
            warnings.warn(f"Non-positive edge ignored: w({id1},{id2}) = {weight}", RuntimeWarning, stacklevel=2)
            return
        self._thaw()
        src, dst = self.get_node_idx(id1), self.get_node_idx(id2)
        known = self._w.get((src << _SHIFT) | dst)
        if known is not None and known != weight:
            warnings.warn(f"edge from {id1} to {id2} exists, with value of {known:.2f}. "
                          f"Now overwrite to {weight:.2f}.", RuntimeWarning, stacklevel=2)
        self._put(src, dst, weight)
        if not directed:
            self._put(dst, src, weight)

    @staticmethod
    def _read_edge_line(edge_line, weighted, delimiter):
        cols = edge_line.strip().split(delimiter)
        if weighted and len(cols) != 3:
            raise ValueError("Expecting three columns in the edge list file for a weighted graph, "
                             f"got {len(cols)} instead: {edge_line!r}")
        return cols[0].strip(), cols[1].strip(), (float(cols[-1]) if weighted else 1.0)

    def read(self, path, weighted, directed, delimiter="\t"):
        """Read a 2- or 3-column edge list (native bulk reader when the graph is still empty and the
        file is well formed, line-by-line otherwise -- same result either way)."""
        if not self._node_ids and not self._w and self._csr is None and self._num_edges == 0:
            bulk = _native_edgelist(path, weighted, directed, delimiter)
            if bulk is not None:
                *arrays, ids, insertions = bulk
                self.set_node_ids(ids)
                self._csr = tuple(arrays)
                self._num_edges = insertions
                return
        with open(path, encoding="utf-8") as stream:
            for line in stream:
                self.add_edge(*self._read_edge_line(line, weighted, delimiter), directed)

    def save(self, path, unweighted=False, delimiter="\t"):
        with open(path, "w", encoding="utf-8") as out:
            for head, tail, w in self.edges_iter:
                cols = [self.nodes[head], self.nodes[tail]] + ([] if unweighted else [str(w)])
                out.write(delimiter.join(cols) + "\n")

    def to_csr(self):
        """``(indptr, indices, data)`` with every row sorted by neighbour index."""
        return self._arrays()[:3]

    def to_dense(self):
        indptr, indices, _, w64 = self._arrays()
        n = self.num_nodes
        mat = np.zeros((n, n))
        mat[np.repeat(np.arange(n), np.diff(indptr.astype(np.int64))), indices] = w64
        return mat

    @classmethod
    def from_mat(cls, adj_mat, node_ids, **kwargs):
        g = cls(**kwargs)
        for node_id in node_ids:
            g.add_node(node_id)
        adj_mat = np.asarray(adj_mat)
        for i, j in zip(*np.nonzero(adj_mat)):
            g._put(i, j, adj_mat[i, j])
        return g


def _npz_memmap(path, wanted):
    """Read-only memory maps of the members of an ``.npz`` that are stored uncompressed, C-ordered and already in
    the wanted dtype; members that are not are simply left out (the caller falls back to ``np.load``)."""
    import struct
    import zipfile

    out = {}
    try:
        with zipfile.ZipFile(path) as zf, open(path, "rb") as fh:
            for info in zf.infolist():
                name = info.filename[:-4] if info.filename.endswith(".npy") else info.filename
                if name not in wanted or info.compress_type != zipfile.ZIP_STORED:
                    continue
                fh.seek(info.header_offset)
                hdr = fh.read(30)
                if hdr[:4] != b"PK\x03\x04":
                    continue
                n_name, n_extra = struct.unpack("<HH", hdr[26:30])
                fh.seek(info.header_offset + 30 + n_name + n_extra)
                version = np.lib.format.read_magic(fh)
                read_hdr = np.lib.format.read_array_header_1_0 if version == (1, 0) else np.lib.format.read_array_header_2_0
                shape, fortran, dtype = read_hdr(fh)
                if fortran or dtype != np.dtype(wanted[name]) or len(shape) != 1:
                    continue
                if shape[0] == 0:
                    out[name] = np.zeros(0, dtype=dtype)
                else:
                    out[name] = np.memmap(path, dtype=dtype, mode="r", offset=fh.tell(), shape=shape)
    except (OSError, ValueError, zipfile.BadZipFile, struct.error):
        return {}
    return out


class SparseGraph(BaseGraph):
    """CSR graph: ``indptr`` (uint32), ``indices`` (uint32, ascending per row), ``data`` (float32)."""

    def __init__(self):
        super().__init__()
        self.data = self.indptr = self.indices = None

    @property
    def num_edges(self):
        if self.indptr is None:
            raise ValueError("Empty graph.")
        return self.indptr[-1]

    def _adopt(self, adj):
        self.set_node_ids(adj.nodes)
        self.indptr, self.indices, self.data = adj.to_csr()

    def read_edg(self, path, weighted, directed, delimiter="\t"):
        adj = AdjlstGraph()
        adj.read(path, weighted, directed, delimiter)
        self._adopt(adj)

    def read_npz(self, path, weighted, implicit_ids=False):
        """Load ``IDs`` / ``data`` / ``indptr`` / ``indices`` (a scipy CSR npz works with implicit IDs).

        Reference: graph.py:447-486.  Arrays stored uncompressed in the right dtype (what ``save`` writes) are
        MEMORY-MAPPED instead of read: a graph of RMAT-22 size (0.8 GB of CSR) goes from the page cache to the GPU
        (``pw_csr_create`` copies straight from the mapping) without an intermediate host copy."""
        members = _npz_memmap(path, {"indptr": np.uint32, "indices": np.uint32, "data": np.float32})
        raw = np.load(path)
        if "data" not in raw.files:
            raise ValueError("Adjacency matrix data not found.")

        def member(name, dtype):
            if name in members:
                return members[name]
            return np.ascontiguousarray(raw[name], dtype=dtype)

        self.indptr = member("indptr", np.uint32)
        self.indices = member("indices", np.uint32)
        if weighted:
            self.data = member("data", np.float32)
        else:   # unweighted: every stored weight counts as 1 (graph.py:480)
            self.data = np.ones(self.indices.size, dtype=np.float32)
        self.set_node_ids(raw.get("IDs"), implicit_ids=implicit_ids, num_nodes=int(self.indptr.size - 1))

    def save(self, path):
        np.savez(path, IDs=self.nodes, data=self.data, indptr=self.indptr, indices=self.indices)

    @classmethod
    def from_adjlst_graph(cls, adjlst_graph, **kwargs):
        g = cls(**kwargs)
        g._adopt(adjlst_graph)
        return g

    @classmethod
    def from_mat(cls, adj_mat, node_ids, **kwargs):
        g = cls(**kwargs)
        g._adopt(AdjlstGraph.from_mat(adj_mat, node_ids))
        g.set_node_ids(node_ids)
        return g

    @classmethod
    def from_csr(cls, indptr, indices, data=None, node_ids=None, **kwargs):
        """Build directly from CSR arrays (rows must be ascending and duplicate-free)."""
        g = cls(**kwargs)
        g.indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
        g.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        g.data = (np.ones(g.indices.size, dtype=np.float32) if data is None
                  else np.ascontiguousarray(data, dtype=np.float32))
        g.set_node_ids(node_ids, implicit_ids=node_ids is None, num_nodes=int(g.indptr.size - 1))
        return g


class DenseGraph(BaseGraph):
    """Dense graph: ``data`` float64[N, N] and the derived ``nonzero`` bool mask."""

    def __init__(self):
        super().__init__()
        self._data = self._nonzero = None

    @property
    def num_edges(self):
        if self.nonzero is None:
            raise ValueError("Empty graph.")
        return self.nonzero.sum()

    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, data):
        self._data = data.astype(float)
        self._nonzero = np.array(self._data != 0, dtype=bool)

    nonzero = property(lambda self: self._nonzero)

    def read_npz(self, path, weighted, implicit_ids=False):
        raw = np.load(path)
        self.data = raw["data"]
        if not weighted:
            self.data = self.nonzero * 1.0
        self.set_node_ids(raw.get("IDs"), implicit_ids=implicit_ids, num_nodes=self.data.shape[0])

    def read_edg(self, path, weighted, directed, delimiter="\t"):
        adj = AdjlstGraph()
        adj.read(path, weighted, directed, delimiter)
        self.set_node_ids(adj.nodes)
        self.data = adj.to_dense()

    def save(self, path):
        np.savez(path, data=self.data, IDs=self.nodes)

    @classmethod
    def from_adjlst_graph(cls, adjlst_graph, **kwargs):
        g = cls(**kwargs)
        g.set_node_ids(adjlst_graph.nodes)
        g.data = adjlst_graph.to_dense()
        return g

    @classmethod
    def from_mat(cls, adj_mat, node_ids, **kwargs):
        g = cls(**kwargs)
        g.data = adj_mat
        g.set_node_ids(node_ids)
        return g
