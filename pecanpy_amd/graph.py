"""Host-side graph containers: the *input layout contract* of the walk engine.

Same public surface as the reference's ``pecanpy.graph`` (reference src/pecanpy/graph.py):
``BaseGraph`` (IDs, :19-105), ``AdjlstGraph`` (edge-list reader/writer, :108-386), ``SparseGraph``
(CSR ``indptr:uint32[N+1]``, ``indices:uint32[nnz]`` ascending per row, ``data:float32[nnz]``,
:389-528) and ``DenseGraph`` (``data:float64[N,N]`` + ``nonzero:bool[N,N]``, :531-657).
These are cold, host-only paths; only the array layouts they produce
matter to the GPU engine.
"""
import warnings

import numpy as np

__all__ = ["BaseGraph", "AdjlstGraph", "SparseGraph", "DenseGraph"]


class BaseGraph:
    """Node-ID bookkeeping shared by every graph flavour."""

    def __init__(self):
        self._node_ids = []
        self._node_idmap = {}

    @property
    def nodes(self):
        """List of node IDs, index = node index."""
        return self._node_ids

    @property
    def num_nodes(self):
        return len(self.nodes)

    @property
    def num_edges(self):
        raise NotImplementedError(
            f"{self.__class__.__name__} does not have num_edges, use the "
            f"derived classes like SparseGraph and DenseGraph instead.",
        )

    @property
    def density(self):
        n = self.num_nodes
        return self.num_edges / n / (n - 1)

    def set_node_ids(self, node_ids, implicit_ids=False, num_nodes=None):
        """Install the ID list (or canonical ``"0".."N-1"`` IDs when none are available)."""
        if node_ids is not None and not implicit_ids:
            self._node_ids = list(node_ids)
        else:
            if num_nodes is None:
                raise ValueError("Need to specify `num_nodes` when setting implicit node IDs.")
            self._node_ids = [str(i) for i in range(num_nodes)]
            if not implicit_ids:
                warnings.warn(
                    "WARNING: Implicitly set node IDs to the canonical node ordering due to "
                    "missing IDs field in the raw CSR npz file. This warning message can be "
                    "suppressed by setting implicit_ids to True in the read_npz function call, "
                    "or by setting the --implicit_ids flag in the CLI",
                    stacklevel=2,
                )
        self._node_idmap = {nid: idx for idx, nid in enumerate(self._node_ids)}

    def get_has_nbrs(self):
        raise NotImplementedError

    def get_move_forward(self):
        raise NotImplementedError


class AdjlstGraph(BaseGraph):
    """Adjacency-list graph used only to read / write edge-list files.

    ``_data[i]`` maps neighbour index -> weight for node ``i``; nodes are numbered in order of
    first appearance in the edge list.
    """

    def __init__(self):
        super().__init__()
        self._data = []
        self._num_edges = 0

    @property
    def edges_iter(self):
        for head, nbrs in enumerate(self._data):
            for tail in sorted(nbrs):
                yield head, tail, nbrs[tail]

    @property
    def edges(self):
        return list(self.edges_iter)

    @property
    def num_edges(self):
        return self._num_edges

    @staticmethod
    def _read_edge_line(edge_line, weighted, delimiter):
        terms = edge_line.strip().split(delimiter)
        id1, id2 = terms[0].strip(), terms[1].strip()
        weight = 1.0
        if weighted:
            if len(terms) != 3:
                raise ValueError(
                    f"Expecting three columns in the edge list file for a "
                    f"weighted graph, got {len(terms)} instead: {edge_line!r}",
                )
            weight = float(terms[-1])
        return id1, id2, weight

    @staticmethod
    def _is_valid_edge_weight(id1, id2, weight):
        if weight <= 0:
            warnings.warn(
                f"Non-positive edge ignored: w({id1},{id2}) = {weight}",
                RuntimeWarning,
                stacklevel=2,
            )
            return False
        return True

    def _check_edge_existence(self, id1, id2, idx1, idx2, weight):
        old = self._data[idx1].get(idx2)
        if old is not None and old != weight:
            warnings.warn(
                f"edge from {id1} to {id2} exists, with value of {old:.2f}. "
                f"Now overwrite to {weight:.2f}.",
                RuntimeWarning,
                stacklevel=2,
            )

    def get_node_idx(self, node_id):
        self.add_node(node_id)
        return self._node_idmap[node_id]

    def add_node(self, node_id):
        if node_id not in self._node_idmap:
            self._node_idmap[node_id] = self.num_nodes
            self.nodes.append(node_id)
            self._data.append({})

    def _add_edge_from_idx(self, idx1, idx2, weight):
        self._data[idx1][idx2] = weight
        self._num_edges += 1

    def add_edge(self, id1, id2, weight=1.0, directed=False):
        """Insert an edge (both directions unless ``directed``); non-positive weights are skipped."""
        if not self._is_valid_edge_weight(id1, id2, weight):
            return
        idx1 = self.get_node_idx(id1)
        idx2 = self.get_node_idx(id2)
        self._check_edge_existence(id1, id2, idx1, idx2, weight)
        self._add_edge_from_idx(idx1, idx2, weight)
        if not directed:
            self._add_edge_from_idx(idx2, idx1, weight)

    def read(self, path, weighted, directed, delimiter="\t"):
        """Read a 2- or 3-column edge list."""
        with open(path, encoding="utf-8") as f:
            for line in f:
                self.add_edge(*self._read_edge_line(line, weighted, delimiter), directed)

    def save(self, path, unweighted=False, delimiter="\t"):
        with open(path, "w", encoding="utf-8") as f:
            for h, t, w in self.edges_iter:
                cols = [self.nodes[h], self.nodes[t]]
                if not unweighted:
                    cols.append(str(w))
                f.write(delimiter.join(cols) + "\n")

    def to_csr(self):
        """CSR arrays with every row sorted by neighbour index."""
        n = len(self.nodes)
        deg = np.fromiter((len(r) for r in self._data), dtype=np.int64, count=n)
        indptr = np.zeros(n + 1, dtype=np.uint32)
        indptr[1:] = np.cumsum(deg)
        nnz = int(indptr[-1])
        indices = np.zeros(nnz, dtype=np.uint32)
        data = np.zeros(nnz, dtype=np.float32)
        pos = 0
        for row in self._data:
            for j in sorted(row):
                indices[pos] = j
                data[pos] = row[j]
                pos += 1
        return indptr, indices, data

    def to_dense(self):
        n = len(self.nodes)
        mat = np.zeros((n, n))
        for src, nbrs in enumerate(self._data):
            for dst, w in nbrs.items():
                mat[src, dst] = w
        return mat

    @classmethod
    def from_mat(cls, adj_mat, node_ids, **kwargs):
        g = cls(**kwargs)
        for node_id in node_ids:
            g.add_node(node_id)
        rows, cols = np.nonzero(adj_mat)
        for i, j in zip(rows, cols):
            g._add_edge_from_idx(i, j, adj_mat[i, j])
        return g


class SparseGraph(BaseGraph):
    """CSR graph: ``indptr`` (uint32), ``indices`` (uint32, ascending per row), ``data`` (float32)."""

    def __init__(self):
        super().__init__()
        self.data = None
        self.indptr = None
        self.indices = None

    @property
    def num_edges(self):
        if self.indptr is None:
            raise ValueError("Empty graph.")
        return self.indptr[-1]

    def read_edg(self, path, weighted, directed, delimiter="\t"):
        adj = AdjlstGraph()
        adj.read(path, weighted, directed, delimiter)
        self.set_node_ids(adj.nodes)
        self.indptr, self.indices, self.data = adj.to_csr()

    def read_npz(self, path, weighted, implicit_ids=False):
        """Load ``IDs``/``data``/``indptr``/``indices`` (a scipy CSR npz works with implicit IDs)."""
        raw = np.load(path)
        self.indptr = raw["indptr"].astype(np.uint32)
        self.indices = raw["indices"].astype(np.uint32)
        self.data = raw["data"].astype(np.float32)
        if self.data is None:
            raise ValueError("Adjacency matrix data not found.")
        if not weighted:
            self.data[:] = 1.0
        self.set_node_ids(raw.get("IDs"), implicit_ids=implicit_ids,
                          num_nodes=int(self.indptr.size - 1))

    def save(self, path):
        np.savez(path, IDs=self.nodes, data=self.data, indptr=self.indptr, indices=self.indices)

    @classmethod
    def from_adjlst_graph(cls, adjlst_graph, **kwargs):
        g = cls(**kwargs)
        g.set_node_ids(adjlst_graph.nodes)
        g.indptr, g.indices, g.data = adjlst_graph.to_csr()
        return g

    @classmethod
    def from_mat(cls, adj_mat, node_ids, **kwargs):
        g = cls(**kwargs)
        g.set_node_ids(node_ids)
        g.indptr, g.indices, g.data = AdjlstGraph.from_mat(adj_mat, node_ids).to_csr()
        return g

    @classmethod
    def from_csr(cls, indptr, indices, data=None, node_ids=None, **kwargs):
        """Build directly from CSR arrays (rows must be ascending and duplicate-free)."""
        g = cls(**kwargs)
        g.indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
        g.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        if data is None:
            data = np.ones(g.indices.size, dtype=np.float32)
        g.data = np.ascontiguousarray(data, dtype=np.float32)
        g.set_node_ids(node_ids, implicit_ids=node_ids is None, num_nodes=int(g.indptr.size - 1))
        return g


class DenseGraph(BaseGraph):
    """Dense graph: ``data`` float64[N, N] and the derived ``nonzero`` bool mask."""

    def __init__(self):
        super().__init__()
        self._data = None
        self._nonzero = None

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

    @property
    def nonzero(self):
        return self._nonzero

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
