"""Synthetic graph generators for the benchmark configurations.

Host-side NumPy only; these build the *inputs* of the hot path (CSR arrays in the layout of the
reference's ``SparseGraph``: ``indptr:uint32[N+1]``, ``indices:uint32[nnz]`` ascending per row,
``data:float32[nnz]`` -- reference ``src/pecanpy/graph.py:409-413``).
"""
import numpy as np

__all__ = ["rmat_csr", "er_dense_mask", "hash_edge_weights", "csr_from_edges"]


def csr_from_edges(src, dst, num_nodes, weights=None):
    """Sorted, duplicate-free CSR from directed (src, dst) pairs (already symmetrised if needed)."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    key = src * np.int64(num_nodes) + dst
    if weights is None:
        key = np.unique(key)
        data = None
    else:
        key, first = np.unique(key, return_index=True)
        data = np.asarray(weights, dtype=np.float32)[first]
    rows = key // num_nodes
    cols = (key - rows * num_nodes).astype(np.uint32)
    counts = np.bincount(rows, minlength=num_nodes)
    indptr = np.zeros(num_nodes + 1, dtype=np.uint64)
    np.cumsum(counts, out=indptr[1:])
    if indptr[-1] >= 2**32:
        raise ValueError("nnz does not fit uint32 (reference CSR uses uint32 indptr)")
    if data is None:
        data = np.ones(cols.size, dtype=np.float32)
    return indptr.astype(np.uint32), cols, data


def rmat_csr(scale, edge_factor=8, a=0.57, b=0.19, c=0.19, seed=1, weighted=False):
    """Graph500-style Kronecker (R-MAT) graph: ``2**scale`` vertices, ``edge_factor * 2**scale``
    generated undirected edges, self loops dropped, de-duplicated, symmetrised, rows sorted, no
    vertex relabelling.  Returns ``(indptr, indices, data)``.

    ``weighted=True`` draws symmetric float32 weights in (0, 1] from a hash of the undirected
    edge (BASELINE config C5).
    """
    n = 1 << scale
    m = edge_factor << scale
    rng = np.random.default_rng(seed)
    src = np.zeros(m, dtype=np.int64)
    dst = np.zeros(m, dtype=np.int64)
    ab = a + b
    abc = a + b + c
    for _ in range(scale):
        r = rng.random(m)
        sbit = r >= ab
        dbit = ((r >= a) & (r < ab)) | (r >= abc)
        src = (src << 1) | sbit
        dst = (dst << 1) | dbit
    keep = src != dst
    src, dst = src[keep], dst[keep]
    s2 = np.concatenate([src, dst])
    d2 = np.concatenate([dst, src])
    del src, dst
    indptr, indices, data = csr_from_edges(s2, d2, n)
    if weighted:
        data = hash_edge_weights(indptr, indices, seed)
    return indptr, indices, data


def hash_edge_weights(indptr, indices, seed=1):
    """Symmetric float32 weights in (0, 1] from a 64-bit mix of the undirected edge (u<v)."""
    n = indptr.size - 1
    deg = np.diff(indptr.astype(np.int64))
    rows = np.repeat(np.arange(n, dtype=np.uint64), deg)
    cols = indices.astype(np.uint64)
    lo = np.minimum(rows, cols)
    hi = np.maximum(rows, cols)
    x = lo * np.uint64(0x9E3779B97F4A7C15) + hi * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(seed)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xFF51AFD7ED558CCD)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xC4CEB9FE1A85EC53)
    x ^= x >> np.uint64(33)
    # 24 random bits -> (0, 1] with float32-exact values
    u = ((x >> np.uint64(40)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0**-24)
    return u.astype(np.float32)


def er_dense_mask(n, density, seed=1):
    """Erdos-Renyi undirected adjacency as a symmetric boolean matrix without self loops."""
    rng = np.random.default_rng(seed)
    upper = rng.random((n, n)) < density
    upper = np.triu(upper, 1)
    return upper | upper.T
