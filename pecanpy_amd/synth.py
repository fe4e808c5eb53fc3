"""Synthetic graph generators for the benchmark configurations.

Host-side NumPy only; these build the *inputs* of the hot path (CSR arrays in the layout of the
reference's ``SparseGraph``: ``indptr:uint32[N+1]``, ``indices:uint32[nnz]`` ascending per row,
``data:float32[nnz]`` -- reference ``src/pecanpy/graph.py:409-413``).
"""
import numpy as np

__all__ = ["rmat_csr", "er_dense_mask", "hash_edge_weights", "csr_from_edges", "ring_lattice_csr", "holme_kim_csr",
           "bipartite_hubs_csr", "gnm_csr"]


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


# ---- graph families for the exactness evidence of the lane path (tests/test_gpu_verify.py) ----------------------
# R-MAT rows rarely produce the arithmetic coincidences on which a float32 rounding argument can fail (power-of-two
# row totals, values that tie in every binade, common neighbours in contiguous blocks): these families do.

def ring_lattice_csr(n, k):
    """2k-regular ring lattice: vertex i is adjacent to i +- 1..k (mod n).  Every edge closes many triangles
    (neighbours at ring distance t share 2k - t - 1 [t <= k] neighbours), all degrees are 2k -- a power of two when
    k is -- and the common neighbours of an edge sit in contiguous blocks of the row."""
    if not 2 * k < n:
        raise ValueError("need 2k < n")
    offs = np.concatenate([np.arange(-k, 0), np.arange(1, k + 1)]).astype(np.int64)
    cols = (np.arange(n, dtype=np.int64)[:, None] + offs[None, :]) % n
    cols.sort(axis=1)
    indptr = (np.arange(n + 1, dtype=np.uint64) * np.uint64(2 * k)).astype(np.uint32)
    indices = cols.astype(np.uint32).reshape(-1)
    return indptr, indices, np.ones(indices.size, dtype=np.float32)


def holme_kim_csr(n, m, p_triad=0.7, seed=1):
    """Holme-Kim growing network: preferential attachment with triad formation (after every preferential link to w,
    with probability p_triad the next link goes to a neighbour of w): power-law degrees AND high clustering -- long
    rows in which a large share of the neighbours are common neighbours.  Sequential by definition (host loop)."""
    rng = np.random.default_rng(seed)
    adj = [[] for _ in range(n)]
    rep = []                              # every vertex once per incident edge: uniform choice = preferential
    for i in range(m + 1):                # seed clique
        for j in range(i):
            adj[i].append(j); adj[j].append(i); rep += [i, j]
    src, dst = [], []
    rnd = rng.random(n * m)
    pick = rng.integers(0, 1 << 62, n * m * 2)
    t = 0
    for v in range(m + 1, n):
        mine = set()
        last = -1
        for _ in range(m):
            w = -1
            if last >= 0 and rnd[t] < p_triad:
                cand = adj[last]
                for _try in range(4):
                    c = cand[pick[2 * t] % len(cand)] if _try == 0 else cand[(pick[2 * t] >> (8 * _try)) % len(cand)]
                    if c != v and c not in mine:
                        w = c
                        break
            if w < 0:
                for _try in range(8):
                    c = rep[(pick[2 * t + 1] >> (6 * _try)) % len(rep)]
                    if c != v and c not in mine:
                        w = c
                        break
            t += 1
            if w < 0:
                continue
            mine.add(w)
            last = w
        for w in mine:
            adj[v].append(w); adj[w].append(v); rep += [v, w]
            src.append(v); dst.append(w)
    for i in range(m + 1):
        for j in range(i):
            src.append(i); dst.append(j)
    s2 = np.array(src + dst, dtype=np.int64)
    d2 = np.array(dst + src, dtype=np.int64)
    return csr_from_edges(s2, d2, n)


def bipartite_hubs_csr(n_hubs, n_leaves, hub_degree, seed=1):
    """Random bipartite graph: n_hubs vertices adjacent to hub_degree random leaves each.  No triangles at all (every
    common-neighbour list is empty: the closed form of the exact decision) and rows of hub_degree entries, long enough
    for the float32 drift to matter."""
    rng = np.random.default_rng(seed)
    src, dst = [], []
    for h in range(n_hubs):
        leaves = rng.choice(n_leaves, size=hub_degree, replace=False).astype(np.int64) + n_hubs
        src.append(np.full(hub_degree, h, dtype=np.int64))
        dst.append(leaves)
    s, d = np.concatenate(src), np.concatenate(dst)
    return csr_from_edges(np.concatenate([s, d]), np.concatenate([d, s]), n_hubs + n_leaves)


def gnm_csr(n, m, seed=1):
    """Sparse Erdos-Renyi G(n, m): m random undirected edges (self loops and duplicates dropped)."""
    rng = np.random.default_rng(seed)
    s = rng.integers(0, n, m)
    d = rng.integers(0, n, m)
    keep = s != d
    s, d = s[keep], d[keep]
    return csr_from_edges(np.concatenate([s, d]), np.concatenate([d, s]), n)
