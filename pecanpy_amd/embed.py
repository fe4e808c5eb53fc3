"""Skip-gram embeddings from a walk matrix on the GPU (reference: gensim ``Word2Vec(walks, sg=1, ...)`` in
``Base.embed`` / ``cli.learn_embeddings``, src/pecanpy/pecanpy.py:276-290, cli.py:307-325).

``train_sgns`` runs word2vec's skip-gram-with-negative-sampling update as a HIP kernel (``pw_sgns_train``,
csrc/sgns.hip.h) directly on the ``uint32[n_jobs, L+2]`` matrix the walk engine produces -- no ``List[List[str]]`` corpus
in between.  Same model and defaults as gensim's (negative=5, ns_exponent=0.75, sample=1e-3, alpha 0.025 -> 1e-4,
shrunk windows over the subsampled walk); not bit-comparable with gensim's own random streams -- the deterministic
single-wavefront mode is checked against a sequential CPU restatement of the algorithm instead (tests/test_gpu_sgns.py).
"""
import ctypes as C

import numpy as np

from . import _lib

__all__ = ["train_sgns", "save_word2vec_format"]


def train_sgns(walk_matrix, num_nodes, dim=128, window=10, epochs=1, negative=5, alpha=0.025, min_alpha=1e-4,
               sample=1e-3, seed=None, device=0, workers=0):
    """float32[num_nodes, dim] input vectors (``wv``) after ``epochs`` passes over the walks.

    ``workers=0`` (default): hogwild, as many wavefronts as the corpus feeds; ``workers=1``: one wavefront in sentence
    order -- deterministic under ``seed`` (the counterpart of gensim's ``workers=1``), slow."""
    lib = _lib.load()
    mat = np.ascontiguousarray(walk_matrix, dtype=np.uint32)
    if mat.ndim != 2 or mat.shape[1] < 3:
        raise ValueError("walk matrix must be uint32[n_walks, walk_length + 2]")
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1)[0])
    out = np.zeros((int(num_nodes), int(dim)), dtype=np.float32)
    _lib.check(lib.pw_sgns_train(int(device), mat.ctypes.data, mat.shape[0], mat.shape[1] - 2, int(num_nodes), int(dim),
                                 int(window), int(negative), int(epochs), float(alpha), float(min_alpha), float(sample),
                                 int(seed) & 0xFFFFFFFF, int(workers), out.ctypes.data))
    return out


def save_word2vec_format(path, node_ids, vectors):
    """The text format gensim's ``KeyedVectors.save_word2vec_format`` writes (cli.py:323-325)."""
    with open(path, "w", encoding="utf-8") as f:
        f.write(f"{len(node_ids)} {vectors.shape[1]}\n")
        for name, vec in zip(node_ids, vectors):
            f.write(str(name) + " " + " ".join(f"{x:.6f}" for x in vec) + "\n")
