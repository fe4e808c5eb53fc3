"""Known-answer vectors held by the reference's own walk test (data only).

Source of the numbers: reference test/test_walk.py:10-18 (5-node adjacency), :21-82 (expected
walks for ``simulate_walks(2, 3)``, ``p=q=1``, ``random_state=0``).  Each walk is written here as
a 4-letter word over the node IDs a..e; the label "PreCompFirstOrder" of the reference table is
produced by ``PreComp`` there (test/test_walk.py:89), hence identical to "PreComp".
"""
import numpy as np

IDS = list("abcde")
_EDGES = [(0, 1), (1, 2), (2, 3), (2, 4), (3, 4)]
MAT = np.zeros((5, 5), dtype=int)
for _u, _v in _EDGES:
    MAT[_u, _v] = MAT[_v, _u] = 1

_WORDS = {
    "FirstOrderUnweighted": "cbcd dcde edcb edcb baba babc cede dcbc abcd abcb",
    "PreComp": "cded dcde edce edec bcec bcdc cded dced abab abce",
    "SparseOTF": "cded decd eced eced bcec babc cede dece abcb abcd",
    "DenseOTF": "cded decd eced eced bcec babc cede dece abcb abcd",
}
_WORDS["PreCompFirstOrder"] = _WORDS["PreComp"]
WALKS = {k: [list(w) for w in v.split()] for k, v in _WORDS.items()}
