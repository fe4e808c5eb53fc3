"""``pecanpy`` command line on the MI355X walk engine.

Flag names, defaults, the mode sanity rules and the stage order follow the reference CLI
(src/pecanpy/cli.py: flags :27-176, ``check_mode`` :179-254, ``read_graph`` :257-304, pipeline
:307-351) so that existing invocations keep working.  Walks come from the GPU; the skip-gram stage
uses gensim when it is importable and otherwise writes the walks (one per line) to ``--output``.

    pecanpy --input demo/karate.edg --output karate.emb --mode SparseOTF --p 0.5 --q 2
"""
import argparse
import os
import warnings

import numpy as np

from . import graph
from . import pecanpy
from .wrappers import Timer

MODES = ("DenseOTF", "FirstOrderUnweighted", "PreComp", "PreCompFirstOrder", "SparseOTF")

# (flag, argparse keywords) -- one table instead of twenty add_argument calls
_OPTIONS = (
    ("--input", dict(required=True, help="graph file: .edg edge list or .npz (CSR / dense)")),
    ("--output", dict(required=True, help="embedding file (.npz -> IDs/data arrays, else word2vec text)")),
    ("--task", dict(default="pecanpy", choices=("pecanpy", "tocsr", "todense"),
                    help="run node2vec, or only convert the edge list to CSR / dense .npz")),
    ("--mode", dict(default="SparseOTF", choices=MODES, help="walk engine variant")),
    ("--dimensions", dict(type=int, default=128, help="embedding size")),
    ("--walk-length", dict(type=int, default=80, help="steps per walk")),
    ("--num-walks", dict(type=int, default=10, help="walks per start vertex")),
    ("--window-size", dict(type=int, default=10, help="skip-gram window")),
    ("--epochs", dict(type=int, default=1, help="skip-gram epochs")),
    ("--workers", dict(type=int, default=0, help="host threads for the skip-gram stage (0 = all)")),
    ("--p", dict(type=float, default=1, help="return parameter")),
    ("--q", dict(type=float, default=1, help="in-out parameter")),
    ("--weighted", dict(action="store_true", help="third column of the edge list holds weights")),
    ("--directed", dict(action="store_true", help="do not add the reverse of every edge")),
    ("--verbose", dict(action="store_true", help="progress output")),
    ("--extend", dict(action="store_true", help="node2vec+ (weighted graphs)")),
    ("--gamma", dict(type=float, default=0, help="node2vec+ noisy-edge threshold = mean + gamma * std")),
    ("--random_state", dict(type=int, default=None, help="seed of the walk stream")),
    ("--delimiter", dict(type=str, default="\t", help="column separator of the edge list")),
    ("--implicit_ids", dict(action="store_true", help=".npz without IDs: use 0..N-1")),
)


def parse_args(argv=None):
    """The reference's flag set (``--walk-length`` style and ``--random_state`` style both as there)."""
    parser = argparse.ArgumentParser(
        prog="pecanpy",
        description="node2vec / node2vec+ embeddings; random walks generated on AMD MI355X GPUs",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for flag, kw in _OPTIONS:
        parser.add_argument(flag, **kw)
    return parser.parse_args(argv)


def _advise(better, current, why):
    warnings.warn(f"{why}: {better} is recommended over {current} (current selection)", stacklevel=3)


def check_mode(g, args):
    """Reject impossible mode / parameter combinations and recommend a better mode (cli.py:179-254)."""
    mode = args.mode
    first_order = args.p == 1 and args.q == 1

    # hard constraints of the two first-order modes
    if mode == "FirstOrderUnweighted" and (args.weighted or not first_order):
        raise ValueError(f"FirstOrderUnweighted only works when weighted = False and p = q = 1, got "
                         f"weighted={args.weighted}, p={args.p}, q={args.q}")
    if mode == "PreCompFirstOrder" and not first_order:
        raise ValueError(f"PreCompFirstOrder only works when p = q = 1, got p={args.p}, q={args.q}")
    if mode == "FirstOrderUnweighted":
        return

    # p = q = 1: a first-order mode does the same walk for less
    if first_order:
        if not args.weighted:
            _advise("FirstOrderUnweighted", mode, "p = q = 1 on an unweighted graph")
        elif mode != "PreCompFirstOrder":
            _advise("PreCompFirstOrder", mode, "p = q = 1")
        return

    # second order: pick by size and density
    n, rho = g.num_nodes, g.density
    if rho >= 0.2:
        best, why = "DenseOTF", f"network density = {rho:.3f} (>= 0.2)"
    elif rho < 0.001 and n < 10000:
        best, why = "PreComp", f"network density = {rho:.2e} (< 0.001) with {n} nodes (< 10000)"
    else:
        best, why = "SparseOTF", f"network density = {rho:.3g} with {n} nodes"
    if mode != best:
        _advise(best, mode, why)


def _convert_only(args):
    target = graph.SparseGraph() if args.task == "tocsr" else graph.DenseGraph()
    target.read_edg(args.input, args.weighted, args.directed, args.delimiter)
    target.save(args.output)
    raise SystemExit(0)


@Timer("load Graph")
def read_graph(args):
    """Build the graph object of ``--mode`` from ``--input`` (or convert and exit for tocsr/todense)."""
    if args.directed and args.extend:
        raise NotImplementedError("Node2vec+ not implemented for directed graph yet.")
    if args.extend and not args.weighted:
        print("NOTE: node2vec+ is equivalent to node2vec for unweighted graphs.")
    if args.task != "pecanpy":
        _convert_only(args)

    engine_cls = getattr(pecanpy, args.mode)
    g = engine_cls(args.p, args.q, args.workers, args.verbose, args.extend, args.gamma, args.random_state)
    if args.input.endswith(".npz"):
        g.read_npz(args.input, args.weighted, implicit_ids=args.implicit_ids)
    else:
        g.read_edg(args.input, args.weighted, args.directed, args.delimiter)
    check_mode(g, args)
    return g


@Timer("pre-compute transition probabilities")
def preprocess(g):
    g.preprocess_transition_probs()


@Timer("generate walks")
def simulate_walks(args, g):
    return g.simulate_walks(args.num_walks, args.walk_length)


def _dump_walks(path, walks):
    with open(path, "w", encoding="utf-8") as out:
        out.writelines(" ".join(w) + "\n" for w in walks)


def _walk_matrix(g, walks):
    """ID lists -> uint32[n_walks, L + 2] (the engine's matrix; only needed when gensim is absent)."""
    index = {name: i for i, name in enumerate(g.nodes)}
    width = max((len(w) for w in walks), default=1) + 1
    mat = np.zeros((len(walks), width), dtype=np.uint32)
    for r, w in enumerate(walks):
        mat[r, : len(w)] = [index[x] for x in w]
        mat[r, -1] = len(w)
    return mat


@Timer("train embeddings")
def learn_embeddings(args, walks, g=None):
    """Skip-gram over the walk corpus (cli.py:307-325): gensim when it is installed, else the GPU trainer of this
    package (``pecanpy_amd.embed``); ``PECANPY_AMD_DUMP_WALKS=1`` writes the walks (one per line) instead."""
    if os.environ.get("PECANPY_AMD_DUMP_WALKS"):
        _dump_walks(args.output, walks)
        return
    try:
        from gensim.models import Word2Vec
    except ImportError:
        Word2Vec = None
    if Word2Vec is None:
        if g is None:
            _dump_walks(args.output, walks)
            warnings.warn(f"gensim is not installed and no graph was passed: {len(walks)} walks written to "
                          f"{args.output} instead of embeddings", stacklevel=2)
            return
        from .embed import save_word2vec_format, train_sgns

        vecs = train_sgns(_walk_matrix(g, walks), g.num_nodes, dim=args.dimensions, window=args.window_size,
                          epochs=args.epochs, seed=args.random_state)
        if args.output.endswith(".npz"):
            np.savez(args.output, IDs=g.nodes, data=vecs)
        else:
            save_word2vec_format(args.output, g.nodes, vecs)
        return
    w2v = Word2Vec(walks, vector_size=args.dimensions, window=args.window_size, min_count=0, sg=1,
                   workers=args.workers, epochs=args.epochs, seed=args.random_state)
    vectors = w2v.wv
    if args.output.endswith(".npz"):
        np.savez(args.output, IDs=vectors.index_to_key, data=vectors.vectors)
    else:
        vectors.save_word2vec_format(args.output)


def main(argv=None):
    """read graph -> preprocess -> walks (GPU) -> embeddings."""
    args = parse_args(argv)
    args.workers = args.workers or (os.cpu_count() or 1)
    g = read_graph(args)
    preprocess(g)
    learn_embeddings(args, simulate_walks(args, g), g)


if __name__ == "__main__":
    main()
