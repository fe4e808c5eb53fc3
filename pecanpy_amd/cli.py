"""Command line front-end: the reference's ``pecanpy`` CLI on the MI355X walk engine.

Same flags, defaults, mode checks and stage order as reference src/pecanpy/cli.py (:27-176 flags,
:179-254 ``check_mode``, :257-304 ``read_graph``, :307-351 pipeline).  Walk generation runs on the
GPU; the skip-gram step uses gensim when it is installed (embedding training is outside the scope
of this engine) and otherwise ``--output`` receives the walks themselves (one walk per line).

    pecanpy --input demo/karate.edg --output karate.emb --mode SparseOTF --p 0.5 --q 2
"""
import argparse
import os
import warnings

import numpy as np

from . import graph
from . import pecanpy
from .wrappers import Timer

MODES = ["DenseOTF", "FirstOrderUnweighted", "PreComp", "PreCompFirstOrder", "SparseOTF"]


def parse_args(argv=None):
    """Parse node2vec arguments (flag set of the reference CLI)."""
    ap = argparse.ArgumentParser(
        description="Run pecanpy, a parallelized, efficient, and accelerated Python implementation "
                    "of node2vec (walks generated on AMD MI355X GPUs)",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter,
    )
    ap.add_argument("--input", required=True, help="Input graph (.edg or .npz) file path.")
    ap.add_argument("--output", required=True,
                    help="Output embeddings file path. Save as .npz file if the specified file path "
                         "ends with .npz, otherwise save as a text file using the gensim "
                         "save_word2vec_format method.")
    ap.add_argument("--task", default="pecanpy", choices=["pecanpy", "tocsr", "todense"],
                    help="Task to be performed.")
    ap.add_argument("--mode", default="SparseOTF", choices=MODES, help="PecanPy execution mode.")
    ap.add_argument("--dimensions", type=int, default=128, help="Number of dimensions.")
    ap.add_argument("--walk-length", type=int, default=80, help="Length of walk per source.")
    ap.add_argument("--num-walks", type=int, default=10, help="Number of walks per source.")
    ap.add_argument("--window-size", type=int, default=10, help="Context size for optimization.")
    ap.add_argument("--epochs", type=int, default=1, help="Number of epochs in SGD when training Word2Vec")
    ap.add_argument("--workers", type=int, default=0,
                    help="Number of parallel workers (0 to use all available threads).")
    ap.add_argument("--p", type=float, default=1, help="Return hyperparameter.")
    ap.add_argument("--q", type=float, default=1, help="Inout hyperparameter.")
    ap.add_argument("--weighted", action="store_true", help="Boolean specifying (un)weighted.")
    ap.add_argument("--directed", action="store_true", help="Graph is (un)directed.")
    ap.add_argument("--verbose", action="store_true", help="Print out training details")
    ap.add_argument("--extend", action="store_true", help="Use node2vec+ extension")
    ap.add_argument("--gamma", type=float, default=0, help="Noisy edge threshold parameter.")
    ap.add_argument("--random_state", type=int, default=None, help="Random seed for generating random walks.")
    ap.add_argument("--delimiter", type=str, default="\t", help="Delimiter used between node IDs.")
    ap.add_argument("--implicit_ids", action="store_true",
                    help="If set, use canonical node ordering for the node IDs.")
    return ap.parse_args(argv)


def check_mode(g, args):
    """Mode sanity checks and recommendations by graph size / density (reference cli.py:179-254)."""
    mode, weighted, p, q = args.mode, args.weighted, args.p, args.q

    if mode == "FirstOrderUnweighted":
        if not p == q == 1 or weighted:
            raise ValueError(
                f"FirstOrderUnweighted only works when weighted = False and "
                f"p = q = 1, got {weighted=}, {p=}, {q=}",
            )
        return
    if p == q == 1 and not weighted:
        warnings.warn(
            "When p = 1 and q = 1 with unweighted graph, it is highly recommended to use "
            f"FirstOrderUnweighted over {mode} (current selection). The runtime could be improved "
            "greatly with improved  memory usage.",
            stacklevel=2,
        )
        return
    if mode == "PreCompFirstOrder":
        if not p == q == 1:
            raise ValueError(f"PreCompFirstOrder only works when p = q = 1, got {p=}, {q=}")
        return
    if p == 1 == q:
        warnings.warn(
            "When p = 1 and q = 1, it is highly recommended to use PreCompFirstOrder over "
            f"{mode} (current selection). The runtime could be improved greatly with low memory usage.",
            stacklevel=2,
        )
        return

    size, dens = g.num_nodes, g.density
    if dens >= 0.2 and mode != "DenseOTF":
        warnings.warn(f"Network density = {dens:.3f} (> 0.2), it is recommended to use DenseOTF "
                      f"over {mode} (current selection)", stacklevel=2)
    if dens < 0.001 and size < 10000 and mode != "PreComp":
        warnings.warn(f"Network density = {dens:.2e} (< 0.001) with {size} nodes (< 10000), it is "
                      f"recommended to use PreComp over {mode} (current selection)", stacklevel=2)
    if 0.001 <= dens < 0.2 and mode != "SparseOTF":
        warnings.warn(f"Network density = {dens:.3f}, it is recommended to use SparseOTF over "
                      f"{mode} (current selection)", stacklevel=2)
    if dens < 0.001 and size >= 10000 and mode != "SparseOTF":
        warnings.warn(f"Network density = {dens:.3f} (< 0.001) with {size} nodes (>= 10000), it is "
                      f"recommended to use SparseOTF over {mode} (current selection)", stacklevel=2)


@Timer("load Graph")
def read_graph(args):
    """Read the input network as CSR (sparse modes) or dense matrix (DenseOTF)."""
    if args.directed and args.extend:
        raise NotImplementedError("Node2vec+ not implemented for directed graph yet.")
    if args.extend and not args.weighted:
        print("NOTE: node2vec+ is equivalent to node2vec for unweighted graphs.")

    if args.task in ("tocsr", "todense"):  # conversion only
        g = graph.SparseGraph() if args.task == "tocsr" else graph.DenseGraph()
        g.read_edg(args.input, args.weighted, args.directed, args.delimiter)
        g.save(args.output)
        raise SystemExit(0)

    cls = getattr(pecanpy, args.mode, None)
    g = cls(args.p, args.q, args.workers, args.verbose, args.extend, args.gamma, args.random_state)
    if args.input.endswith(".npz"):
        g.read_npz(args.input, args.weighted, implicit_ids=args.implicit_ids)
    else:
        g.read_edg(args.input, args.weighted, args.directed, args.delimiter)
    check_mode(g, args)
    return g


@Timer("train embeddings")
def learn_embeddings(args, walks):
    """Skip-gram on the walk corpus (gensim; reference cli.py:307-325)."""
    try:
        from gensim.models import Word2Vec
    except ImportError:
        path = args.output
        with open(path, "w", encoding="utf-8") as f:
            for walk in walks:
                f.write(" ".join(walk) + "\n")
        warnings.warn(f"gensim is not installed: wrote the {len(walks)} walks to {path} instead of "
                      "embeddings (Word2Vec training is outside this engine)", stacklevel=2)
        return
    model = Word2Vec(walks, vector_size=args.dimensions, window=args.window_size, min_count=0, sg=1,
                     workers=args.workers, epochs=args.epochs, seed=args.random_state)
    if args.output.endswith(".npz"):
        np.savez(args.output, IDs=model.wv.index_to_key, data=model.wv.vectors)
    else:
        model.wv.save_word2vec_format(args.output)


@Timer("pre-compute transition probabilities")
def preprocess(g):
    g.preprocess_transition_probs()


@Timer("generate walks")
def simulate_walks(args, g):
    return g.simulate_walks(args.num_walks, args.walk_length)


def main(argv=None):
    """Pipeline: read graph -> preprocess -> walks (GPU) -> embeddings."""
    args = parse_args(argv)
    if args.workers == 0:
        args.workers = os.cpu_count() or 1
    g = read_graph(args)
    preprocess(g)
    walks = simulate_walks(args, g)
    learn_embeddings(args, walks)


if __name__ == "__main__":
    main()
