#!/usr/bin/env python3
"""Experiment (GPU box): a shard-sized job array walked in chunks, one engine after the other vs TWO engines (two handles of
the same graph: stream, queues and draw buffer of their own) driven by two host threads on alternate chunks -- does the
second handle's first round fill the wave slots the first one's short later rounds leave empty?
usage: python tools/two_engines.py [--scale 22] [--jobs 5600000]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from pecanpy_amd.engine import WalkEngine, tapered_bounds
    from pecanpy_amd.synth import rmat_csr

    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--jobs", type=int, default=5_600_000)
    ap.add_argument("--passes", type=int, default=3)
    args = ap.parse_args()
    L = 80
    indptr, indices, _ = rmat_csr(args.scale, seed=1)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 10)
    np.random.RandomState(0).shuffle(starts)
    starts = starts[: args.jobs]
    engines = [WalkEngine.from_csr(indptr, indices, None) for _ in range(2)]
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    ref = torch.empty((starts.size, L + 2), dtype=torch.int32, device="cuda")
    out = torch.empty_like(ref)
    deg = np.diff(indptr)
    draws = np.concatenate([[0], np.cumsum(np.where(deg[starts] > 0, L, 0), dtype=np.int64)])

    def walk(eng, lo, hi, seed, dst):
        eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts[lo:hi], L, seed=seed, stream_skip=int(draws[lo]), out=dst[lo:hi])

    def timed(fn, label, check):
        ms = []
        for k in range(args.passes + 1):
            if check:
                walk(engines[0], 0, starts.size, k, ref)
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn(k)
            torch.cuda.synchronize()
            ms.append(round((time.perf_counter() - t) * 1e3, 2))
            if check and not torch.equal(ref, out):
                print(f"{label}: MISMATCH at pass {k}", flush=True)
        print(f"{label}: ms {ms[1:]}", flush=True)

    timed(lambda k: walk(engines[0], 0, starts.size, k, out), "one call", True)
    def cuts(b):
        return list(zip(b[:-1], b[1:]))

    schemes = {"4 tapered": tapered_bounds(starts.size, 4), "2 (75/25)": cuts([0, starts.size * 3 // 4, starts.size]),
               "4 equal": cuts([starts.size * i // 4 for i in range(5)]), "8 equal": cuts([starts.size * i // 8 for i in range(9)])}
    for name, chunks in schemes.items():

        def seq(k):
            for lo, hi in chunks:
                walk(engines[0], lo, hi, k, out)

        def two(k):
            def run(e):
                for lo, hi in chunks[e::2]:
                    walk(engines[e], lo, hi, k, out)
            th = [threading.Thread(target=run, args=(e,)) for e in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()

        timed(seq, f"{name}, one engine", True)
        timed(two, f"{name}, two engines", True)


if __name__ == "__main__":
    main()
