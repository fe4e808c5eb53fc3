#!/usr/bin/env python3
"""DenseOTF on a WEIGHTED dense graph (the regime the reference's README recommends DenseOTF for; rw/dense_rw.py:74-118):
Erdos-Renyi N nodes, density 0.25, hashed U(0,1] float64 weights, node2vec and node2vec+ -- walk-steps/s of the float64 path
(round 6: walk_dense_weighted_kernel, csrc/walk_dense_w.hip.h -- the float64-bounded decision over one stream of cur's compressed
row; PECANPY_AMD_DENSE_NO_WFAST=1: walk_kernel<double, true, ...>, the exact float64 binade scan, rounds 1-5).
usage: python tools/dense_weighted_bench.py [N=20000] [num_walks=10] [walk_length=80] [density=0.25]
Prints one JSON line per mode (declared bytes per step: 12 d(cur) + N / 8 + 12, node2vec+ adds 8 d(prev))."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from pecanpy_amd.engine import WalkEngine
    from pecanpy_amd import _lib
    import ctypes as C

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 80
    dens = float(sys.argv[4]) if len(sys.argv) > 4 else 0.25
    rng = np.random.default_rng(1)
    t = time.time()
    up = np.triu(rng.random((n, n), dtype=np.float32) < dens, 1)
    w = rng.random((n, n), dtype=np.float32).astype(np.float64) * 0.999 + 0.001
    data = np.where(up, w, 0.0)
    del up, w
    data = data + data.T
    print(f"# ER-{n} weighted dense matrix in {time.time() - t:.1f}s, nnz {int((data != 0).sum())}", flush=True)
    t = time.perf_counter()
    eng = WalkEngine.from_dense(data)
    print(f"# pw_dense_create {time.perf_counter() - t:.1f}s", flush=True)
    thr = np.zeros(n, dtype=np.float32)
    _lib.check(_lib.load().pw_noise_thresholds_dense(C.c_void_p(data.ctypes.data), n, C.c_double(0.0), C.c_void_p(thr.ctypes.data)))
    eng.set_thresholds(thr)
    deg = (data != 0).sum(1)
    del data
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * W)
    np.random.RandomState(0).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    for extend in (False, True):
        for k in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = eng.simulate_device("DenseOTF", 0.5, 2.0, extend, d_starts, L, seed=k)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) * 1e3
        st = eng.last_stats
        steps = st["total_steps"]
        # reference format: 8 N + 2 N + 12 per step, node2vec+ adds 8 N + 4 N.  This build's declared format: 12 bytes
        # per non-zero of cur's row (float64 weight + uint32 column) + N / 8 of prev's packed row + 12 (draw, output); node2vec+
        # adds 8 bytes per non-zero of prev's row (its weights, gathered by rank; its columns are never read)
        dmean = float(deg.mean())
        ref_b = steps * ((10 if not extend else 22) * n + 12)
        ours = steps * (12 * dmean + n / 8 + (8 * dmean if extend else 0) + 12)
        print(json.dumps({"workload": f"ER-{n} density {dens} weighted DenseOTF {'node2vec+' if extend else 'node2vec'} p=0.5 q=2, {W} x {L}",
                          "ms_per_pass": round(ms, 2), "value": round(steps / ms / 1e3, 2), "unit": "million walk-steps/s",
                          "walk_kernel_ms": round(st["walk_kernel_ms"], 2), "redo_walks": st["redo_walks"],
                          "roofline": {"bound": "hbm", "declared_bytes": ours, "achieved": round(ours / ms / 1e6, 1), "peak": 8000.0,
                                       "unit": "GB/s", "frac": round(ours / st["walk_kernel_ms"] / 1e6 / 8000, 3)},
                          "reference_format_bytes": ref_b,
                          "bounded_kernel": not bool(os.environ.get("PECANPY_AMD_DENSE_NO_WFAST"))}), flush=True)


if __name__ == "__main__":
    main()
