#!/usr/bin/env python3
"""Lane kernel vs wave-per-walk kernel on the same graph (GPU box): identical matrices, timings, counters.
usage: python tools/lanes_check.py [scales...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pecanpy_amd.engine import WalkEngine  # noqa: E402
from pecanpy_amd.synth import rmat_csr  # noqa: E402

scales = [int(a) for a in sys.argv[1:]] or [14, 18, 20]
W, L = 10, 80
dev = torch.device("cuda", 0)
for s in scales:
    indptr, indices, data = rmat_csr(s, seed=1)
    n = indptr.size - 1
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * W)
    np.random.RandomState(0).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).to(dev)
    t = time.time()
    lanes = WalkEngine.from_csr(indptr, indices, None)
    t_l = time.time() - t
    os.environ["PECANPY_AMD_NO_LANES"] = "1"
    t = time.time()
    wave = WalkEngine.from_csr(indptr, indices, None)
    t_w = time.time() - t
    del os.environ["PECANPY_AMD_NO_LANES"]
    print(f"scale {s}: create lanes {t_l:.2f}s wave {t_w:.2f}s", flush=True)
    for p, q in ((0.5, 2.0), (0.25, 4.0), (1.0, 1.0), (2.0, 0.5)):
        for rep in range(2):
            a = lanes.simulate_device("SparseOTF", p, q, False, d_starts, L, seed=0)
        sa = dict(lanes.last_stats)
        b = wave.simulate_device("SparseOTF", p, q, False, d_starts, L, seed=0)
        sb = dict(wave.last_stats)
        same = torch.equal(a, b)
        nbad = int((a != b).any(dim=1).sum().item())
        print(f"  p={p} q={q}: equal={same} bad_rows={nbad} lanes {sa['walk_kernel_ms']:.1f} ms (lane kernel {sa['lane_kernel_ms']:.1f}) "
              f"wave {sb['walk_kernel_ms']:.1f} ms  steps {sa['total_steps']} / {sb['total_steps']}  overflow {sa['overflow_reads']}/{sb['overflow_reads']} "
              f"redo {sa['redo_walks']} amb {sa['ambiguous_steps']} chain {sa['wave_chain_steps']} probes {sa['list_entries_read']} lane_kernel={sa['lane_kernel']}  "
              f"-> {sa['total_steps'] / sa['walk_kernel_ms'] / 1e3:.0f} vs {sb['total_steps'] / sb['walk_kernel_ms'] / 1e3:.0f} Msteps/s", flush=True)
        if not same:
            rows = (a != b).any(dim=1).nonzero()[:3, 0].tolist()
            for r_ in rows:
                ra, rb = a[r_].tolist(), b[r_].tolist()
                j = next(i for i in range(len(ra)) if ra[i] != rb[i])
                print(f"    row {r_}: first diff at col {j}: lanes {ra[max(0,j-2):j+2]} wave {rb[max(0,j-2):j+2]} len {ra[-1]}/{rb[-1]}")
    lanes.close()
    wave.close()
