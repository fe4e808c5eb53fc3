#!/usr/bin/env python3
"""What ONE simulate_walks()-sized call costs in the FIRST process of a fresh box (device memory nobody has used yet: ~23 ms per GB
of hipMalloc): handle creation, then one pw_simulate call on host pointers at the BASELINE size, nothing warmed up.
usage (one gpurun call per variant -- only the first process of a box is cold): python tools/cold_call.py [scale]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pecanpy_amd.engine import WalkEngine  # noqa: E402
from pecanpy_amd.synth import rmat_csr  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
indptr, indices, data = rmat_csr(scale, seed=1)
n = indptr.size - 1
starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 10)
np.random.RandomState(0).shuffle(starts)
t = time.perf_counter()
eng = WalkEngine.from_csr(indptr, indices, None)
t_create = time.perf_counter() - t
t = time.perf_counter()
host = eng.simulate("SparseOTF", 0.5, 2, False, starts, 80, seed=0)
t_first = time.perf_counter() - t
steps = eng.last_stats["total_steps"]
del host
t = time.perf_counter()
host = eng.simulate("SparseOTF", 0.5, 2, False, starts, 80, seed=1)
t_second = time.perf_counter() - t
print(f"RMAT-{scale} cold box: create {t_create * 1e3:.0f} ms (incl. HIP start-up), first host call {t_first * 1e3:.0f} ms "
      f"({steps / t_first / 1e6:.0f} M steps/s; with creation {steps / (t_first + t_create) / 1e6:.0f}), second {t_second * 1e3:.0f} ms; "
      f"ring={'off' if os.environ.get('PECANPY_AMD_NO_RING') else 'on'}")
