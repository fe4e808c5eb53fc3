#!/usr/bin/env python3
"""Host-pointer boundary (pw_simulate: NumPy in, NumPy out) against the device-resident call, same workload.
usage: python tools/host_path.py [scale]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pecanpy_amd.engine import WalkEngine  # noqa: E402
from pecanpy_amd.synth import rmat_csr  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
indptr, indices, data = rmat_csr(scale, seed=1)
n = indptr.size - 1
starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 10)
np.random.RandomState(0).shuffle(starts)
eng = WalkEngine.from_csr(indptr, indices, None)
d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
for _ in range(2):
    t = time.perf_counter(); dev = eng.simulate_device("SparseOTF", 0.5, 2, False, d_starts, 80, seed=0); torch.cuda.synchronize(); t_dev = time.perf_counter() - t
host = None
for _ in range(2):
    del host   # (releasing the previous 0.86 GB result costs ~40 ms of munmap: not part of the call)
    t = time.perf_counter(); host = eng.simulate("SparseOTF", 0.5, 2, False, starts, 80, seed=0); t_host = time.perf_counter() - t
ok = np.array_equal(host, dev.cpu().numpy().view(np.uint32))
gb = host.nbytes / 1e9
print(f"RMAT-{scale}: device call {t_dev * 1e3:.1f} ms, host call {t_host * 1e3:.1f} ms ({gb:.2f} GB out, "
      f"{gb / max(t_host - t_dev, 1e-9):.1f} GB/s for the copies), equal={ok}")
