import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pecanpy_amd.engine import WalkEngine
from pecanpy_amd.synth import rmat_csr
indptr, indices, data = rmat_csr(15, seed=4, weighted=True)
n = indptr.size - 1
starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 3)
np.random.RandomState(2).shuffle(starts)
d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
eng = WalkEngine.from_csr(indptr, indices, data)
os.environ["PECANPY_AMD_NO_WLANES"] = "1"
ref = eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 40, seed=7).clone()
del os.environ["PECANPY_AMD_NO_WLANES"]
for env in ({}, {"PECANPY_AMD_NO_WCKPT": "1"}):
    os.environ.update(env)
    a = eng.simulate_device("SparseOTF", 0.5, 2.0, False, d_starts, 40, seed=7)
    st = eng.last_stats
    bad = (a != ref).any(dim=1)
    print(env, "lane_kernel", st["lane_kernel"], "eager", st["eager_steps"], "bad rows", int(bad.sum().item()))
    if bad.any():
        deg = np.diff(indptr.astype(np.int64))
        rows = bad.nonzero()[:5, 0].tolist()
        for r_ in rows:
            ra, rb = a[r_].tolist(), ref[r_].tolist()
            j = next(i for i in range(len(ra)) if ra[i] != rb[i])
            print("  row", r_, "first diff col", j, "cur", rb[j-1], "deg(cur)", int(deg[rb[j-1]]), "prev", rb[j-2] if j>=2 else None, "lane", ra[j], "ref", rb[j])
    for k in env: del os.environ[k]
