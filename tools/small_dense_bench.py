import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from pecanpy_amd.engine import WalkEngine
from pecanpy_amd.synth import er_dense_mask
for n in (1000, 2000, 4000, 8000, 12000):
    adj = er_dense_mask(n, 0.25, seed=2)
    mat = adj.astype(np.float64)
    starts = np.concatenate([np.arange(n, dtype=np.uint32)] * 10)
    np.random.RandomState(0).shuffle(starts)
    d_starts = torch.from_numpy(starts.view(np.int32)).cuda()
    from oracle import pyoracle as orc
    bits = orc.pack_adjacency(adj)
    res = {}
    for name, eng in (("compressed", WalkEngine.from_dense(mat)), ("bits", WalkEngine.from_dense_bits(bits, n))):
        for k in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            out = eng.simulate_device("DenseOTF", 0.5, 2.0, False, d_starts, 80, seed=k)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 1e3
        res[name] = (round(ms, 2), round(eng.last_stats["total_steps"] / ms / 1e3, 1), out)
    same = torch.equal(res["compressed"][2], res["bits"][2])
    print(n, {k: v[:2] for k, v in res.items()}, "same walks:", same, flush=True)
