#!/usr/bin/env python3
"""Offline fuzz of the FLOATS form's interval decision (round 6; host code of csrc/seqscan.h through pw_selftest_lane_unit_tight):
lane_decide_unit_bounded -> lane_tight_values against the sequential float32 loops.  Random rows (1 to 84 000 entries, class mixes
from no common neighbour to 70 %, random prev), random float32 biases 1/q, 1/p with p, q in [0.05, 20], draws uniform and within
three ulps / 3e-7 of the chain's partial sums.  Every position the interval decision gives must equal the chain's.
usage: python tools/fuzz_unit_tight.py [seconds=600] [seed=1]"""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pecanpy_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzz_bounded import chain32, targets
AMB=0xFFFFFFFD
lib=_lib.load()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng=np.random.default_rng(seed)
t0=time.time(); st=[0,0,0,0,0]  # verdicts, amb after bounded, settled by tight, wrong, left
while time.time()-t0<seconds:
    n = int(rng.choice([3, 7, 40, 64, 65, 300, 2500, 20000, 60000]) * rng.uniform(0.6, 1.4)) or 1
    cls = (rng.random(n) < rng.choice([0.0, 0.01, 0.02, 0.2, 0.7])).astype(np.uint8)
    if n > 1 and rng.random() < 0.7: cls[rng.integers(0, n)] = 2
    w_out, w_prev = np.float32(1.0 / rng.uniform(0.05, 20.0)), np.float32(1.0 / rng.uniform(0.05, 20.0))
    vals = np.where(cls == 1, np.float32(1.0), np.where(cls == 0, w_out, w_prev)).astype(np.float32)
    c = chain32(vals)
    r = targets(rng, c)
    chain, lane, tight = (np.empty(r.size, dtype=np.uint32) for _ in range(3))
    _lib.check(lib.pw_selftest_lane_unit_tight(cls.ctypes.data_as(C.c_void_p), n, float(w_out), float(w_prev), r.ctypes.data_as(C.c_void_p), r.size,
               chain.ctypes.data_as(C.c_void_p), lane.ctypes.data_as(C.c_void_p), tight.ctypes.data_as(C.c_void_p)))
    want = np.searchsorted(c.astype(np.float64), r, side="left").astype(np.uint32)
    assert np.array_equal(chain, want)
    amb = lane == AMB; dec = tight != AMB
    st[0]+=r.size; st[1]+=int(amb.sum()); st[2]+=int((amb&dec).sum()); st[3]+=int((tight[dec]!=chain[dec]).sum()); st[4]+=int((~dec).sum())
    if st[3]:
        bad=np.flatnonzero(dec&(tight!=chain))[:5]
        print("WRONG", n, float(w_out), float(w_prev), [(float(r[b]), int(tight[b]), int(chain[b]), int(lane[b])) for b in bad]); sys.exit(1)
print(f"fuzz_unit_tight seed {seed}: {st[0]} verdicts, {st[1]} left open by the bound, {st[2]} of them settled by the interval decision, {st[4]} left to the chain, {st[3]} wrong")
