#!/usr/bin/env python3
"""Offline fuzz of the float64-bounded decisions of round 5 (host code of csrc/seqscan.h through the self-test hooks):
lane_decide_unit_bounded (closed-form prefix sums, sum-of-prefix-sums drift bound) and lane_decide_weighted (tables + the
same bound) against the sequential float32 loops.  Random rows (sizes, class mixes, biases / weights), draws uniform and
within a few ulps of the chain's partial sums.  Every verdict that is not "left open" must equal the chain's position.
usage: python tools/fuzz_bounded.py [seconds=600] [seed=1]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pecanpy_amd import _lib  # noqa: E402

AMB = 0xFFFFFFFD


def chain32(vals):
    tot = np.float32(0)
    for v in vals:
        tot = np.float32(tot + v)
    x = (vals / tot).astype(np.float32)
    return np.cumsum(x, dtype=np.float32) if False else np.array(np.frompyfunc(lambda a, b: np.float32(a + b), 2, 1).accumulate(x, dtype=object), dtype=np.float32)


def targets(rng, c, n_uni=200, n_near=200):
    cd = c.astype(np.float64)
    idx = rng.integers(0, c.size, n_near)
    ulps = rng.integers(-3, 4, n_near)
    near = cd[idx]
    for _ in range(3):
        near = np.where(ulps > 0, np.nextafter(near, 2.0), np.where(ulps < 0, np.nextafter(near, 0.0), near))
        ulps = ulps - np.sign(ulps)
    rel = cd[rng.integers(0, c.size, n_near)] * (1 + rng.normal(0, 3e-7, n_near))
    return np.clip(np.concatenate([rng.random(n_uni), near, rel]), 0.0, np.nextafter(1.0, 0.0))


def main(seconds=600.0, seed=1):
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    t0 = time.time()
    stats = {"unit": [0, 0, 0], "weighted": [0, 0, 0]}     # verdicts, decided, wrong
    while time.time() - t0 < seconds:
        n = int(rng.choice([1, 2, 3, 7, 40, 64, 65, 300, 2500, 20000]) * rng.uniform(0.6, 1.4)) or 1
        cls = (rng.random(n) < rng.choice([0.0, 0.02, 0.2, 0.7])).astype(np.uint8)
        if n > 1 and rng.random() < 0.7:
            cls[rng.integers(0, n)] = 2
        # ---- unit rows, arbitrary biases
        w_out, w_prev = np.float32(1.0 / rng.uniform(0.05, 20.0)), np.float32(1.0 / rng.uniform(0.05, 20.0))
        vals = np.where(cls == 1, np.float32(1.0), np.where(cls == 0, w_out, w_prev)).astype(np.float32)
        c = chain32(vals)
        r = targets(rng, c)
        chain, lane = (np.empty(r.size, dtype=np.uint32) for _ in range(2))
        _lib.check(lib.pw_selftest_lane_unit_bounded(cls.ctypes.data_as(C.c_void_p), n, float(w_out), float(w_prev), r.ctypes.data_as(C.c_void_p),
                                                     r.size, chain.ctypes.data_as(C.c_void_p), lane.ctypes.data_as(C.c_void_p)))
        want = np.searchsorted(c.astype(np.float64), r, side="left").astype(np.uint32)
        assert np.array_equal(chain, want)
        dec = lane != AMB
        stats["unit"][0] += r.size; stats["unit"][1] += int(dec.sum()); stats["unit"][2] += int((lane[dec] != chain[dec]).sum())
        # ---- weighted rows: base values, common neighbours / prev differ by same-signed deltas (sign of q - 1)
        q = rng.uniform(0.05, 20.0)
        w = (rng.random(n) * 0.999 + 0.001).astype(np.float32)
        base = (w.astype(np.float64) / q).astype(np.float32)
        alpha = 1.0 / q + (1.0 - 1.0 / q) * rng.random(n)
        step = np.where(rng.random(n) < 0.5, w, (w.astype(np.float64) * alpha).astype(np.float32)).astype(np.float32)
        vals = np.where(cls == 1, step, np.where(cls == 2, (w.astype(np.float64) / rng.uniform(0.05, 20.0)).astype(np.float32), base)).astype(np.float32)
        c = chain32(vals)
        r = targets(rng, c)
        chain, lane = (np.empty(r.size, dtype=np.uint32) for _ in range(2))
        _lib.check(lib.pw_selftest_lane_weighted(vals.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p), cls.ctypes.data_as(C.c_void_p), n,
                                                 r.ctypes.data_as(C.c_void_p), r.size, chain.ctypes.data_as(C.c_void_p), lane.ctypes.data_as(C.c_void_p)))
        dec = lane != AMB
        stats["weighted"][0] += r.size; stats["weighted"][1] += int(dec.sum()); stats["weighted"][2] += int((lane[dec] != chain[dec]).sum())
        if stats["unit"][2] or stats["weighted"][2]:
            print("WRONG DECISION", n, float(w_out), float(w_prev), q, stats)
            sys.exit(1)
    print(f"fuzz_bounded: {time.time() - t0:.0f} s, seed {seed}: " + ", ".join(f"{k}: {v[0]} verdicts, {v[1]} decided, {v[2]} wrong" for k, v in stats.items()))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 600.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
