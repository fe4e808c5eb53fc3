"""Host-side pieces of the random stream and of the exact-sum arithmetic (no GPU needed).

* MT19937 jump-ahead (csrc/mtjump.hpp) against NumPy's legacy RandomState = the stream the
  reference consumes.
* binade scan (csrc/seqscan.h) against sequential float32 / float64 cumsum + searchsorted, the
  semantics of reference pecanpy.py:556-557 under Numba.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as orc
from pecanpy_amd import _lib


def sample(seed, offset, n):
    out = np.zeros(n, dtype=np.float64)
    _lib.check(_lib.load().pw_mt_random_sample(seed, offset, n, C.c_void_p(out.ctypes.data)))
    return out


@pytest.mark.parametrize("seed", [0, 1, 12345, 2**32 - 1])
def test_stream_matches_numpy_legacy(seed):
    ref = np.random.RandomState(seed).random_sample(50000)
    for off in [0, 1, 311, 312, 313, 623, 624, 19999, 40000]:
        assert np.array_equal(sample(seed, off, 16), ref[off:off + 16])


def test_jump_ahead_far_offsets_match_sequential_generation():
    for seed, off in [(7, 10**6), (3, 123456789), (0, 2 * 10**8 + 1)]:
        assert np.array_equal(sample(seed, off, 8), orc.random_sample(seed, off, 8))


def _scan(x, r, use_target, chunk):
    lib = _lib.load()
    idx = C.c_uint32()
    if x.dtype == np.float32:
        s = C.c_float()
        _lib.check(lib.pw_selftest_seqscan_f32(C.c_void_p(x.ctypes.data), x.size, float(r), int(use_target), chunk,
                                               C.byref(idx), C.byref(s)))
        return idx.value, np.float32(s.value)
    s = C.c_double()
    _lib.check(lib.pw_selftest_seqscan_f64(C.c_void_p(x.ctypes.data), x.size, float(r), int(use_target), chunk,
                                           C.byref(idx), C.byref(s)))
    return idx.value, np.float64(s.value)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_binade_scan_equals_sequential_cumsum(dtype):
    rng = np.random.default_rng(0)
    dyadic = np.array([1, 0.5, 2, 0.25, 4], dtype=dtype)
    for trial in range(400):
        d = int(rng.choice([1, 2, 3, 7, 63, 64, 65, 300, 2245, 20000]))
        kind = trial % 3
        if kind == 0:
            w = (rng.random(d) + 1e-3).astype(dtype)
        elif kind == 1:
            w = rng.choice(dyadic, d)          # exact ties (round half to even) are common here
        else:
            w = ((rng.random(d) ** 10) + 1e-7).astype(dtype)
        cs = np.cumsum(w, dtype=dtype)         # sequential, same dtype (Numba np.cumsum / .sum())
        _, tot = _scan(w, 0.0, False, int(rng.choice([1, 64, 256])))
        assert tot.tobytes() == cs[-1].tobytes()
        pr = (w / cs[-1]).astype(dtype)
        cdf = np.cumsum(pr, dtype=dtype)
        for r in list(rng.random(3)) + [0.0, float(cdf[-1]), float(np.nextafter(cdf[-1], 2, dtype=dtype))]:
            want = int(np.searchsorted(cdf, r, side="left"))
            got, _ = _scan(pr, r, True, int(rng.choice([5, 64, 256])))
            assert got == want, (d, kind, r)
