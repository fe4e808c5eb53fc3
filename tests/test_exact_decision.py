"""The mathematical claim behind the lazy walk step, tested on the host without a GPU: whenever the
exact-arithmetic decision (csrc/seqscan.h: exact_thresholds_f32, the very function the kernel calls)
declares a CDF search decided, its index equals what the reference's sequential float32
cumsum + searchsorted returns.  Targets are random and adversarial (exactly on / one ulp around the
float32 partial sums and the exact rational partial sums)."""
import ctypes as C

import numpy as np
import pytest

from pecanpy_amd import _lib

AMBIGUOUS = 0xFFFFFFFF


def decide(cls, w_out, w_prev, r):
    lib = _lib.load()
    cls = np.ascontiguousarray(cls, dtype=np.uint8)
    r = np.ascontiguousarray(r, dtype=np.float64)
    chain = np.empty(r.size, dtype=np.uint32)
    exact = np.empty(r.size, dtype=np.uint32)
    _lib.check(lib.pw_selftest_exact_decision(cls.ctypes.data_as(C.c_void_p), cls.size, w_out, w_prev,
                                              r.ctypes.data_as(C.c_void_p), r.size,
                                              chain.ctypes.data_as(C.c_void_p), exact.ctypes.data_as(C.c_void_p)))
    return chain, exact


def float32_prefix(cls, w_out, w_prev):
    """float32 partial sums of the reference (sequential) and the exact rational ones."""
    w = np.where(cls == 1, 1.0, np.where(cls == 0, w_out, w_prev))
    tot = np.float32(w.sum())                      # exact by construction (dyadic, small)
    x = (w.astype(np.float32) / tot).astype(np.float32)
    c = np.zeros(cls.size, dtype=np.float32)
    acc = np.float32(0)
    for k, v in enumerate(x):
        acc = np.float32(acc + v)
        c[k] = acc
    return c, np.cumsum(w) / w.sum()


def random_row(rng, n, p_common, with_prev):
    cls = (rng.random(n) < p_common).astype(np.uint8)
    if with_prev:
        cls[rng.integers(0, n)] = 2
    return cls


BIASES = [(0.5, 2.0), (2.0, 0.5), (0.25, 4.0), (1.0, 1.0), (4.0, 0.125), (0.0625, 16.0), (8.0, 1.0)]


@pytest.mark.parametrize("w_out,w_prev", BIASES)
def test_decided_cases_agree_with_the_float32_chain(w_out, w_prev):
    rng = np.random.default_rng(int(w_out * 64 + w_prev * 1024))
    decided = total = 0
    for n in (1, 2, 3, 7, 40, 64, 65, 300, 1500, 6000):
        for p_common in (0.0, 0.05, 0.4, 1.0):
            cls = random_row(rng, n, p_common, with_prev=bool(rng.integers(0, 2)))
            c32, exact_cdf = float32_prefix(cls, w_out, w_prev)
            cd = c32.astype(np.float64)
            targets = [rng.random(400), cd, np.nextafter(cd, 0.0), np.nextafter(cd, 2.0),
                       exact_cdf, np.nextafter(exact_cdf, 0.0), np.nextafter(exact_cdf, 2.0),
                       exact_cdf * (1 - 2.0 ** -24), exact_cdf * (1 + 2.0 ** -24), np.array([0.0, 1e-300, 1 - 2.0 ** -53])]
            r = np.clip(np.concatenate(targets), 0.0, np.nextafter(1.0, 0.0))
            chain, exact = decide(cls, w_out, w_prev, r)
            # the hook's chain is the reference semantics: cross-check it with NumPy once per row
            want = np.searchsorted(cd, r, side="left")
            assert np.array_equal(chain, want.astype(np.uint32))
            ok = exact != AMBIGUOUS
            assert np.array_equal(exact[ok], chain[ok]), (n, p_common, w_out, w_prev)
            decided += int(ok[:400].sum())
            total += 400
    assert decided / total > 0.5          # the shortcut is not vacuous on uniform targets


def test_short_rows_are_almost_always_decided():
    rng = np.random.default_rng(5)
    cls = random_row(rng, 200, 0.1, True)
    chain, exact = decide(cls, 0.5, 2.0, rng.random(20000))
    ok = exact != AMBIGUOUS
    assert ok.mean() > 0.98 and np.array_equal(exact[ok], chain[ok])


def test_rejects_non_dyadic_biases():
    lib = _lib.load()
    cls = np.zeros(4, dtype=np.uint8)
    r = np.array([0.5])
    out = np.empty(1, dtype=np.uint32)
    rc = lib.pw_selftest_exact_decision(cls.ctypes.data_as(C.c_void_p), 4, 0.3, 2.0, r.ctypes.data_as(C.c_void_p), 1,
                                        out.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc != 0


def decide64(cls, w_out, w_prev, r):
    lib = _lib.load()
    cls = np.ascontiguousarray(cls, dtype=np.uint8)
    r = np.ascontiguousarray(r, dtype=np.float64)
    chain = np.empty(r.size, dtype=np.uint32)
    exact = np.empty(r.size, dtype=np.uint32)
    _lib.check(lib.pw_selftest_exact_decision_f64(cls.ctypes.data_as(C.c_void_p), cls.size, w_out, w_prev,
                                                  r.ctypes.data_as(C.c_void_p), r.size,
                                                  chain.ctypes.data_as(C.c_void_p), exact.ctypes.data_as(C.c_void_p)))
    return chain, exact


@pytest.mark.parametrize("w_out,w_prev", BIASES)
def test_float64_flavour_of_the_dense_kernel(w_out, w_prev):
    rng = np.random.default_rng(int(w_out * 64 + w_prev * 1024) + 7)
    for n in (1, 3, 64, 1000, 30000):
        cls = random_row(rng, n, 0.25, True)
        w = np.where(cls == 1, 1.0, np.where(cls == 0, w_out, w_prev))
        tot = w.sum()
        c64 = np.empty(n)
        acc = 0.0
        for k, v in enumerate(w / tot):          # sequential float64, like Numba's cumsum
            acc = acc + v
            c64[k] = acc
        exact_cdf = np.cumsum(w) / tot
        r = np.clip(np.concatenate([rng.random(300), c64, np.nextafter(c64, 0.0), np.nextafter(c64, 2.0), exact_cdf,
                                    np.nextafter(exact_cdf, 0.0), np.nextafter(exact_cdf, 2.0)]), 0.0, np.nextafter(1.0, 0.0))
        chain, exact = decide64(cls, w_out, w_prev, r)
        assert np.array_equal(chain, np.searchsorted(c64, r, side="left").astype(np.uint32))
        ok = exact != AMBIGUOUS
        assert np.array_equal(exact[ok], chain[ok]), (n, w_out, w_prev)
        assert ok[:300].mean() > 0.99             # float64 drift is far below one unit: practically always decided


# ---- the lane kernel's per-thread form of the same decision (seqscan.h: lane_decide) ---------------------------
LANE_AMBIGUOUS = 0xFFFFFFFD
LANE_REDO = 0xFFFFFFFC
LANE_CHAIN_END = 0xFFFFFFFB
LANE_TIE = 0xFFFFFFFA


def lane_decide(cls, w_out, w_prev, r, use_hints=False):
    lib = _lib.load()
    cls = np.ascontiguousarray(cls, dtype=np.uint8)
    r = np.ascontiguousarray(r, dtype=np.float64)
    chain = np.empty(r.size, dtype=np.uint32)
    lane = np.empty(r.size, dtype=np.uint32)
    kmax = np.empty(r.size, dtype=np.uint32)
    chain_lane = np.empty(r.size, dtype=np.uint32)
    probes = np.empty(r.size, dtype=np.uint32)
    refined = np.empty(r.size, dtype=np.uint32)
    _lib.check(lib.pw_selftest_lane_decide(cls.ctypes.data_as(C.c_void_p), cls.size, w_out, w_prev,
                                           r.ctypes.data_as(C.c_void_p), r.size, chain.ctypes.data_as(C.c_void_p),
                                           lane.ctypes.data_as(C.c_void_p), kmax.ctypes.data_as(C.c_void_p),
                                           chain_lane.ctypes.data_as(C.c_void_p), int(use_hints),
                                           probes.ctypes.data_as(C.c_void_p), refined.ctypes.data_as(C.c_void_p)))
    lane_decide.refined = refined
    lane_decide.chain_lane = chain_lane
    lane_decide.probes = probes
    return chain, lane, kmax


@pytest.mark.parametrize("w_out,w_prev", BIASES)
def test_lane_decision_from_common_neighbour_positions(w_out, w_prev):
    """One thread, bisection over the positions of the common neighbours + closed-form "out" runs: decided
    indices equal the float32 chain; undecided ones need no more than kmax leading positions; and the
    decision agrees with the wave kernel's rank-search form on which targets are decided."""
    rng = np.random.default_rng(int(w_out * 64 + w_prev * 1024) + 7)
    decided = total = ties = chained = reads_plain = reads_hint = 0
    for n in (1, 2, 3, 7, 40, 64, 65, 300, 1500, 6000):
        for p_common in (0.0, 0.02, 0.3, 0.9, 1.0):
            for with_prev in (False, True):
                cls = random_row(rng, n, p_common, with_prev)
                c32, exact_cdf = float32_prefix(cls, w_out, w_prev)
                cd = c32.astype(np.float64)
                targets = [rng.random(300), cd, np.nextafter(cd, 0.0), np.nextafter(cd, 2.0), exact_cdf,
                           np.nextafter(exact_cdf, 0.0), np.nextafter(exact_cdf, 2.0),
                           np.array([0.0, 1e-300, 1 - 2.0 ** -53])]
                r = np.clip(np.concatenate(targets), 0.0, np.nextafter(1.0, 0.0))
                chain, lane, kmax = lane_decide(cls, w_out, w_prev, r)
                assert not (lane == LANE_REDO).any()
                ok = lane != LANE_AMBIGUOUS
                assert np.array_equal(lane[ok], chain[ok]), (n, p_common, with_prev, w_out, w_prev)
                amb = ~ok
                assert ((chain[amb] < kmax[amb]) | (chain[amb] == n)).all()
                assert (kmax[amb] <= n).all()
                _, exact = decide(cls, w_out, w_prev, r)
                assert np.array_equal(exact != AMBIGUOUS, ok)      # same bound, same verdicts
                # the per-thread float chain (lane_chain): over the ambiguous prefix / the whole row it returns
                # the reference's index, "never reached", or declines on a rounding tie
                cl_ = lane_decide.chain_lane
                tie = cl_ == LANE_TIE
                end = cl_ == LANE_CHAIN_END
                assert np.array_equal(cl_[~tie & ~end], chain[~tie & ~end]), (n, p_common, with_prev, w_out, w_prev)
                assert (chain[end] == n).all()
                ties += int(tie.sum())
                chained += int(tie.size)
                # guided by the hint table: identical answers from fewer list reads
                plain_reads = int(lane_decide.probes[:300].sum())
                chain_h, lane_h, kmax_h = lane_decide(cls, w_out, w_prev, r, use_hints=True)
                assert np.array_equal(lane_h, lane) and np.array_equal(kmax_h, kmax)
                assert np.array_equal(lane_decide.chain_lane, cl_)
                reads_plain += plain_reads
                reads_hint += int(lane_decide.probes[:300].sum())
                decided += int(ok[:300].sum())
                total += 300
    assert decided / total > 0.5
    assert ties / chained < 0.2          # the per-thread chain rarely has to decline
    assert reads_hint < reads_plain      # (entries read; a 4-entry window counts 4 but is ONE access)


def test_lane_decision_first_step_row():
    """First step of a walk: no prev, every neighbour weighs 1 (w_out passed as 1.0)."""
    rng = np.random.default_rng(11)
    for n in (1, 5, 64, 1000, 70000):
        cls = np.zeros(n, dtype=np.uint8)
        r = np.concatenate([rng.random(2000), (np.arange(1, min(n, 500) + 1) / n)])
        r = np.clip(r, 0.0, np.nextafter(1.0, 0.0))
        chain, lane, kmax = lane_decide(cls, 1.0, 2.0, r)
        ok = lane != LANE_AMBIGUOUS
        assert np.array_equal(lane[ok], chain[ok])
        assert ((chain[~ok] < kmax[~ok]) | (chain[~ok] == n)).all()


def test_lane_decision_row_outside_exact_range_is_redone():
    cls = np.zeros(40, dtype=np.uint8)
    cls[3] = 2
    _, lane, _ = lane_decide(cls, 2.0 ** -30, 1.0, np.array([0.3]))   # total 39 * 2^-30 + 1 is not a float32
    assert lane[0] == LANE_REDO


@pytest.mark.parametrize("w_out,w_prev", BIASES)
def test_refined_decision_of_ambiguous_steps(w_out, w_prev):
    """lane_refine computes the drift of the float32 chain from per-binade class counts instead of bounding it: whatever
    it decides must be the chain's answer -- on uniform targets and on targets placed exactly on / one ulp around
    every float32 partial sum, where a drift estimate that is off by one ulp flips the answer -- and it must settle
    most of the steps the a-priori bound leaves open."""
    rng = np.random.default_rng(int(w_out * 64 + w_prev * 1024) + 99)
    amb_u = res_u = 0
    for n in (3, 40, 65, 300, 1500, 6000, 20000):
        for p_common in (0.0, 0.02, 0.3, 0.9):
            for with_prev in (False, True):
                cls = random_row(rng, n, p_common, with_prev)
                c32, exact_cdf = float32_prefix(cls, w_out, w_prev)
                cd = c32.astype(np.float64)
                sub = slice(None, None, max(1, n // 600))
                targets = [rng.random(1500), cd[sub], np.nextafter(cd[sub], 0.0), np.nextafter(cd[sub], 2.0), exact_cdf[sub],
                           np.nextafter(exact_cdf[sub], 0.0), np.nextafter(exact_cdf[sub], 2.0)]
                r = np.clip(np.concatenate(targets), 0.0, np.nextafter(1.0, 0.0))
                chain, lane, _ = lane_decide(cls, w_out, w_prev, r)
                ref = lane_decide.refined
                amb = lane == LANE_AMBIGUOUS
                assert np.array_equal(ref[~amb], lane[~amb])
                settled = amb & (ref != LANE_AMBIGUOUS)
                assert np.array_equal(ref[settled], chain[settled]), (n, p_common, with_prev, w_out, w_prev)
                amb_u += int(amb[:1500].sum())
                res_u += int(settled[:1500].sum())
    assert amb_u == 0 or res_u / amb_u > 0.5


def lane_tight(cls, w_out, w_prev, r):
    lib = _lib.load()
    cls = np.ascontiguousarray(cls, dtype=np.uint8)
    r = np.ascontiguousarray(r, dtype=np.float64)
    chain = np.empty(r.size, dtype=np.uint32)
    lane = np.empty(r.size, dtype=np.uint32)
    tight = np.empty(r.size, dtype=np.uint32)
    _lib.check(lib.pw_selftest_lane_tight(cls.ctypes.data_as(C.c_void_p), cls.size, w_out, w_prev,
                                          r.ctypes.data_as(C.c_void_p), r.size, chain.ctypes.data_as(C.c_void_p),
                                          lane.ctypes.data_as(C.c_void_p), tight.ctypes.data_as(C.c_void_p)))
    return chain, lane, tight


def structured_rows(rng, n):
    """Rows that stress the interval argument: commons clustered at either end / in one binade, prev early and late,
    totals that are powers of two (exact values: no drift at all) and totals whose reciprocal ties in a high binade."""
    rows = []
    for p_common in (0.0, 0.005, 0.02, 0.1, 0.3):
        for with_prev in (False, True):
            rows.append(random_row(rng, n, p_common, with_prev))
    for frac in (0.01, 0.1, 0.5):
        m = max(1, int(n * frac))
        head = np.zeros(n, dtype=np.uint8); head[:m] = 1            # every common neighbour in the low binades
        tail = np.zeros(n, dtype=np.uint8); tail[n - m:] = 1         # ... in the top binade
        mid = np.zeros(n, dtype=np.uint8); mid[n // 2 - m // 2: n // 2 - m // 2 + m] = 1
        rows += [head, tail, mid]
    for row in list(rows[-3:]):
        if n > 4:
            early, late = row.copy(), row.copy()
            early[1] = 2
            late[n - 2] = 2
            rows += [early, late]
    return rows


@pytest.mark.parametrize("w_out,w_prev", BIASES)
def test_interval_decision_of_ambiguous_steps(w_out, w_prev):
    """lane_tight bounds the float32 chain's systematic drift from the class counts alone (no list access): whatever it
    decides must be the chain's answer -- uniform targets, and targets exactly on / one ulp around every float32
    partial sum and every exact partial sum, where an interval that is one ulp too narrow flips the answer."""
    rng = np.random.default_rng(int(w_out * 64 + w_prev * 1024) + 321)
    amb_u = res_u = 0
    for n in (3, 40, 65, 300, 1500, 4096, 6000, 20000, 70000):
        for cls in structured_rows(rng, n):
            c32, exact_cdf = float32_prefix(cls, w_out, w_prev)
            cd = c32.astype(np.float64)
            sub = slice(None, None, max(1, n // 500))
            targets = [rng.random(1000), cd[sub], np.nextafter(cd[sub], 0.0), np.nextafter(cd[sub], 2.0), exact_cdf[sub],
                       np.nextafter(exact_cdf[sub], 0.0), np.nextafter(exact_cdf[sub], 2.0),
                       np.array([0.5, 0.25, 0.125, np.nextafter(0.5, 0.0), np.nextafter(0.25, 0.0)])]
            r = np.clip(np.concatenate(targets), 0.0, np.nextafter(1.0, 0.0))
            chain, lane, tight = lane_tight(cls, w_out, w_prev, r)
            amb = lane == LANE_AMBIGUOUS
            assert np.array_equal(tight[~amb], lane[~amb])
            settled = amb & (tight != LANE_AMBIGUOUS)
            assert np.array_equal(tight[settled], chain[settled]), (n, w_out, w_prev, np.flatnonzero(settled & (tight != chain))[:5])
            amb_u += int(amb[:1000].sum())
            res_u += int(settled[:1000].sum())
    assert amb_u == 0 or res_u / amb_u > 0.15   # (rows here are dense in common neighbours; hub rows of real graphs: ~0.9)


def test_interval_decision_power_of_two_totals():
    """Row totals 2^k: every value is a power of two, every addition exact -- the interval collapses to a point."""
    rng = np.random.default_rng(5)
    for n in (1024, 4096, 32768):
        cls = np.zeros(n, dtype=np.uint8)           # first step of a walk: all weights 1, total n
        r = np.clip(np.concatenate([rng.random(3000), np.arange(1, 400) / n, np.nextafter(np.arange(1, 400) / n, 0.0)]), 0.0,
                    np.nextafter(1.0, 0.0))
        chain, lane, tight = lane_tight(cls, 1.0, 2.0, r)
        ok = tight != LANE_AMBIGUOUS
        assert np.array_equal(tight[ok], chain[ok])
