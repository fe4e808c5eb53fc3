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


# ---- the lane kernel's per-thread form of the same decision (seqscan.h: lane_decide, lane_tight, lane_chain) -----
# Every check below runs twice: on the host build of the routines, and (marked gpu) with one DEVICE thread per target,
# i.e. through the code the walk kernels execute -- device code generation, v_rcp_f32 in lane_tight, -ffp-contract=off.
LANE_AMBIGUOUS = 0xFFFFFFFD
LANE_REDO = 0xFFFFFFFC
LANE_CHAIN_END = 0xFFFFFFFB
LANE_TIE = 0xFFFFFFFA

BACKENDS = [pytest.param(False, id="host"), pytest.param(True, id="device", marks=pytest.mark.gpu)]


class LaneRun:
    def __init__(self, cls, w_out, w_prev, r, device):
        lib = _lib.load()
        cls = np.ascontiguousarray(cls, dtype=np.uint8)
        r = np.ascontiguousarray(r, dtype=np.float64)
        self.chain, self.lane, self.kmax, self.tight, self.chain_lane = (np.empty(r.size, dtype=np.uint32) for _ in range(5))
        _lib.check(lib.pw_selftest_lane(int(device), 0, cls.ctypes.data_as(C.c_void_p), cls.size, w_out, w_prev,
                                        r.ctypes.data_as(C.c_void_p), r.size, *(a.ctypes.data_as(C.c_void_p) for a in
                                        (self.chain, self.lane, self.kmax, self.tight, self.chain_lane))))


def adversarial_targets(rng, cls, w_out, w_prev, n_uniform, stride=1):
    """Uniform draws + targets exactly on / one ulp around every float32 partial sum and every exact partial sum."""
    c32, exact_cdf = float32_prefix(cls, w_out, w_prev)
    cd = c32.astype(np.float64)
    sub = slice(None, None, stride)
    targets = [rng.random(n_uniform), cd[sub], np.nextafter(cd[sub], 0.0), np.nextafter(cd[sub], 2.0), exact_cdf[sub],
               np.nextafter(exact_cdf[sub], 0.0), np.nextafter(exact_cdf[sub], 2.0),
               np.array([0.0, 1e-300, 1 - 2.0 ** -53, 0.5, 0.25, 0.125, np.nextafter(0.5, 0.0), np.nextafter(0.25, 0.0)])]
    return np.clip(np.concatenate(targets), 0.0, np.nextafter(1.0, 0.0)), cd


@pytest.mark.parametrize("device", BACKENDS)
@pytest.mark.parametrize("w_out,w_prev", BIASES)
def test_lane_decision_from_common_neighbour_positions(w_out, w_prev, device):
    """One thread, bisection over the positions of the common neighbours + closed-form "out" runs: decided
    indices equal the float32 chain; undecided ones need no more than kmax leading positions; and the
    decision agrees with the wave kernel's rank-search form on which targets are decided."""
    rng = np.random.default_rng(int(w_out * 64 + w_prev * 1024) + 7)
    decided = total = ties = chained = 0
    for n in (1, 2, 3, 7, 40, 64, 65, 300, 1500, 6000):
        for p_common in (0.0, 0.02, 0.3, 0.9, 1.0):
            for with_prev in (False, True):
                cls = random_row(rng, n, p_common, with_prev)
                r, cd = adversarial_targets(rng, cls, w_out, w_prev, 300)
                run = LaneRun(cls, w_out, w_prev, r, device)
                chain, lane, kmax = run.chain, run.lane, run.kmax
                assert np.array_equal(chain, np.searchsorted(cd, r, side="left").astype(np.uint32))   # (the hook's own chain)
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
                cl_ = run.chain_lane
                tie = cl_ == LANE_TIE
                end = cl_ == LANE_CHAIN_END
                assert np.array_equal(cl_[~tie & ~end], chain[~tie & ~end]), (n, p_common, with_prev, w_out, w_prev)
                assert (chain[end] == n).all()
                # the interval decision: whatever it settles is the chain's answer
                settled = amb & (run.tight != LANE_AMBIGUOUS)
                assert np.array_equal(run.tight[ok], lane[ok]) and np.array_equal(run.tight[settled], chain[settled])
                ties += int(tie.sum())
                chained += int(tie.size)
                decided += int(ok[:300].sum())
                total += 300
    assert decided / total > 0.5
    assert ties / chained < 0.2          # the per-thread chain rarely has to decline


@pytest.mark.parametrize("device", BACKENDS)
def test_lane_decision_first_step_row(device):
    """First step of a walk: no prev, every neighbour weighs 1 (w_out passed as 1.0).  70000 entries: uint32 positions."""
    rng = np.random.default_rng(11)
    for n in (1, 5, 64, 1000, 70000):
        cls = np.zeros(n, dtype=np.uint8)
        r = np.concatenate([rng.random(2000), (np.arange(1, min(n, 500) + 1) / n)])
        r = np.clip(r, 0.0, np.nextafter(1.0, 0.0))
        run = LaneRun(cls, 1.0, 2.0, r, device)
        ok = run.lane != LANE_AMBIGUOUS
        assert np.array_equal(run.lane[ok], run.chain[ok])
        assert ((run.chain[~ok] < run.kmax[~ok]) | (run.chain[~ok] == n)).all()


def test_lane_decision_row_outside_exact_range_is_redone():
    cls = np.zeros(40, dtype=np.uint8)
    cls[3] = 2
    run = LaneRun(cls, 2.0 ** -30, 1.0, np.array([0.3]), False)   # total 39 * 2^-30 + 1 is not a float32
    assert run.lane[0] == LANE_REDO


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


def lattice_rows(n):
    """Rows as a 2^k-regular ring lattice / clique-rich graph produces them: power-of-two degrees, the common
    neighbours in contiguous blocks around prev, prev in the middle -- totals that are powers of two or one unit
    off, where float32 values tie in every binade."""
    rows = []
    for frac in (0.25, 0.5, 0.75):
        m = int(n * frac)
        row = np.zeros(n, dtype=np.uint8)
        row[(n - m) // 2:(n - m) // 2 + m] = 1
        row[n // 2] = 2
        rows.append(row)
        alt = np.zeros(n, dtype=np.uint8)
        alt[::2] = 1                                                 # every second neighbour common
        alt[n // 2 + 1 if n > 2 else 0] = 2
        rows.append(alt)
    return rows


@pytest.mark.parametrize("device", BACKENDS)
@pytest.mark.parametrize("w_out,w_prev", BIASES)
def test_interval_decision_of_ambiguous_steps(w_out, w_prev, device):
    """lane_tight bounds the float32 chain's systematic drift from the class counts alone (no list access): whatever it
    decides must be the chain's answer -- uniform targets, and targets exactly on / one ulp around every float32
    partial sum and every exact partial sum, where an interval that is one ulp too narrow flips the answer."""
    rng = np.random.default_rng(int(w_out * 64 + w_prev * 1024) + 321)
    amb_u = res_u = 0
    for n in (3, 40, 65, 300, 1500, 4096, 6000, 20000, 70000):
        for cls in structured_rows(rng, n) + (lattice_rows(n) if n in (4096, 65, 1500) else []):
            r, _ = adversarial_targets(rng, cls, w_out, w_prev, 1000, stride=max(1, n // 500))
            run = LaneRun(cls, w_out, w_prev, r, device)
            chain, lane, tight = run.chain, run.lane, run.tight
            amb = lane == LANE_AMBIGUOUS
            assert np.array_equal(tight[~amb], lane[~amb])
            settled = amb & (tight != LANE_AMBIGUOUS)
            assert np.array_equal(tight[settled], chain[settled]), (n, w_out, w_prev, np.flatnonzero(settled & (tight != chain))[:5])
            cl_ = run.chain_lane
            good = (cl_ != LANE_TIE) & (cl_ != LANE_CHAIN_END)
            assert np.array_equal(cl_[good], chain[good])
            amb_u += int(amb[:1000].sum())
            res_u += int(settled[:1000].sum())
    assert amb_u == 0 or res_u / amb_u > 0.15   # (rows here are dense in common neighbours; hub rows of real graphs: ~0.9)


@pytest.mark.parametrize("device", BACKENDS)
def test_interval_decision_power_of_two_totals(device):
    """Row totals 2^k: every value is a power of two, every addition exact -- the interval collapses to a point."""
    rng = np.random.default_rng(5)
    for n in (1024, 4096, 32768):
        cls = np.zeros(n, dtype=np.uint8)           # first step of a walk: all weights 1, total n
        r = np.clip(np.concatenate([rng.random(3000), np.arange(1, 400) / n, np.nextafter(np.arange(1, 400) / n, 0.0)]), 0.0,
                    np.nextafter(1.0, 0.0))
        run = LaneRun(cls, 1.0, 2.0, r, device)
        ok = run.tight != LANE_AMBIGUOUS
        assert np.array_equal(run.tight[ok], run.chain[ok])


# ---- the FLOATS form of a lane-kernel step: 1/p or 1/q not a power of two (arbitrary float32 row values) ----------
FLOAT_BIASES = [(1 / 1.7, 1 / 0.3), (1 / 0.37, 1 / 3.0), (0.9, 1.1), (1 / 3.0, 1.0), (7.3, 0.01), (1 / 1.3, 1 / 0.4)]


def float_chain_reference(cls, w_out, w_prev):
    """float32 row total and partial sums of w / tot, sequential (Numba's arr.sum() and np.cumsum)."""
    w = np.where(cls == 1, np.float32(1.0), np.where(cls == 0, np.float32(w_out), np.float32(w_prev))).astype(np.float32)
    tot = np.float32(0)
    for v in w:
        tot = np.float32(tot + v)
    x = (w / tot).astype(np.float32)
    c = np.zeros(cls.size, dtype=np.float32)
    acc = np.float32(0)
    for k, v in enumerate(x):
        acc = np.float32(acc + v)
        c[k] = acc
    return tot, c


@pytest.mark.parametrize("device", BACKENDS)
@pytest.mark.parametrize("w_out,w_prev", FLOAT_BIASES)
def test_float_chain_step_for_non_dyadic_biases(w_out, w_prev, device):
    """Two closed-form chains of one thread (row total, then the search over w / tot) equal the sequential float32
    loops for arbitrary positive float32 biases -- uniform draws and draws on / one ulp around every partial sum."""
    lib = _lib.load()
    w_out, w_prev = float(np.float32(w_out)), float(np.float32(w_prev))
    rng = np.random.default_rng(int(w_out * 977 + w_prev * 131) + 5)
    declined = total = 0
    rows = []
    for n in (1, 2, 5, 33, 64, 65, 400, 3000, 20000, 70000):
        for p_common in (0.0, 0.03, 0.4, 1.0):
            for with_prev in (False, True):
                rows.append(random_row(rng, n, p_common, with_prev))
    rows += lattice_rows(4096) + structured_rows(rng, 1500)
    for cls in rows:
        n = cls.size
        tot, c32 = float_chain_reference(cls, w_out, w_prev)
        cd = c32.astype(np.float64)
        sub = slice(None, None, max(1, n // 400))
        r = np.clip(np.concatenate([rng.random(400), cd[sub], np.nextafter(cd[sub], 0.0), np.nextafter(cd[sub], 2.0),
                                    np.array([0.0, 1e-300, 1 - 2.0 ** -53])]), 0.0, np.nextafter(1.0, 0.0))
        chain, lane = (np.empty(r.size, dtype=np.uint32) for _ in range(2))
        tots = np.empty(2 * r.size, dtype=np.float32)
        _lib.check(lib.pw_selftest_lane_floats(int(device), 0, np.ascontiguousarray(cls).ctypes.data_as(C.c_void_p), n, w_out, w_prev,
                                               r.ctypes.data_as(C.c_void_p), r.size, chain.ctypes.data_as(C.c_void_p),
                                               lane.ctypes.data_as(C.c_void_p), tots.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(chain, np.searchsorted(cd, r, side="left").astype(np.uint32))       # the hook's own reference
        assert (tots[0::2] == tot).all()
        tie = lane == LANE_TIE
        assert (tots[1::2][~tie] == tot).all(), (n, w_out, w_prev)                                  # row total, bit for bit
        end = lane == LANE_CHAIN_END
        assert np.array_equal(lane[~tie & ~end], chain[~tie & ~end]), (n, w_out, w_prev)
        assert (chain[end] == n).all()
        declined += int(tie.sum())
        total += r.size
    assert declined / total < 0.05


@pytest.mark.parametrize("w_out,w_prev", FLOAT_BIASES + [(1.0, 1.0)])
def test_bounded_decision_of_the_float_step(w_out, w_prev):
    """Round 5: the FLOATS step decides from the closed-form real prefix sums of the three row values with a rigorous bound on
    the float32 chain (lane_decide_unit_bounded).  Every verdict it gives -- uniform draws, draws exactly on, one ulp and
    1e-7 around every partial sum of the float32 chain, the first / last representable draws -- equals the sequential loops;
    k_safe never passes the reference's position (checked in the hook); and the bound is not vacuous: it settles most draws
    of short rows and leaves open what sits on a partial sum."""
    lib = _lib.load()
    w_out, w_prev = float(np.float32(w_out)), float(np.float32(w_prev))
    rng = np.random.default_rng(int(w_out * 877 + w_prev * 31) + 9)
    rows = []
    for n in (1, 2, 5, 33, 64, 65, 400, 3000, 20000, 70000):
        for p_common in (0.0, 0.03, 0.4, 1.0):
            for with_prev in (False, True):
                rows.append(random_row(rng, n, p_common, with_prev))
    rows += lattice_rows(4096) + structured_rows(rng, 1500)
    settled_uniform = total_uniform = open_on_sum = total_on_sum = 0
    for cls in rows:
        n = cls.size
        tot, c32 = float_chain_reference(cls, w_out, w_prev)
        cd = c32.astype(np.float64)
        sub = slice(None, None, max(1, n // 300))
        uni = rng.random(300)
        on = cd[sub]
        r = np.clip(np.concatenate([uni, on, np.nextafter(on, 0.0), np.nextafter(on, 2.0), on * (1 - 1e-7), on * (1 + 1e-7),
                                    np.array([0.0, 1e-300, 1 - 2.0 ** -53])]), 0.0, np.nextafter(1.0, 0.0))
        chain, lane = (np.empty(r.size, dtype=np.uint32) for _ in range(2))
        _lib.check(lib.pw_selftest_lane_unit_bounded(np.ascontiguousarray(cls).ctypes.data_as(C.c_void_p), n, w_out, w_prev,
                                                     r.ctypes.data_as(C.c_void_p), r.size, chain.ctypes.data_as(C.c_void_p),
                                                     lane.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(chain, np.searchsorted(cd, r, side="left").astype(np.uint32))       # the hook's own reference
        decided = lane != LANE_AMBIGUOUS
        assert np.array_equal(lane[decided], chain[decided]), (n, w_out, w_prev, np.flatnonzero(lane[decided] != chain[decided])[:5])
        settled_uniform += int(decided[:300].sum()); total_uniform += 300
        open_on_sum += int((~decided[300:300 + on.size]).sum()); total_on_sum += on.size
    assert settled_uniform / total_uniform > 0.6, (settled_uniform, total_uniform)
    assert open_on_sum / total_on_sum > 0.9, (open_on_sum, total_on_sum)      # a draw ON a partial sum cannot be settled by a bound


@pytest.mark.parametrize("w_out,w_prev", FLOAT_BIASES + [(1.0, 1.0)])
def test_interval_decision_of_the_float_step(w_out, w_prev):
    """Round 6: in front of its float chains the FLOATS step runs the interval decision for ARBITRARY float32 values
    (lane_tight_values: the chain's systematic drift bounded from the class counts the bounded decision has -- the routine
    the dyadic form uses, fed with the real sum of the values instead of an integer mass).  Every position it gives
    equals the sequential loops -- uniform draws, draws exactly on, one ulp and 1e-7 around every partial sum -- and it is
    not vacuous: of the uniform draws the bound leaves open on long rows it settles most."""
    lib = _lib.load()
    w_out, w_prev = float(np.float32(w_out)), float(np.float32(w_prev))
    rng = np.random.default_rng(int(w_out * 577 + w_prev * 37) + 3)
    rows = []
    for n in (2, 5, 33, 64, 65, 400, 3000, 20000, 70000):
        for p_common in (0.0, 0.01, 0.03, 0.4, 1.0):
            for with_prev in (False, True):
                rows.append(random_row(rng, n, p_common, with_prev))
    rows += lattice_rows(4096) + structured_rows(rng, 1500)
    open_uni = settled_uni = 0
    for cls in rows:
        n = cls.size
        tot, c32 = float_chain_reference(cls, w_out, w_prev)
        cd = c32.astype(np.float64)
        sub = slice(None, None, max(1, n // 300))
        uni = rng.random(600)
        on = cd[sub]
        r = np.clip(np.concatenate([uni, on, np.nextafter(on, 0.0), np.nextafter(on, 2.0), on * (1 - 1e-7), on * (1 + 1e-7),
                                    on * (1 - 3e-6), on * (1 + 3e-6), np.array([0.0, 1e-300, 1 - 2.0 ** -53])]), 0.0, np.nextafter(1.0, 0.0))
        chain, lane, tight = (np.empty(r.size, dtype=np.uint32) for _ in range(3))
        _lib.check(lib.pw_selftest_lane_unit_tight(np.ascontiguousarray(cls).ctypes.data_as(C.c_void_p), n, w_out, w_prev,
                                                   r.ctypes.data_as(C.c_void_p), r.size, chain.ctypes.data_as(C.c_void_p),
                                                   lane.ctypes.data_as(C.c_void_p), tight.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(chain, np.searchsorted(cd, r, side="left").astype(np.uint32))       # the hook's own reference
        decided = tight != LANE_AMBIGUOUS
        assert np.array_equal(tight[decided], chain[decided]), (n, w_out, w_prev, np.flatnonzero(tight[decided] != chain[decided])[:5])
        assert np.array_equal(tight[lane != LANE_AMBIGUOUS], lane[lane != LANE_AMBIGUOUS])          # (a verdict of the bound stands)
        if n >= 3000:
            amb = lane[:600] == LANE_AMBIGUOUS
            open_uni += int(amb.sum()); settled_uni += int((amb & decided[:600]).sum())
    assert open_uni > 500 and settled_uni / open_uni > 0.3, (settled_uni, open_uni)   # (rows full of common neighbours are the hard ones)


# ---- weighted rows: float64 prefix sums + a rigorous bound on the float32 chain (lane_decide_weighted) ------------------
def _weighted_run(vals, base, cls, r):
    vals = np.ascontiguousarray(vals, np.float32)
    base = np.ascontiguousarray(base, np.float32)
    cls = np.ascontiguousarray(cls, np.uint8)
    r = np.ascontiguousarray(r, np.float64)
    chain = np.zeros(r.size, np.uint32)
    lane = np.zeros(r.size, np.uint32)
    _lib.check(_lib.load().pw_selftest_lane_weighted(
        vals.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p), cls.ctypes.data_as(C.c_void_p), vals.size,
        r.ctypes.data_as(C.c_void_p), r.size, chain.ctypes.data_as(C.c_void_p), lane.ctypes.data_as(C.c_void_p)))
    return chain, lane


@pytest.mark.parametrize("n", [1, 3, 17, 64, 300, 1000, 3000, 9000, 70000])
def test_weighted_lane_decision_never_disagrees_with_the_float32_chain(n):
    """Every target the decision SETTLES must be the position the reference's sequential float32 cumsum / searchsorted
    gives -- random targets and targets placed exactly on, one ulp around and 1e-7 around partial sums of the chain --
    for node2vec and node2vec+-like rows (commons with arbitrary deltas), dyadic and non-dyadic p, q; what it leaves
    open is reported, not decided.  Rows beyond a few thousand entries are mostly ambiguous (the bound grows with k)."""
    rs = np.random.RandomState(n)
    for trial in range(6):
        w = (rs.random_sample(n) * 0.999 + 0.001).astype(np.float32)
        if trial == 5:
            w = np.ldexp(np.float32(1.0), rs.randint(-6, 3, n)).astype(np.float32)      # dyadic weights: tie-heavy chains
        q = float(rs.choice([2.0, 0.5, 1.7, 0.3, 4.0, 1.0]))
        p = float(rs.choice([0.5, 2.0, 0.3, 1.0]))
        cls = np.zeros(n, np.uint8)
        cls[rs.choice(n, int(rs.randint(0, max(1, n // 3))), replace=False)] = 1
        cand = np.flatnonzero(cls == 0)
        if cand.size and rs.random_sample() < 0.8:
            cls[rs.choice(cand)] = 2
        base = (w.astype(np.float64) / q).astype(np.float32)
        vals = base.copy()
        com = cls == 1
        if trial % 2:                      # node2vec+-like: a common neighbour is an in-edge (w) or an out-edge (w * alpha)
            alpha = 1.0 / q + (1.0 - 1.0 / q) * rs.random_sample(n)
            ext = (w.astype(np.float64) * alpha).astype(np.float32)
            vals[com] = np.where(rs.random_sample(int(com.sum())) < 0.5, w[com], ext[com])
        else:
            vals[com] = w[com]
        vals[cls == 2] = (w[cls == 2].astype(np.float64) / p).astype(np.float32)
        t = np.float32(0)
        for x in vals:
            t = np.float32(t + x)
        c = np.cumsum((vals / t).astype(np.float32), dtype=np.float32).astype(np.float64)
        pick = rs.choice(n, min(n, 150), replace=False)
        r = np.concatenate([rs.random_sample(300), c[pick], np.nextafter(c[pick], 0), np.nextafter(c[pick], 2),
                            c[pick] * (1 - 1e-7), c[pick] * (1 + 1e-7), [0.0, 0.9999999999]])
        r = np.clip(r, 0, 0.9999999999)
        chain, lane = _weighted_run(vals, base, cls, r)
        decided = lane != 0xFFFFFFFD
        assert np.array_equal(lane[decided], chain[decided]), (n, p, q, trial)
        if n <= 300:
            assert decided[:300].mean() > 0.9          # short rows: the bound is far below the spacing of the partial sums
