// seqscan.h -- bit-exact *parallel* evaluation of a sequential floating-point running sum.
//
// The reference samples the next vertex with  cdf = np.cumsum(probs); np.searchsorted(cdf, r)
// (src/pecanpy/pecanpy.py:556-557) after  probs = w / w.sum()  (src/pecanpy/rw/sparse_rw.py:89);
// under Numba both reductions are naive left-to-right loops in the array dtype, so the sampled
// index depends on the rounding of every partial sum.  A wavefront prefix-scan of floats rounds
// differently.  This header provides the arithmetic that makes a wave-parallel scan reproduce the
// sequential chain bit for bit:
//
//   While the running sum c stays inside one binade [2^e, 2^(e+1)) it is an integer multiple C of
//   ulp = 2^(e-MANT).  Adding x >= 0 with round-to-nearest-even gives  C' = C + inc(x, parity(C)),
//   where inc is the integer rounding of x/ulp (ties resolved by the parity of C + floor(x/ulp)).
//   So inside a binade the chain is an *integer* scan of per-element parity-functions
//   (a0 = increment when C is even, a1 = when C is odd), which is associative and exact.
//   The one element per binade whose sum reaches 2^(e+1) is added with a real floating-point add,
//   and the scan restarts after it in the new binade.
//
// Everything here is plain integer code usable on host (tests / CPU emulation) and device.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PW_HD __host__ __device__ __forceinline__
#define PW_HD_CALL __host__ __device__ __attribute__((noinline))   // a real call: own register allocation
#else
#define PW_HD inline
#define PW_HD_CALL inline
#endif

namespace pw {

template <typename T> struct FloatTraits;

template <> struct FloatTraits<float> {
    using UInt = uint32_t;
    static constexpr int MANT = 23;          // explicit mantissa bits
    static constexpr int EXP_MASK = 0xff;
    static PW_HD UInt bits(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __float_as_uint(x);
#else
        UInt u; memcpy(&u, &x, sizeof(u)); return u;
#endif
    }
    static PW_HD float from_bits(UInt u) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __uint_as_float(u);
#else
        float x; memcpy(&x, &u, sizeof(u)); return x;
#endif
    }
};

template <> struct FloatTraits<double> {
    using UInt = uint64_t;
    static constexpr int MANT = 52;
    static constexpr int EXP_MASK = 0x7ff;
    static PW_HD UInt bits(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return (UInt)__double_as_longlong(x);
#else
        UInt u; memcpy(&u, &x, sizeof(u)); return u;
#endif
    }
    static PW_HD double from_bits(UInt u) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __longlong_as_double((long long)u);
#else
        double x; memcpy(&x, &u, sizeof(u)); return x;
#endif
    }
};

// Parity function: C -> C + (C even ? a0 : a1).  Values saturate at SAT (anything >= the binade
// top 2^(MANT+1) is only ever compared against it, never used numerically).
template <typename T> struct Inc {
    typename FloatTraits<T>::UInt a0, a1;
};

template <typename T> struct Binade {
    using UInt = typename FloatTraits<T>::UInt;
    static constexpr int MANT = FloatTraits<T>::MANT;
    static constexpr UInt ONE = (UInt)1;
    static constexpr UInt TOP = ONE << (MANT + 1);   // C reaches TOP  <=>  sum leaves the binade
    static constexpr UInt SAT = ONE << (MANT + 2);   // saturation value for increments

    // Biased exponent / integer significand of a finite c > 0 (denormals: eb = 1, C < 2^MANT).
    static PW_HD int eb_of(T c) {
        int eb = (int)((FloatTraits<T>::bits(c) >> MANT) & FloatTraits<T>::EXP_MASK);
        return eb == 0 ? 1 : eb;
    }
    static PW_HD UInt sig_of(T c) {
        UInt b = FloatTraits<T>::bits(c);
        UInt frac = b & ((ONE << MANT) - 1);
        int eb = (int)((b >> MANT) & FloatTraits<T>::EXP_MASK);
        return eb == 0 ? frac : (frac | (ONE << MANT));
    }
    // c = C * 2^(eb - bias - MANT) for C < TOP
    static PW_HD T make(UInt C, int eb) {
        if (C < (ONE << MANT)) return FloatTraits<T>::from_bits(C);               // denormal (eb == 1)
        return FloatTraits<T>::from_bits(((UInt)eb << MANT) | (C & ((ONE << MANT) - 1)));
    }

    // Increment function of adding x >= 0 to an accumulator whose biased exponent is eb.
    static PW_HD Inc<T> quantize(T x, int eb) {
        UInt M = sig_of(x);
        int s = eb - eb_of(x);
        Inc<T> r;
        if (M == 0) { r.a0 = r.a1 = 0; return r; }
        if (s <= 0) { r.a0 = r.a1 = SAT; return r; }        // x >= 2^e: the sum certainly leaves the binade
        if (s > MANT + 2) { r.a0 = r.a1 = 0; return r; }    // x < ulp/4: absorbed
        UInt fl = M >> s;
        UInt rem = M & ((ONE << s) - 1);
        UInt half = ONE << (s - 1);
        if (rem > half) { r.a0 = r.a1 = fl + 1; }
        else if (rem < half) { r.a0 = r.a1 = fl; }
        else {                                              // exact tie: round half to even
            UInt odd = fl & 1;
            r.a0 = fl + odd;        // C even: C+fl has the parity of fl
            r.a1 = fl + (odd ^ 1);  // C odd : C+fl is odd iff fl is even
        }
        return r;
    }

    // h = "f then g"
    static PW_HD Inc<T> compose(Inc<T> f, Inc<T> g) {
        Inc<T> h;
        UInt x0 = f.a0 + ((f.a0 & 1) ? g.a1 : g.a0);
        UInt x1 = f.a1 + ((f.a1 & 1) ? g.a0 : g.a1);
        h.a0 = x0 > SAT ? SAT : x0;
        h.a1 = x1 > SAT ? SAT : x1;
        return h;
    }

    static PW_HD UInt apply(UInt C, Inc<T> f) { return C + ((C & 1) ? f.a1 : f.a0); }

    // Smallest integer T_r with  C >= T_r  <=>  (double)(C * ulp) >= r   (r >= 0), clamped to TOP.
    static PW_HD UInt threshold(double r, int eb);
};

template <> PW_HD uint32_t Binade<float>::threshold(double r, int eb) {
    // ulp = 2^(eb-150); r/ulp = r * 2^(150-eb) is an exact scaling (no overflow: r < 1, eb >= 1)
    double scaled = r;
    int k = 150 - eb;               // 0 < k <= 149
    // multiply by 2^k in two exact steps (2^k itself always fits a double)
    scaled *= FloatTraits<double>::from_bits((uint64_t)(1023 + (k >> 1)) << 52);
    scaled *= FloatTraits<double>::from_bits((uint64_t)(1023 + (k - (k >> 1))) << 52);
    if (!(scaled < 16777216.0)) return TOP;
    uint32_t t = (uint32_t)scaled;  // floor
    if ((double)t < scaled) t++;
    return t;
}

template <> PW_HD uint64_t Binade<double>::threshold(double r, int eb) {
    int k = 1075 - eb;              // ulp = 2^(eb-1075)
    double scaled = r;
    // up to three exact power-of-two scalings keep every intermediate finite
    while (k > 0 && scaled < 9007199254740992.0) {
        int step = k > 512 ? 512 : k;
        scaled *= FloatTraits<double>::from_bits((uint64_t)(1023 + step) << 52);
        k -= step;
    }
    if (k > 0 || !(scaled < 9007199254740992.0)) return TOP;
    uint64_t t = (uint64_t)scaled;
    if ((double)t < scaled) t++;
    return t;
}


// ---- exact-arithmetic decision of a float32 CDF search (unit-weight rows, dyadic biases) -------------
// Setting: every element of a row weighs a whole number of units (>= 1), E(k) = exact mass of elements
// 0..k, `units` = exact total, so the exact CDF is E(k) / units; the reference adds the float32 values
// x = fl(weight / tot) one by one (np.cumsum) and returns the first k with c_k >= r (np.searchsorted).
// As long as the earlier sums are below r the float chain obeys
//     |c_j - E(j) / units| <= (sum_{i<=j} E(i) + E(j)) * 2^-24 / units
//                          <= ((j + 1) (R + wmax) - j (j + 1) / 2 + R + wmax) * 2^-24 / units =: zr / units
// with R = r * units (one relative rounding 2^-24 per addition, applied to the partial sum being rounded;
// E(i) <= E(j) - (j - i); 2^-24 relative on the three values).  With lo = ceil(R - zr), hi = ceil(R + zr):
// every j below the first k1 with E(k1) >= lo has c_j < r (induction), and E(k1) >= hi gives
// c_k1 >= r -- then k1 is the reference's answer.  `prefix` bounds j + 1 from above (length of the
// classified prefix or the row); E(k) >= k + 1 bounds it by R + 2 as well.
struct ExactThresholds {
    uint32_t lo, hi;
};
PW_HD ExactThresholds exact_thresholds_f32(double R, uint32_t prefix, uint32_t wmax_units) {
    const double wmax = (double)wmax_units + 2.0;
    const double jb = (double)prefix < R + 2.0 ? (double)prefix : R + 2.0;
    const double zr = ((jb + 6.0) * (R + wmax) - 0.5 * jb * (jb - 1.0)) * (1.0001 / 16777216.0) + 1e-6;
    const double lo = ceil(R - zr);
    ExactThresholds t;
    t.lo = lo > 0.0 ? (uint32_t)lo : 0u;
    t.hi = (uint32_t)ceil(R + zr);
    return t;
}

// the drift bound zr itself (units) for partial sums up to mass R
PW_HD double drift_bound_f32(double R, uint32_t prefix, uint32_t wmax_units) {
    const double wmax = (double)wmax_units + 2.0;
    const double jb = (double)prefix < R + 2.0 ? (double)prefix : R + 2.0;
    return ((jb + 6.0) * (R + wmax) - 0.5 * jb * (jb - 1.0)) * (1.0001 / 16777216.0) + 1e-6;
}

// float64 flavour (DenseOTF): the same bound with 2^-53; R = r * units itself is rounded (2^-52 relative).
struct ExactThresholds64 {
    uint64_t lo, hi;
};
PW_HD ExactThresholds64 exact_thresholds_f64(double R, double prefix, double wmax_units) {
    const double wmax = wmax_units + 2.0;
    const double jb = prefix < R + 2.0 ? prefix : R + 2.0;
    const double zr = ((jb + 6.0) * (R + wmax) - 0.5 * jb * (jb - 1.0)) * (1.0001 / 9007199254740992.0) +
                      R * (1.0 / 4503599627370496.0) + 1e-9;
    const double lo = ceil(R - zr);
    ExactThresholds64 t;
    t.lo = lo > 0.0 ? (uint64_t)lo : 0ull;
    t.hi = (uint64_t)ceil(R + zr);
    return t;
}

// First position k >= s of a run without common neighbours whose exact mass reaches th:
//   E(k) = base + (#"out" elements in [s, k]) << sh_out + (prev inside [s, k] ? wp : 0)
// (prev_pos == 0xffffffff: prev is not in the run).  Returns k, its mass through e_k.  Integer arithmetic.
PW_HD uint32_t solve_out_run(uint32_t s, uint32_t base, uint32_t th, uint32_t prev_pos, uint32_t sh_out, uint32_t wp,
                             uint32_t &e_k) {
    const uint32_t wo_m1 = (1u << sh_out) - 1u;
    const uint32_t need = th > base ? th - base : 0u;
    uint32_t k = need ? s + ((need + wo_m1) >> sh_out) - 1u : s;   // first k with (k - s + 1) << sh_out >= need
    e_k = base + ((k - s + 1u) << sh_out);
    if (prev_pos != 0xffffffffu && prev_pos >= s && k >= prev_pos) {   // every k < prev_pos stays below th
        const uint32_t need2 = need > wp ? need - wp : 0u;
        k = s + ((need2 + wo_m1) >> sh_out);                         // first k with ((k - s) << sh_out) + wp >= need
        if (k < prev_pos) k = prev_pos;
        e_k = base + ((k - s) << sh_out) + wp;
    }
    return k;
}

// ---- guided search over a common-neighbour list ---------------------------------------------------------------
// Both per-thread routines below look for the first list entry whose (monotone) partial mass reaches a target.  A
// plain bisection costs log2(n) DEPENDENT scattered loads -- the lane kernel's critical path -- so the search starts
// from a guess: one unaligned 16-byte load fetches entries [g - 1, g + 3) around the guessed index g, and when the
// guess is right (entry g - 1 below the target, one of the next three at or above it) the search is over after that
// single access.  Whatever the window does not settle is finished by bisection, so a wrong guess costs time, never
// correctness.  The guesses come from a per-edge HINT table (csrc/walk_lanes.hip.h: hint_build_kernel): bucket b
// of width Wd holds the first index whose prev-less mass m_i = ((P_i - i) << hs_out) + ((i + 1) << hs_in) reaches
// b * Wd, Wd = M / n + 1, M = ((d - n) << hs_out) + (n << hs_in); one 4-byte word packs hint[b] | hint[b+1] << 16.
struct ListWin {
    uint32_t v[4];
};
struct __attribute__((packed, aligned(4))) ListWinRaw {
    uint32_t v[4];
};
PW_HD ListWin load_list_window(const uint32_t *p) {
    const ListWinRaw raw = *(const ListWinRaw *)p;   // one 16-byte load, 4-byte aligned
    ListWin w;
    w.v[0] = raw.v[0]; w.v[1] = raw.v[1]; w.v[2] = raw.v[2]; w.v[3] = raw.v[3];
    return w;
}

// floor(a / b) for a, b < 2^52, b > 0, through one float64 division (a 64-bit integer division costs ~200
// instructions on the GPU; this is ~35)
PW_HD uint64_t div_floor_small(uint64_t a, uint64_t b) {
    uint64_t q = (uint64_t)((double)a / (double)b);   // correctly rounded quotient: off by at most one
    if (q * b > a) q--;
    else if ((q + 1u) * b <= a) q++;
    return q;
}
PW_HD uint32_t hint_bucket_width(uint32_t d, uint32_t n, uint32_t hs_in, uint32_t hs_out) {
    const uint64_t m = ((uint64_t)(d - n) << hs_out) + ((uint64_t)n << hs_in);
    return n ? (uint32_t)div_floor_small(m, n) + 1u : 1u;
}

struct ListHints {
    const uint32_t *h;    // this edge's hint words (nullptr: none)
    uint32_t hs_in, hs_out;
    float wd;             // bucket width in hint units
    // index the search should start from for mass tau (float arithmetic: the bucket may be off by one now and
    // then -- the search verifies its guess)
    PW_HD uint32_t guess(float tau, uint32_t n, uint32_t &reads) const {
        if (!h) return 0xffffffffu;
        const float bf = tau / wd;
        uint32_t b = bf >= (float)(n - 1u) ? n - 1u : (bf > 0.0f ? (uint32_t)bf : 0u);
        reads++;
        return h[b] & 0xffffu;
    }
};
// wd = the bucket width the table was built with (hint_bucket_width; the lane kernel keeps it in the edge record)
PW_HD ListHints make_hints(const uint32_t *h, uint32_t hs_in, uint32_t hs_out, uint32_t wd, uint32_t n) {
    ListHints lh;
    lh.h = (h && n > 0 && n <= 0xffffu && wd) ? h : nullptr;
    lh.hs_in = hs_in;
    lh.hs_out = hs_out;
    lh.wd = (float)wd;
    return lh;
}

// Hint words of one list (n entries at positions cl[], row degree d): out[b] = hint[b] | hint[b + 1] << 16 for
// b in [0, n), hint[b] = number of entries whose prev-less mass is below b * Wd (hint[n] = n).  Lists longer than
// 65535 entries get no hints (make_hints ignores the table for them); out[] is left untouched.
PW_HD void build_list_hints(const uint32_t *cl, uint32_t n, uint32_t d, uint32_t hs_in, uint32_t hs_out, uint32_t *out) {
    if (n == 0 || n > 0xffffu) return;
    const uint64_t wd = hint_bucket_width(d, n, hs_in, hs_out);
    uint32_t i = 0, prev = 0;   // prev = hint[b - 1]
    for (uint32_t b = 1; b <= n; b++) {
        uint32_t hb = n;
        if (b < n) {
            const uint64_t lim = (uint64_t)b * wd;
            while (i < n && (((uint64_t)(cl[i] - i) << hs_out) + ((uint64_t)(i + 1u) << hs_in)) < lim) i++;
            hb = i;
        }
        out[b - 1] = prev | (hb << 16);
        prev = hb;
    }
}

// First index f in [lo_min, n) with ev(f, P_f) >= target (n when none); ev monotone non-decreasing in the index.
// below: entry f - 1 and its value (has_below == false when f == lo_min); at: entry f and its value (p_at ==
// 0xffffffff when f == n).  g = guessed index (0xffffffff: none).
struct SearchResult {
    uint32_t f;
    uint32_t p_below, p_at;
    uint64_t v_below, v_at;
    bool has_below;
    ListWin win;     // the window the search loaded: entries [w0, w0 + 4), w0 == 0xffffffff: none
    uint32_t w0;
};
template <class Eval>
PW_HD SearchResult guided_search(const uint32_t *cl, uint32_t lo_min, uint32_t n, uint32_t g, const Eval &ev, uint64_t target,
                                 uint32_t &reads, const ListWin *pre = nullptr, uint32_t pre_w0 = 0xffffffffu) {
    SearchResult r;
    r.p_below = 0; r.v_below = 0; r.has_below = false; r.p_at = 0xffffffffu; r.v_at = 0;
    r.w0 = 0xffffffffu;
    r.win = ListWin{{0, 0, 0, 0}};
    uint32_t lo = lo_min, hi = n;
    if (lo < hi && (g != 0xffffffffu || pre_w0 != 0xffffffffu)) {
        uint32_t w0;
        ListWin w;
        if (pre_w0 != 0xffffffffu) {   // window fetched ahead of time (lane_chain's prefetch)
            w0 = pre_w0;
            w = *pre;
        } else {
            if (g < lo_min) g = lo_min;
            if (g > n) g = n;
            w0 = g > lo_min ? g - 1u : g;
            if (w0 >= n) w0 = n - 1u;
            w = load_list_window(cl + w0);   // may run past the list end: the lists are padded
            reads += 4;
        }
        r.win = w;
        r.w0 = w0;
        bool any_below = false, any_at = false;
#pragma unroll
        for (uint32_t t = 0; t < 4; t++) {
            const uint32_t idx = w0 + t;
            if (idx < hi && idx >= lo) {
                const uint64_t v = ev(idx, w.v[t]);
                if (v >= target) { hi = idx; r.p_at = w.v[t]; r.v_at = v; any_at = true; }
                else { lo = idx + 1u; r.p_below = w.v[t]; r.v_below = v; r.has_below = true; any_below = true; }
            }
        }
        // a near miss is finished by galloping away from the window instead of bisecting the whole list
        if (lo < hi && any_below && !any_at) {          // guess too low: probe lo + 3, lo + 11, lo + 27, ...
            uint32_t step = 4;
            while (lo < hi) {
                const uint32_t idx = hi - lo > step ? lo + step - 1u : hi - 1u;
                const uint32_t P = cl[idx];
                reads++;
                const uint64_t v = ev(idx, P);
                if (v >= target) { hi = idx; r.p_at = P; r.v_at = v; break; }
                lo = idx + 1u; r.p_below = P; r.v_below = v; r.has_below = true;
                step <<= 1;
            }
        } else if (lo < hi && any_at && !any_below) {   // guess too high: probe hi - 4, hi - 12, ...
            uint32_t step = 4;
            while (lo < hi) {
                const uint32_t idx = hi - lo > step ? hi - step : lo;
                const uint32_t P = cl[idx];
                reads++;
                const uint64_t v = ev(idx, P);
                if (v < target) { lo = idx + 1u; r.p_below = P; r.v_below = v; r.has_below = true; break; }
                hi = idx; r.p_at = P; r.v_at = v;
                step <<= 1;
            }
        }
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t P = cl[mid];
        reads++;
        const uint64_t v = ev(mid, P);
        if (v >= target) { hi = mid; r.p_at = P; r.v_at = v; }
        else { lo = mid + 1u; r.p_below = P; r.v_below = v; r.has_below = true; }
    }
    r.f = lo;
    return r;
}

// ---- the exact decision evaluated by ONE thread from the positions of the common neighbours (lane kernel) ------
// Row of d neighbours; cl[0..n_in) = ascending positions of the common neighbours of prev and cur ("in", weight
// 1), pp = position of prev (weight w_prev; 0xffffffff: prev is not a neighbour), everything else "out" (weight
// w_out); w_out, w_prev powers of two.  Returns the position np.searchsorted(np.cumsum(float32 probs), r) selects
// when the decision is certain; LANE_AMBIGUOUS when a partial sum of the exact CDF lies inside the drift bound
// (the float chain then needs the first `kmax` positions at most); LANE_REDO when the row is outside the exact
// range.  The run structure: the i-th common neighbour sits at P_i with exact mass
//   E(P_i) = ((P_i - i - [pp < P_i]) << sh_out) + ((i + 1) << sh_in) + ([pp < P_i] << sh_prev),
// monotone in i, so the first i with E(P_i) >= lo is found by a (guided) search; between P_{i-1} and P_i the row
// consists of "out" positions (and possibly prev), where the first position reaching lo is a closed form
// (solve_out_run).
constexpr uint32_t LANE_AMBIGUOUS = 0xfffffffdu;
constexpr uint32_t LANE_REDO = 0xfffffffcu;

struct LaneStep {
    float tot;        // exact row total (float32)
    uint32_t kmax;    // ambiguous steps: leading positions the float chain can need
    uint32_t probes;  // list / hint entries read
    uint32_t k1;      // ambiguous steps: every position below k1 is known to stay below r
    uint32_t f;       // ... number of common neighbours before k1
    uint32_t shifts;  // ... sh_in | sh_out << 8 | sh_prev << 16 (weights in units of the smallest)
    uint32_t p_next;  // ... position of the first common neighbour at or after k1 (0xffffffff: none)
};

struct MassEval {   // E(P_i) in units of the smallest weight
    uint32_t pp, sh_in, sh_out, sh_prev;
    PW_HD uint64_t operator()(uint32_t i, uint32_t P) const {
        const uint32_t pv = pp < P ? 1u : 0u;   // 0xffffffff compares greater than any position
        return (uint64_t)(((P - i - pv) << sh_out) + ((i + 1u) << sh_in) + (pv << sh_prev));
    }
};

PW_HD uint32_t lane_decide(uint32_t d, uint32_t n_in, uint32_t pp, double r, float w_out, float w_prev,
                           const uint32_t *cl, LaneStep &ls, const uint32_t *hint = nullptr, uint32_t hs_in = 0,
                           uint32_t hs_out = 0, uint32_t hint_wd = 0) {
    const uint32_t n_pv = pp != 0xffffffffu ? 1u : 0u;
    ls.probes = 0;
    if (n_in + n_pv > d) return LANE_REDO;
    const uint32_t n_out = d - n_in - n_pv;
    float u = 1.0f;
    if (n_out && w_out < u) u = w_out;
    if (n_pv && w_prev < u) u = w_prev;
    const double td = (double)n_in + (double)n_out * (double)w_out + (double)n_pv * (double)w_prev;
    if (!(td <= 16777216.0 * (double)u)) return LANE_REDO;   // every partial sum exact: tot = exact sum
    ls.tot = (float)td;
    const uint32_t sh_u = (FloatTraits<float>::bits(u) >> 23) & 0xffu;
    const uint32_t sh_in = (127u - sh_u) & 31u,   // classes that do not occur in the row get shift 0
                   sh_out = n_out ? (((FloatTraits<float>::bits(w_out) >> 23) & 0xffu) - sh_u) & 31u : 0u,
                   sh_prev = n_pv ? (((FloatTraits<float>::bits(w_prev) >> 23) & 0xffu) - sh_u) & 31u : 0u;
    const double units = ldexp(td, (int)(127u - sh_u));   // td / u, exact
    uint32_t sh_max = sh_in > sh_out ? sh_in : sh_out;
    if (sh_prev > sh_max) sh_max = sh_prev;
    const ExactThresholds th = exact_thresholds_f32(r * units, d, 1u << sh_max);
    const uint32_t lo_th = th.lo, hi_th = th.hi;
    const uint32_t wp = 1u << sh_prev;
    // first common neighbour whose exact mass reaches lo_th
    uint32_t s_run = 0, base = 0, p_f = 0xffffffffu, e_f = 0, f_below = 0;
    if (n_in) {
        uint32_t g = 0xffffffffu;
        if (hint && sh_in >= hs_in) {   // hint units are 2^(sh_in - hs_in) of this step's units
            const ListHints lh = make_hints(hint, hs_in, hs_out, hint_wd, n_in);
            g = lh.guess((float)(lo_th >> (sh_in - hs_in)), n_in, ls.probes);
        }
        const MassEval ev{pp, sh_in, sh_out, sh_prev};
        const SearchResult sr = guided_search(cl, 0u, n_in, g, ev, (uint64_t)lo_th, ls.probes);
        if (sr.has_below) { s_run = sr.p_below + 1u; base = (uint32_t)sr.v_below; }
        if (sr.f < n_in) { p_f = sr.p_at; e_f = (uint32_t)sr.v_at; }
        f_below = sr.f;   // entries whose mass stays below lo_th: exactly the common neighbours before k1
    }
    uint32_t e1;
    uint32_t k1 = solve_out_run(s_run, base, lo_th, (n_pv && pp >= s_run) ? pp : 0xffffffffu, sh_out, wp, e1);
    if (p_f != 0xffffffffu && k1 >= p_f) { k1 = p_f; e1 = e_f; }
    if (k1 < d && e1 >= hi_th) return k1;
    // every j < k1 has c_j < r; the chain reaches r at the latest where E >= hi_th, and E grows by >= 1 per element
    const uint64_t km = (uint64_t)k1 + (uint64_t)(hi_th > lo_th ? hi_th - lo_th : 0u) + 2ull;
    ls.kmax = km < d ? (uint32_t)km : d;
    ls.k1 = k1;
    ls.f = f_below;
    ls.shifts = sh_in | (sh_out << 8) | (sh_prev << 16);
    ls.p_next = p_f;
    return LANE_AMBIGUOUS;
}

// ---- the float32 chain itself, evaluated by ONE thread (lane kernel, ambiguous steps) ---------------------------
// c_k = c_{k-1} + x_class(k), sequential float32 additions (np.cumsum), first k with (double)c_k >= r
// (np.searchsorted, reference src/pecanpy/pecanpy.py:556-557).  Same arithmetic as the wavefront version
// (walk_sparse.hip.h: unit_chain): while the sum stays inside one binade it advances by a fixed integer number of
// ulps per class, so the first position whose sum reaches the target / the binade top is found in closed form --
// here by a bisection over the positions of the common neighbours (the only irregular class) and a division inside
// the run of "out" neighbours that holds it; the one addition per binade that crosses the top is a real float add.
// Rows: cl[0..n_in) ascending positions of the common neighbours (value x_in), pp = position of prev (x_prev,
// 0xffffffff: none), everything else x_out.  Only the first kend positions are examined.
// Returns the position, LANE_CHAIN_END when no partial sum of the first kend elements reaches r, LANE_TIE when a
// value sits exactly half way between two representable sums in some binade (parity dependent rounding: left to
// the wavefront chain, which implements it).
constexpr uint32_t LANE_CHAIN_END = 0xfffffffbu;
constexpr uint32_t LANE_TIE = 0xfffffffau;

#if !defined(__HIP_DEVICE_COMPILE__)
// host-side instrumentation of lane_chain (self test): elements added one by one after the head, binade iterations
static thread_local uint64_t g_lane_seq_elems = 0, g_lane_binades = 0;
#define PW_LANE_STAT(x) x
#else
#define PW_LANE_STAT(x)
#endif
constexpr uint32_t LANE_HEAD = 32;        // leading elements added one by one
constexpr uint32_t LANE_PF = 6;           // binades whose list window is fetched ahead of time (5 words each)
constexpr uint32_t LANE_TIE_BUDGET = 4096;  // runs walked one by one inside binades with a rounding tie

struct ChainEval {   // partial sum (in ulps of the current binade) after common neighbour i at position P
    uint64_t C, ii, io;
    uint32_t k, i0, lim;
    PW_HD uint64_t operator()(uint32_t i, uint32_t P) const {
        if (P >= lim) return ~0ull;   // outside the examined prefix
        const uint32_t cin = i - i0 + 1u;
        return C + (uint64_t)cin * ii + (uint64_t)((P - k + 1u) - cin) * io;
    }
};

PW_HD uint32_t lane_chain(uint32_t kend, uint32_t n_in, uint32_t pp, double r, float x_in, float x_out, float x_prev,
                          const uint32_t *cl, uint32_t &reads, const uint32_t *hint = nullptr, uint32_t hs_in = 0,
                          uint32_t hs_out = 0, uint32_t hint_wd = 0, uint32_t *pf = nullptr, uint32_t pf_stride = 1) {
    using B = Binade<float>;
    const ListHints lh = make_hints(hint, hs_in, hs_out, hint_wd, n_in);
    const float hint_units_per_one = ldexpf(1.0f / x_in, (int)hs_in);   // hint units of mass per unit of the sum
    // Prefetch.  The chain is a sequence of binades, each ending with a search of the list -- a chain of dependent
    // scattered loads (hint word -> window -> cursor), ~60 memory round trips per chain.  Where the sum leaves a
    // binade hardly depends on the roundings before it: the top of binade eb is the value 2^(eb - 126), so the hint
    // word and the list window every binade will need can be requested NOW, all at once (two rounds of independent
    // loads); the windows wait in `pf` (LDS on the device: slot j = words [5 j, 5 j + 5) = {w0, entries}) and the
    // searches below only touch memory again when a window does not settle them.
    const int eb_t = B::eb_of((float)r);                 // binade of the target
    const int eb_lo = eb_t - (int)LANE_PF + 1;
    if (pf && lh.h) {
        uint32_t g[LANE_PF];
#pragma unroll
        for (int j = 0; j < (int)LANE_PF; j++) {
            const int eb = eb_lo + j;
            g[j] = 0xffffffffu;
            if (eb >= 1) g[j] = lh.guess((eb == eb_t ? (float)r : ldexpf(1.0f, eb - 126)) * hint_units_per_one, n_in, reads);
        }
#pragma unroll
        for (int j = 0; j < (int)LANE_PF; j++) {
            uint32_t w0 = 0xffffffffu;
            ListWin w = ListWin{{0, 0, 0, 0}};
            if (g[j] != 0xffffffffu) {
                w0 = g[j] > 0 ? g[j] - 1u : 0u;
                if (w0 >= n_in) w0 = n_in - 1u;
                w = load_list_window(cl + w0);
                reads += 4;
            }
            uint32_t *slot = pf + (size_t)(5 * j) * pf_stride;
            slot[0] = w0;
            slot[pf_stride] = w.v[0]; slot[2 * pf_stride] = w.v[1]; slot[3 * pf_stride] = w.v[2]; slot[4 * pf_stride] = w.v[3];
        }
    }
    float c = 0.0f;
    uint32_t k = 0;    // next element to add
    uint32_t i0 = 0;   // number of common neighbours before k
    // cursor over the list: position of common neighbour i0, served from a cached 4-entry window so that walking
    // the list costs one (dependent) load per four entries instead of one per entry
    ListWin cw = {{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}};
    uint32_t cw0 = 0, next_in = 0xffffffffu;
    reads = 0;   // list entries read (statistics)
#define PW_LANE_CURSOR()                                                              \
    do {                                                                              \
        if (i0 >= n_in) next_in = 0xffffffffu;                                        \
        else {                                                                        \
            if (i0 < cw0 || i0 - cw0 >= 4u) { cw = load_list_window(cl + i0); cw0 = i0; reads += 4; } \
            const uint32_t o_ = i0 - cw0;                                             \
            next_in = o_ == 0 ? cw.v[0] : (o_ == 1 ? cw.v[1] : (o_ == 2 ? cw.v[2] : cw.v[3])); \
        }                                                                             \
    } while (0)
    if (n_in) { cw = load_list_window(cl); reads += 4; }
    PW_LANE_CURSOR();
    // PW_LANE_SEQ(n, stay): n elements one by one (real float32 additions); stops early when the sum leaves binade
    // `stay` (0: never); `hit` = the target was reached at element k.  (A macro, not a lambda: state captured by
    // reference ends up in scratch memory on the device.)
#define PW_LANE_SEQ(n_, stay_)                                                        \
    do {                                                                              \
        for (uint32_t t_ = 0; t_ < (n_) && k < kend; t_++) {                          \
            float x_ = x_out;                                                         \
            if (k == next_in) {                                                       \
                x_ = x_in;                                                            \
                i0++;                                                                 \
                PW_LANE_CURSOR();                                                     \
            } else if (k == pp) x_ = x_prev;                                          \
            c = c + x_;                                                               \
            if ((double)c >= r) { hit = true; break; }                                \
            k++;                                                                      \
            if ((stay_) && B::eb_of(c) != (stay_)) break;                             \
        }                                                                             \
    } while (0)
    bool hit = false;
    // the sum changes binade every few elements at first (and the values tie there half of the time)
    PW_LANE_SEQ(LANE_HEAD, 0);
    if (hit) return k;
    uint32_t tie_budget = LANE_TIE_BUDGET;
    while (k < kend) {
        if (k == pp) {   // prev is a single element: always a real addition (no closed form, no tie question)
            PW_LANE_SEQ(1u, 0);
            if (hit) return k;
            continue;
        }
        const uint32_t lim = (pp != 0xffffffffu && pp > k && pp < kend) ? pp : kend;   // closed form over [k, lim)
        const int eb = B::eb_of(c);
        const uint64_t C = B::sig_of(c);
        const uint64_t Tt = B::threshold(r, eb);   // <= TOP
        const Inc<float> qi = B::quantize(x_in, eb), qo = B::quantize(x_out, eb);
        if ((qi.a0 != qi.a1 && next_in < lim) || qo.a0 != qo.a1) {
            // A value sits exactly half way between two sums of this binade: its increment depends on the parity of
            // the running sum (round half to even), so the counts alone no longer determine the sum.  Walk the
            // binade RUN BY RUN instead: a run of m "out" neighbours adds (C odd ? a1 : a0) once and -- the sum being
            // even after a tying addition -- a0 for each further element (no tie: a0 == a1 throughout); a common
            // neighbour adds its own parity-selected increment.  Cost: one iteration per common neighbour inside the
            // binade instead of one per element.
            bool leave = false;   // the sum left the binade (or reached the target) at element kf
            uint32_t kf = 0;
            uint64_t Cc = C, Cprev = C;
            float xf = 0.0f;
            while (k < lim) {
                if (tie_budget == 0) return LANE_TIE;
                tie_budget--;
                const uint32_t stop = next_in < lim ? next_in : lim;   // end of the "out" run that starts at k
                const uint32_t m = stop - k;
                if (m) {
                    const uint64_t first = (Cc & 1ull) ? qo.a1 : qo.a0;
                    const uint64_t each = qo.a0;   // no tie: a0 == a1; after a tying addition the sum is even
                    // smallest t in [1, m] with Cc + first + (t - 1) * each >= Tt
                    uint64_t t = 1;
                    if (Cc + first < Tt) t = each ? 2ull + div_floor_small(Tt - (Cc + first) - 1ull, each) : 0xffffffffull;
                    if (t <= m) {
                        kf = k + (uint32_t)t - 1u;
                        const uint64_t Cf = Cc + first + (t - 1ull) * each;
                        Cprev = t == 1 ? Cc : Cf - each;
                        Cc = Cf;
                        xf = x_out;
                        leave = true;
                        break;
                    }
                    Cc += first + (uint64_t)(m - 1u) * each;
                    k = stop;
                    if (k >= lim) break;
                }
                // the common neighbour at k
                const uint64_t inc = (Cc & 1ull) ? qi.a1 : qi.a0;
                i0++;
                PW_LANE_CURSOR();
                if (Cc + inc >= Tt) {
                    kf = k;
                    Cprev = Cc;
                    Cc += inc;
                    xf = x_in;
                    leave = true;
                    break;
                }
                Cc += inc;
                k++;
            }
            PW_LANE_STAT(g_lane_seq_elems++);
            if (!leave) {   // [k_start, lim) stays inside the binade and below the target
                if (lim == kend) return LANE_CHAIN_END;
                c = B::make((uint32_t)Cc, eb);
                continue;   // k == lim == pp: prev is added next
            }
            if (Cc < (uint64_t)B::TOP) return kf;   // target reached inside the binade
            c = B::make((uint32_t)Cprev, eb) + xf;
            if ((double)c >= r) return kf;
            k = kf + 1u;   // (the list cursor already points behind kf)
            continue;
        }
        const uint64_t ii = qi.a0, io = qo.a0;
        PW_LANE_STAT(g_lane_binades++);
        // first common neighbour in [k, lim) whose partial sum reaches Tt; the run of "out" neighbours before it
        // starts at s_run with partial sum `base`.  Guess from the hint table: the exact mass at which the sum
        // equals Tt ulps (the float drift only shifts the true index by a few entries, which the window absorbs).
        uint32_t g = 0xffffffffu, pre_w0 = 0xffffffffu;
        ListWin pre = ListWin{{0, 0, 0, 0}};
        if (lh.h) {
            const int slot = eb - eb_lo;
            if (pf && slot >= 0 && slot < (int)LANE_PF) {
                const uint32_t *sp = pf + (size_t)(5 * slot) * pf_stride;
                pre_w0 = sp[0];
                pre.v[0] = sp[pf_stride]; pre.v[1] = sp[2 * pf_stride]; pre.v[2] = sp[3 * pf_stride]; pre.v[3] = sp[4 * pf_stride];
            }
            if (pre_w0 == 0xffffffffu) g = lh.guess(ldexpf((float)Tt, eb - 150) * hint_units_per_one, n_in, reads);
        }
        const ChainEval ev{C, ii, io, k, i0, lim};
        const SearchResult sr = guided_search(cl, i0, n_in, g, ev, Tt, reads, &pre, pre_w0);
        const uint32_t lo = sr.f;
        uint32_t s_run = k, p_f = 0xffffffffu;
        uint64_t base = C, g_f = 0;
        if (sr.has_below) { s_run = sr.p_below + 1u; base = sr.v_below; }
        if (sr.f < n_in && sr.v_at != ~0ull) { p_f = sr.p_at; g_f = sr.v_at; }
        const uint32_t run_end = p_f != 0xffffffffu ? p_f : lim;
        const uint64_t need = Tt > base ? Tt - base : 0ull;
        uint64_t cnt = io ? div_floor_small(need + io - 1ull, io) : 0xffffffffull;   // "out" elements needed
        if (cnt == 0) cnt = 1;
        const uint64_t j = (uint64_t)s_run + cnt - 1ull;
        uint32_t kf;
        uint64_t Cf, incf;
        float xf;
        if (j >= run_end) {
            if (p_f == 0xffffffffu) {
                // [k, lim) stays below the target and inside the binade: its exact closed-form sum
                if (lim == kend) return LANE_CHAIN_END;
                c = B::make((uint32_t)(base + (uint64_t)(lim - s_run) * io), eb);
                k = lim;
                i0 = lo;
                if (sr.w0 != 0xffffffffu) { cw = sr.win; cw0 = sr.w0; }
                PW_LANE_CURSOR();
                continue;
            }
            kf = p_f; Cf = g_f; incf = ii; xf = x_in;
        } else {
            kf = (uint32_t)j;
            Cf = base + (uint64_t)(kf - s_run + 1u) * io;
            incf = io; xf = x_out;
        }
        if (Cf < (uint64_t)B::TOP) return kf;   // the target lies inside this binade and element kf reaches it
        // element kf takes the sum over the binade top: one real float32 addition, then the next binade
        c = B::make((uint32_t)(Cf - incf), eb) + xf;
        if ((double)c >= r) return kf;
        k = kf + 1u;
        i0 = lo + (kf == p_f ? 1u : 0u);
        if (sr.w0 != 0xffffffffu) { cw = sr.win; cw0 = sr.w0; }
        PW_LANE_CURSOR();
    }
    return LANE_CHAIN_END;
#undef PW_LANE_SEQ
#undef PW_LANE_CURSOR
}

// ---- refined decision of an ambiguous step: the drift of the float32 chain, computed instead of bounded ---------
// lane_decide's bound treats every rounding as a worst case; but the chain's roundings are SYSTEMATIC.  While the
// sum stays in binade e, adding a value of class c moves it by inc_{e,c} ulps exactly (no tie), i.e. errs by the
// constant  delta_{e,c} = inc_{e,c} * ulp_e - x_c;  the only other roundings are the one addition per binade that
// crosses its top (|error| <= ulp/2 of the binade entered).  Hence with n_{e,c} = number of class-c additions inside
// binade e before position k0,
//      c_k0 = E(k0) * x_unit  +  sum_e sum_c n_{e,c} * delta_{e,c}  +  (crossing errors),
// and the n_{e,c} follow from the POSITIONS where the sum enters each binade, which lie within the a-priori drift of
// where the exact mass reaches 2^e: an E-space search per binade (mass_locate), independent of one another.  The top
// LANE_RF binades are evaluated this way, everything below is bounded (its ulps are 2^LANE_RF times smaller).  Error
// budget eps (all provable): crossings <= ulp_top; additions below the evaluated binades <= count * ulp/2;
// a boundary misplaced by m positions <= m * 0.75 ulp of its binade; float64 evaluation 2^-45.  The interval
// [c - eps, c + eps] pins the integer significand C of c_k0 to a few candidates; the step is decided when its
// lowest and highest candidate agree on the element that reaches r inside the top binade (closed form, as in
// lane_chain).  Anything else -- ties, prev next in line, a crossing before r, candidates that disagree -- returns
// LANE_AMBIGUOUS and the chain decides.  About 12 % of the RMAT-22 steps enter; ~0.x % leave undecided.
constexpr int LANE_RF = 5;

struct MassPos {
    uint32_t pos;      // first position whose exact mass reaches the target (d when none)
    uint32_t rank;     // common neighbours before pos
    bool common;       // pos is a common neighbour
};
PW_HD MassPos mass_locate(const uint32_t *cl, uint32_t n_in, uint32_t d, uint32_t pp, uint32_t sh_in, uint32_t sh_out,
                          uint32_t sh_prev, uint64_t target, uint32_t &reads) {
    MassPos m;
    uint32_t s_run = 0, base = 0, p_f = 0xffffffffu;
    m.rank = 0;
    if (n_in) {
        const MassEval ev{pp, sh_in, sh_out, sh_prev};
        const SearchResult sr = guided_search(cl, 0u, n_in, 0xffffffffu, ev, target, reads);
        if (sr.has_below) { s_run = sr.p_below + 1u; base = (uint32_t)sr.v_below; }
        if (sr.f < n_in) p_f = sr.p_at;
        m.rank = sr.f;
    }
    uint32_t e1;
    const uint32_t th = target > 0xffffffffull ? 0xffffffffu : (uint32_t)target;
    uint32_t k = solve_out_run(s_run, base, th, (pp != 0xffffffffu && pp >= s_run) ? pp : 0xffffffffu, sh_out, 1u << sh_prev, e1);
    m.common = false;
    if (p_f != 0xffffffffu && k >= p_f) { k = p_f; m.common = true; }
    m.pos = k < d ? k : d;
    return m;
}

PW_HD uint32_t lane_refine(uint32_t d, uint32_t n_in, uint32_t pp, double r, float w_out, float w_prev, const uint32_t *cl,
                           const LaneStep &ls, uint32_t &reads) {
    using B = Binade<float>;
    const uint32_t sh_in = ls.shifts & 0xffu, sh_out = (ls.shifts >> 8) & 0xffu, sh_prev = (ls.shifts >> 16) & 0xffu;
    const uint32_t k1 = ls.k1, i1 = ls.f, kend = ls.kmax;
    if (k1 == 0 || k1 >= kend) return LANE_AMBIGUOUS;
    const bool has_pv = pp != 0xffffffffu;   // (k0 = k1 - 1 is the last position known to stay below r)
    const float x_in = 1.0f / ls.tot, x_out = x_in * w_out, x_pv = x_in * w_prev;
    const double x_u = ldexp((double)x_in, -(int)sh_in);   // float value of one unit of mass (exact)
    const uint32_t pv0 = (has_pv && pp < k1) ? 1u : 0u;
    const uint64_t E0 = ((uint64_t)(k1 - i1 - pv0) << sh_out) + ((uint64_t)i1 << sh_in) + ((uint64_t)pv0 << sh_prev);
    const double v0 = (double)E0 * x_u;                     // exact: 24 x 24 bits
    int ex = 0;
    (void)frexp(v0, &ex);                                   // v0 = m * 2^ex, m in [0.5, 1)
    const int e_t = ex - 1 + 127;                           // binade (biased float32 exponent) of the exact mass value
    if (e_t < 2 || e_t > 126) return LANE_AMBIGUOUS;
    uint32_t sh_max = sh_in > sh_out ? sh_in : sh_out;
    if (sh_prev > sh_max) sh_max = sh_prev;
    // where the sum enters each of the top binades: LANE_RF independent E-space searches, run as ONE bisection loop
    // (every trip issues the probes of all searches before it waits: one memory round trip per level, not LANE_RF)
    MassPos bnd[LANE_RF];
    double eps = ldexp(1.0, e_t - 150);                     // crossing additions: sum of ulp_e / 2 over all binades
    int n_b = 0;
    uint64_t T[LANE_RF];
    uint32_t lo[LANE_RF], hi[LANE_RF], pb[LANE_RF], pa[LANE_RF];   // bisection bounds, entries below / at the target
    uint64_t vb[LANE_RF];
#pragma unroll
    for (int j = 0; j < LANE_RF; j++) {
        const int e = e_t - j;                              // bnd[j] = entry into binade e_t - j
        T[j] = 0; lo[j] = 0; hi[j] = 0; pb[j] = 0xffffffffu; pa[j] = 0xffffffffu; vb[j] = 0;
        if (e < 1) continue;
        const double tau = ldexp(1.0, e - 127) / x_u;       // mass (units) at which the exact sum reaches 2^(e - 127)
        T[j] = (uint64_t)ceil(tau);
        hi[j] = n_in;
        // the float sum enters the binade within the a-priori drift of that position
        const double zeta = 1.01 * drift_bound_f32(tau + (double)(2u << sh_max), d, 1u << sh_max);
        const double m_e = ceil(zeta) + 2.0;
        eps += m_e * 0.75 * ldexp(1.0, e - 150);
        n_b = j + 1;
    }
    {
        const MassEval ev{pp, sh_in, sh_out, sh_prev};
        bool more = true;
        while (more) {
            uint32_t P[LANE_RF];
#pragma unroll
            for (int j = 0; j < LANE_RF; j++) P[j] = lo[j] < hi[j] ? cl[(lo[j] + hi[j]) >> 1] : 0u;
            more = false;
#pragma unroll
            for (int j = 0; j < LANE_RF; j++) {
                if (lo[j] < hi[j]) {
                    const uint32_t mid = (lo[j] + hi[j]) >> 1;
                    const uint64_t v = ev(mid, P[j]);
                    reads++;
                    if (v >= T[j]) { hi[j] = mid; pa[j] = P[j]; }
                    else { lo[j] = mid + 1u; pb[j] = P[j]; vb[j] = v; }
                    more = more || lo[j] < hi[j];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LANE_RF; j++) {
        bnd[j].pos = 0; bnd[j].rank = 0; bnd[j].common = false;
        if (j >= n_b) continue;
        const uint32_t s_run = pb[j] != 0xffffffffu ? pb[j] + 1u : 0u;
        const uint32_t base = pb[j] != 0xffffffffu ? (uint32_t)vb[j] : 0u;
        uint32_t e1;
        const uint32_t th = T[j] > 0xffffffffull ? 0xffffffffu : (uint32_t)T[j];
        uint32_t k = solve_out_run(s_run, base, th, (has_pv && pp >= s_run) ? pp : 0xffffffffu, sh_out, 1u << sh_prev, e1);
        if (lo[j] < n_in && pa[j] != 0xffffffffu && k >= pa[j]) { k = pa[j]; bnd[j].common = true; }
        bnd[j].pos = k < d ? k : d;
        bnd[j].rank = lo[j];
    }
    if (n_b == 0) return LANE_AMBIGUOUS;
    // additions before the lowest evaluated binade: bounded, half an ulp of the binade below it each
    {
        const int e_lo = e_t - (n_b - 1);
        eps += ((double)bnd[n_b - 1].pos + 1.0) * 0.5 * ldexp(1.0, e_lo - 1 - 150);
    }
    // drift of the additions inside the evaluated binades
    double drift = 0.0;
#pragma unroll
    for (int j = 0; j < LANE_RF; j++) {
        if (j >= n_b) continue;
        const int e = e_t - j;
        const uint32_t a = bnd[j].pos;                                     // the crossing addition itself is in eps
        // positions a < pos < b (b = entry into the next binade), or a < pos <= k0 in the top binade
        const uint32_t b = j == 0 ? k1 : bnd[j - 1].pos;
        if (b <= a + 1u) continue;
        const uint32_t n = b - a - 1u;
        const uint32_t rank_b = j == 0 ? i1 : bnd[j - 1].rank;
        const uint32_t c_in = rank_b - bnd[j].rank - (bnd[j].common ? 1u : 0u);
        const uint32_t c_pv = (has_pv && pp > a && pp < b) ? 1u : 0u;
        if (c_in + c_pv > n) return LANE_AMBIGUOUS;                         // (cannot happen)
        const uint32_t c_out = n - c_in - c_pv;
        const Inc<float> qi = B::quantize(x_in, e), qo = B::quantize(x_out, e), qp = B::quantize(x_pv, e);
        if ((c_in && qi.a0 != qi.a1) || (c_out && qo.a0 != qo.a1) || (c_pv && qp.a0 != qp.a1)) return LANE_AMBIGUOUS;
        const double ulp = ldexp(1.0, e - 150);
        drift += (double)c_in * ((double)qi.a0 * ulp - (double)x_in) + (double)c_out * ((double)qo.a0 * ulp - (double)x_out) +
                 (double)c_pv * ((double)qp.a0 * ulp - (double)x_pv);
    }
    eps += ldexp(1.0, -45) + 1e-9 * fabs(drift);
    // integer significand of c_k0 in the top binade: candidates [C_lo, C_hi]
    const double ulp_t = ldexp(1.0, e_t - 150);
    const double c_lo = v0 + drift - eps, c_hi = v0 + drift + eps;
    if (!(c_lo >= ldexp(1.0, e_t - 127)) || !(c_hi < ldexp(1.0, e_t - 126))) return LANE_AMBIGUOUS;   // near a binade boundary
    const uint64_t C_lo = (uint64_t)ceil(c_lo / ulp_t), C_hi = (uint64_t)floor(c_hi / ulp_t);
    if (C_lo > C_hi || C_hi - C_lo > 64) return LANE_AMBIGUOUS;
    // continue inside the top binade from position k1 for both ends of the interval: same element => decided
    const uint64_t Tt = B::threshold(r, e_t);
    if (Tt >= (uint64_t)B::TOP) return LANE_AMBIGUOUS;                     // r lies beyond this binade
    if (has_pv && pp == k1) return LANE_AMBIGUOUS;                          // prev is next: a single real addition
    const uint32_t lim = (has_pv && pp > k1 && pp < kend) ? pp : kend;
    const Inc<float> qi = B::quantize(x_in, e_t), qo = B::quantize(x_out, e_t);
    if (qi.a0 != qi.a1 || qo.a0 != qo.a1) return LANE_AMBIGUOUS;
    const uint64_t ii = qi.a0, io = qo.a0;
    uint32_t answer = LANE_AMBIGUOUS;
    for (int side = 0; side < 2; side++) {
        const uint64_t C = side == 0 ? C_lo : C_hi;
        if (side == 1 && C_hi == C_lo) break;
        if (C >= Tt) return LANE_AMBIGUOUS;                                 // c_k0 >= r would contradict the a-priori bound
        const ChainEval ev{C, ii, io, k1, i1, lim};
        const SearchResult sr = guided_search(cl, i1, n_in, i1 < n_in ? i1 : 0xffffffffu, ev, Tt, reads);
        uint32_t s_run = k1, p_f = 0xffffffffu;
        uint64_t base = C;
        if (sr.has_below) { s_run = sr.p_below + 1u; base = sr.v_below; }
        if (sr.f < n_in && sr.v_at != ~0ull) p_f = sr.p_at;
        const uint32_t run_end = p_f != 0xffffffffu ? p_f : lim;
        const uint64_t need = Tt > base ? Tt - base : 0ull;
        uint64_t cnt = io ? div_floor_small(need + io - 1ull, io) : 0xffffffffull;
        if (cnt == 0) cnt = 1;
        const uint64_t jpos = (uint64_t)s_run + cnt - 1ull;
        uint32_t kf;
        if (jpos >= run_end) {
            if (p_f == 0xffffffffu) return LANE_AMBIGUOUS;                  // not reached before prev / the prefix end
            kf = p_f;
        } else kf = (uint32_t)jpos;
        if (side == 0) answer = kf;
        else if (kf != answer) return LANE_AMBIGUOUS;
    }
    return answer;
}

// ---- interval decision of an ambiguous step: the chain's drift bounded WITHOUT touching the list -------------------
// Same facts as lane_refine -- inside binade e an addition of class c errs by the constant delta_{e,c} -- but the
// per-binade class counts are not looked up: they are eliminated.  The non-crossing additions inside binade e obey
//      a_e X_in + o_e X_out + p_e X_pv = W_e        (X = x + delta: the quantised increments; W_e = span of the sum)
// so their drift is   D_e = (delta_out / X_out) W_e  +  a_e g_e  +  p_e h_e,   g_e = delta_in - delta_out X_in / X_out,
// h_e likewise for prev.  W_e is known to within two elements (a full binade spans 2^e; the top one ends at c_k0,
// known to the a-priori bound), sum a_e <= i1 = common neighbours before k1, p_e <= 1: the drift lies in
//      [ sum_e min D_e(W) + i1 min(0, g_e) + min(0, h_e) - eps,  sum_e max D_e(W) + i1 max(0, g_e) + max(0, h_e) + eps ],
// eps = crossing additions (one per binade, <= ulp_top in total) + additions below the LANE_TB evaluated binades
// (count * ulp / 2) + float64 evaluation.  On RMAT graphs the walks' ambiguous steps sit on hub rows whose common
// neighbours are a percent of the prefix (i1 ~ k1 / 100), so the interval is a small fraction of one increment and
// ~9 in 10 ambiguous steps are settled here, by arithmetic alone; the continuation from k1 only needs the next common
// neighbour's position (ls.p_next, known from lane_decide's search).  Anything doubtful returns LANE_AMBIGUOUS.
#ifndef PW_LANE_TB
#define PW_LANE_TB 6   // evaluated binades (4: +25 % float chains, 6: -4 %; each costs ~100 instructions per ambiguous step)
#endif
constexpr int LANE_TB = PW_LANE_TB;

#if !defined(__HIP_DEVICE_COMPILE__)
static thread_local uint64_t g_tight_reason[24] = {0};
#define TIGHT_BAIL(i) do { g_tight_reason[i]++; return LANE_AMBIGUOUS; } while (0)
#else
#define TIGHT_BAIL(i) return LANE_AMBIGUOUS
#endif
// increments of adding x inside binade eb (sum even / odd, as Binade::quantize) and the error of a0 in ulps of that
// binade, d0 = a0 - x / ulp (exact: the discarded bits of the significand; a1's error is d0 + (a1 - a0))
struct QuantErr {
    uint32_t a0, a1;
    float d0;
};
PW_HD QuantErr quant_err(float x, int eb) {
    using B = Binade<float>;
    const uint32_t M = B::sig_of(x);
    const int s = eb - B::eb_of(x);
    QuantErr q;
    q.a0 = q.a1 = 0;
    q.d0 = 0.0f;
    if (M == 0) return q;
    if (s <= 0) { q.a0 = q.a1 = B::SAT; return q; }
    if (s > B::MANT + 2) { q.d0 = -ldexpf((float)M, -s); return q; }
    const uint32_t fl = M >> s, rem = M & ((1u << s) - 1u), half = 1u << (s - 1);
    if (rem > half) q.a0 = q.a1 = fl + 1u;
    else if (rem < half) q.a0 = q.a1 = fl;
    else { const uint32_t odd = fl & 1u; q.a0 = fl + odd; q.a1 = fl + (odd ^ 1u); }
    q.d0 = (float)(int)(q.a0 - fl) - ldexpf((float)rem, -s);
    return q;
}

PW_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);   // 1 ulp; every use below carries a 2^-20 relative allowance
#else
    return 1.0f / x;
#endif
}

// Arithmetic: everything is kept in ULPS OF THE TOP BINADE as float32 (errors of binade e_t - j scale by 2^-j), the
// exact mass value v0 = E0 * x_unit as a 48-bit integer product split into its integer significand V and a fraction;
// float32 roundings of the interval's terms are covered by a 2^-20 relative allowance on their magnitudes.
PW_HD uint32_t lane_tight(uint32_t d, uint32_t pp, double r, float w_out, float w_prev, const LaneStep &ls) {
    using B = Binade<float>;
    const uint32_t sh_in = ls.shifts & 0xffu, sh_out = (ls.shifts >> 8) & 0xffu, sh_prev = (ls.shifts >> 16) & 0xffu;
    const uint32_t k1 = ls.k1, i1 = ls.f, kend = ls.kmax, p_f = ls.p_next;
    if (k1 == 0 || k1 >= kend) TIGHT_BAIL(1);
    const bool has_pv = pp != 0xffffffffu;
    const float x_in = 1.0f / ls.tot, x_out = x_in * w_out, x_pv = x_in * w_prev;
    const float x_u = ldexpf(x_in, -(int)sh_in);            // float value of one unit of mass (exact)
    const uint32_t pv0 = (has_pv && pp < k1) ? 1u : 0u;
    const uint32_t o0 = k1 - i1 - pv0;
    if (o0 == 0) TIGHT_BAIL(2);                             // (the elimination runs over the "out" class)
    const uint64_t E0 = ((uint64_t)o0 << sh_out) + ((uint64_t)i1 << sh_in) + ((uint64_t)pv0 << sh_prev);
    if (E0 >> 25) TIGHT_BAIL(3);                            // (lane_decide's range: at most 2^24 units)
    // v0 = E0 * x_u = P * 2^(e_u - 150), P = E0 * M_u < 2^49: binade e_t = e_u + sh, integer significand V = P >> sh
    const uint64_t P = E0 * (uint64_t)B::sig_of(x_u);
    if (P < (1ull << 24)) TIGHT_BAIL(3);
#if defined(__HIP_DEVICE_COMPILE__)
    const int top = 63 - __clzll((long long)P);
#else
    const int top = 63 - __builtin_clzll(P);
#endif
    const int sh = top - 23;                                // >= 1
    const int e_t = B::eb_of(x_u) + sh;                     // binade (biased float32 exponent) of the exact mass value
    if (e_t < 2 || e_t > 126) TIGHT_BAIL(3);
    const uint32_t V = (uint32_t)(P >> sh);                 // in [2^23, 2^24)
    const float fr = ldexpf((float)(uint32_t)(P & ((1ull << sh) - 1ull)), -sh);   // v0 / ulp_t = V + fr, fr in [0, 1)
    const float voff = (float)(V - (1u << 23));             // offset of v0 inside its binade, whole ulps (exact)
    const float upu = ldexpf((float)B::sig_of(x_u), -sh);   // ulps (top binade) per unit of mass
    uint32_t sh_max = sh_in > sh_out ? sh_in : sh_out;
    if (sh_prev > sh_max) sh_max = sh_prev;
    // a-priori: |c_k0 - v0| <= Z (drift_bound_f32, evaluated in float32 and inflated), hence the binade of c_k0 itself
    float Z;
    {
        const float units = ldexpf(ls.tot, (int)sh_in);
        const float R = (float)r * units, wmax = (float)(1u << sh_max) + 2.0f;
        const float jb = (float)d < R + 2.0f ? (float)d : R + 2.0f;
        const float zr = ((jb + 6.0f) * (R + wmax) - 0.5f * jb * (jb - 1.0f)) * (1.001f / 16777216.0f) + 1e-6f;
        Z = (1.01f * zr + 1.01f * (float)E0 * (1.0f / 16777216.0f) + 0.01f) * upu * 1.002f + 1.0f;
    }
    if (!(voff - Z >= 0.0f) || !(voff + Z + 2.0f < 8388608.0f)) TIGHT_BAIL(4);
    float xmax = x_in > x_out ? x_in : x_out;
    if (has_pv && x_pv > xmax) xmax = x_pv;
    const float xmax_u = ldexpf((float)B::sig_of(xmax), B::eb_of(xmax) - e_t) * 1.0001f;   // largest value, ulps of the top binade
    float d_lo = 0.0f, d_hi = 0.0f, g_lo = 0.0f, g_hi = 0.0f, h_lo = 0.0f, h_hi = 0.0f, mag = 0.0f;
    float ro_t = 0.0f, scale = 1.0f;
    int jl = 0;                                             // lowest evaluated binade: e_t - jl
#pragma unroll
    for (int j = 0; j < LANE_TB; j++, scale *= 0.5f) {
        const int e = e_t - j;
        if (e < 2) break;
        const QuantErr qo = quant_err(x_out, e);
        if (qo.a0 != qo.a1 || qo.a0 == 0 || qo.a0 >= B::SAT) {   // no constant "out" increment here: this binade and
            if (j == 0) TIGHT_BAIL(5);                            // everything below it are bounded, not evaluated
            break;
        }
        jl = j;
        const float ro = qo.d0 * fast_rcp((float)qo.a0);   // drift per ulp of span, "out" additions only (dimensionless)
        if (j == 0) ro_t = ro;                              // (the top binade's span: below)
        else {                                              // a full binade spans 2^23 of its ulps, less two elements
            float w_lo = 8388608.0f - 2.0f * (xmax_u / scale) - 2.0f;
            if (w_lo < 0.0f) w_lo = 0.0f;
            const float b0 = ro * w_lo * scale, b1 = ro * 8388608.0f * scale;
            d_lo += b0 < b1 ? b0 : b1;
            d_hi += b0 < b1 ? b1 : b0;
            mag += fabsf(b0) + fabsf(b1);
        }
        // (a value >= the binade's bottom always leaves it: a crossing addition, none inside; a tie rounds either way)
        if (i1) {
            const QuantErr qi = quant_err(x_in, e);
            if (qi.a0 < B::SAT) {
                const float g = (qi.d0 - ro * (float)qi.a0) * scale;
                if (g < g_lo) g_lo = g;
                if (g > g_hi) g_hi = g;
                if (qi.a1 != qi.a0) {
                    const float g1 = (qi.d0 + (float)(int)(qi.a1 - qi.a0) - ro * (float)qi.a1) * scale;
                    if (g1 < g_lo) g_lo = g1;
                    if (g1 > g_hi) g_hi = g1;
                }
            }
        }
        if (pv0) {
            const QuantErr qp = quant_err(x_pv, e);
            if (qp.a0 < B::SAT) {
                const float h = (qp.d0 - ro * (float)qp.a0) * scale;
                if (h < h_lo) h_lo = h;
                if (h > h_hi) h_hi = h;
                if (qp.a1 != qp.a0) {
                    const float h1 = (qp.d0 + (float)(int)(qp.a1 - qp.a0) - ro * (float)qp.a1) * scale;
                    if (h1 < h_lo) h_lo = h1;
                    if (h1 > h_hi) h_hi = h1;
                }
            }
        }
    }
    if (e_t - jl <= 2) TIGHT_BAIL(8);
    // crossing additions: one per binade, <= half an ulp of the binade entered (1 ulp of the top one in total);
    // additions before the evaluated binades: count * half an ulp of the binade below the lowest evaluated one;
    // float32 evaluation of the terms: 2^-20 of their magnitudes
    const float low_scale = ldexpf(1.0f, -jl);
    float n_low = 1.02f * 8388608.0f * low_scale / upu + 3.0f;
    if (n_low > (float)k1) n_low = (float)k1;
    const float gi = (float)i1;
    const float eps = 1.0f + n_low * 0.25f * low_scale +
                      (1.0f / 1048576.0f) * (mag + fabsf(ro_t) * 8388608.0f + gi * (g_hi - g_lo) + (h_hi - h_lo)) + 0.02f;
    // the top binade spans [entry, c_k0]: c_k0 from the a-priori bound first, then from the interval that gives
    float lo_off = -Z, hi_off = Z;                          // c_k0 / ulp_t - (V + fr)
#pragma unroll
    for (int it = 0; it < 2; it++) {
        float w_lo = voff + lo_off - xmax_u - 2.0f;
        if (w_lo < 0.0f) w_lo = 0.0f;
        const float b0 = ro_t * w_lo, b1 = ro_t * (voff + hi_off + 2.0f);
        const float n_lo = d_lo + (b0 < b1 ? b0 : b1) + gi * g_lo + h_lo - eps;
        const float n_hi = d_hi + (b0 < b1 ? b1 : b0) + gi * g_hi + h_hi + eps;
        if (n_lo > lo_off) lo_off = n_lo;
        if (n_hi < hi_off) hi_off = n_hi;
    }
    lo_off -= 4e-6f * (fabsf(lo_off) + 1.0f);
    hi_off += 4e-6f * (fabsf(hi_off) + 1.0f);
    if (!(voff + lo_off >= 0.0f) || !(voff + hi_off + 2.0f < 8388608.0f) || !(lo_off <= hi_off)) TIGHT_BAIL(9);
#if !defined(__HIP_DEVICE_COMPILE__)
    if (getenv("PW_TIGHT_DEBUG"))
        fprintf(stderr, "tight k1=%u i1=%u pv0=%u width=%.1f ulps: base=%.1f g=%.1f h=%.1f eps=%.1f io=%u Z=%.1f\n", k1, i1, pv0, hi_off - lo_off,
                d_hi - d_lo, gi * (g_hi - g_lo), h_hi - h_lo, 2 * eps, B::quantize(x_out, e_t).a0, Z);
#endif
    uint64_t C_lo = (uint64_t)((int64_t)V + (int64_t)ceilf(fr + lo_off)), C_hi = (uint64_t)((int64_t)V + (int64_t)floorf(fr + hi_off));
    const uint64_t Tt = B::threshold(r, e_t);
    if (Tt >= (uint64_t)B::TOP) TIGHT_BAIL(10);                     // r lies beyond this binade
    if (C_hi >= Tt) C_hi = Tt - 1ull;                                      // c_k0 < r is known (lane_decide)
    if (C_lo > C_hi) TIGHT_BAIL(11);
    // the chain inside the top binade from position k1, for both ends of the interval: same element => decided
    const Inc<float> qo = B::quantize(x_out, e_t), qi = B::quantize(x_in, e_t), qp = B::quantize(x_pv, e_t);
    const uint32_t lim = (p_f != 0xffffffffu && p_f < kend) ? p_f + 1u : kend;   // classes are known up to p_next
    if (p_f < lim && qi.a0 != qi.a1) TIGHT_BAIL(13);
    const uint32_t sp = (has_pv && pp >= k1 && pp < lim) ? pp : 0xffffffffu;   // prev ahead: the other special position
    if (sp != 0xffffffffu && qp.a0 != qp.a1) TIGHT_BAIL(14);
    const uint64_t io = qo.a0;                                             // (no tie, not 0: checked for the top binade)
    uint32_t pos = k1;
    for (int seg = 0; seg < 3; seg++) {
        // run of "out" positions [pos, nxt), then the special position nxt (prev, or the common neighbour p_next)
        const uint32_t nxt = (sp != 0xffffffffu && sp >= pos) ? sp : ((p_f < lim && p_f >= pos) ? p_f : lim);
        const uint64_t n_hi = div_floor_small(Tt - C_hi + io - 1ull, io);   // additions the upper candidate needs (>= 1)
        const uint64_t run = (uint64_t)(nxt - pos);
        if (n_hi <= run) {                                                 // it gets there inside the run:
            if (C_lo + n_hi * io >= Tt) return pos + (uint32_t)n_hi - 1u;  // ... and so does the lower one, no earlier
            TIGHT_BAIL(16);
        }
        if (nxt >= lim) TIGHT_BAIL(15);                                    // classes beyond are not known here
        C_lo += run * io;
        C_hi += run * io;
        const uint64_t inc = nxt == sp ? qp.a0 : qi.a0;
        C_lo += inc;
        C_hi += inc;
        if (C_hi >= Tt) { if (C_lo >= Tt) return nxt; TIGHT_BAIL(16); }
        if (nxt == p_f) TIGHT_BAIL(15);                                    // the next common neighbour is not known
        pos = nxt + 1u;
    }
    TIGHT_BAIL(15);
}

#undef TIGHT_BAIL
}  // namespace pw
