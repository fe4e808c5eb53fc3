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
#else
#define PW_HD inline
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

// ---- the exact decision evaluated by ONE thread from the positions of the common neighbours (lane kernel) ------
// Row of d neighbours; cl[0..n_in) = ascending positions of the common neighbours of prev and cur ("in", weight
// 1), pp = position of prev (weight w_prev; 0xffffffff: prev is not a neighbour), everything else "out" (weight
// w_out); w_out, w_prev powers of two.  Returns the position np.searchsorted(np.cumsum(float32 probs), r) selects
// when the decision is certain; LANE_AMBIGUOUS when a partial sum of the exact CDF lies inside the drift bound
// (the float chain then needs the first `kmax` positions at most); LANE_REDO when the row is outside the exact
// range.  The run structure: the i-th common neighbour sits at P_i with exact mass
//   E(P_i) = ((P_i - i - [pp < P_i]) << sh_out) + ((i + 1) << sh_in) + ([pp < P_i] << sh_prev),
// monotone in i, so the first i with E(P_i) >= lo is found by bisection; between P_{i-1} and P_i the row consists
// of "out" positions (and possibly prev), where the first position reaching lo is a closed form (solve_out_run).
constexpr uint32_t LANE_AMBIGUOUS = 0xfffffffdu;
constexpr uint32_t LANE_REDO = 0xfffffffcu;

struct LaneStep {
    float tot;        // exact row total (float32)
    uint32_t kmax;    // ambiguous steps: leading positions the float chain can need
    uint32_t probes;  // list entries read by the bisection
};

PW_HD uint32_t lane_decide(uint32_t d, uint32_t n_in, uint32_t pp, double r, float w_out, float w_prev,
                           const uint32_t *cl, LaneStep &ls) {
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
    uint32_t lo = 0, hi = n_in, s_run = 0, base = 0, p_f = 0xffffffffu, e_f = 0;
    while (lo < hi) {
        ls.probes++;
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t P = cl[mid];
        const uint32_t pv = pp < P ? 1u : 0u;   // 0xffffffff compares greater than any position
        const uint32_t ea = ((P - mid - pv) << sh_out) + ((mid + 1u) << sh_in) + (pv << sh_prev);   // E(P)
        if (ea >= lo_th) { hi = mid; p_f = P; e_f = ea; }
        else { lo = mid + 1u; s_run = P + 1u; base = ea; }
    }
    uint32_t e1;
    uint32_t k1 = solve_out_run(s_run, base, lo_th, (n_pv && pp >= s_run) ? pp : 0xffffffffu, sh_out, wp, e1);
    if (p_f != 0xffffffffu && k1 >= p_f) { k1 = p_f; e1 = e_f; }
    if (k1 < d && e1 >= hi_th) return k1;
    // every j < k1 has c_j < r; the chain reaches r at the latest where E >= hi_th, and E grows by >= 1 per element
    const uint64_t km = (uint64_t)k1 + (uint64_t)(hi_th > lo_th ? hi_th - lo_th : 0u) + 2ull;
    ls.kmax = km < d ? (uint32_t)km : d;
    return LANE_AMBIGUOUS;
}

}  // namespace pw
