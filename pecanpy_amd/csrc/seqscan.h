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

// ---- per-edge lists of common-neighbour positions -----------------------------------------------------------------
// The lane kernel's index (walk_lanes.hip.h) stores, for every CSR entry e = (u -> v), the ascending positions in
// row v of the common neighbours of u and v.  Positions of rows of at most 65536 entries are uint16, of longer rows
// uint32; a list starts at a 16-byte boundary (or inside the entry's 64-byte edge line: 8-byte boundary).
struct ListWin {
    uint32_t v[4];
};
// PIVOTS (round 4).  A list that does not fit its edge line leaves the line's 40-byte inline area unused; the index build
// stores there every step-th entry of the list -- 20 uint16 (10 uint32) pivots, entry (k + 1) * step for k = 0.., step =
// n / 21 (n / 11) -- and the searches below bisect the pivots first: the upper ~4.4 (3.5) levels of a long list's
// bisection are then probes of the line the step has fetched anyway (cache hits) instead of dependent trips to the
// overflow array.  One lane's search depth sets the pace of its whole wavefront (the deepest search of an iteration
// averages 9 probes at RMAT-22 against 2.4 for the average lane), and each level is a memory round trip.
constexpr uint32_t LIST_PIVOTS_NARROW = 20, LIST_PIVOTS_WIDE = 10;
PW_HD uint32_t list_pivot_count(uint32_t wide) { return wide ? LIST_PIVOTS_WIDE : LIST_PIVOTS_NARROW; }
// a list of n entries has pivots iff n exceeds their number (then step >= 1; such a list is never inline)
PW_HD bool list_has_pivots(uint32_t wide, uint32_t n) { return n > list_pivot_count(wide); }
PW_HD uint32_t list_pivot_step(uint32_t wide, uint32_t n) { return wide ? n / (LIST_PIVOTS_WIDE + 1u) : n / (LIST_PIVOTS_NARROW + 1u); }

struct ListView {
    const void *p;      // first entry
    uint32_t wide;      // 1: uint32 entries, 0: uint16 entries
    const void *piv = nullptr;   // pivots (same entry width), or nullptr
    uint32_t npiv = 0;           // their number (0: none)
    uint32_t step = 0;           // pivot k = entry (k + 1) * step
    // Lane kernel (PW_LANES_LINE_LDS): bytes 16..63 of the edge line -- row start, list offset and the inline area: the list
    // itself or its pivots -- were copied to LDS when the step that entered the edge was applied (three 16-byte pieces,
    // 1024 bytes apart: global_load_lds writes piece c of lane t at base + 1024 c + 16 t).  `tail` = LDS address of this
    // lane's piece 0; what lives in the line is then read from there instead of from global memory.
    uint32_t tail = 0xffffffffu; // 0xffffffff: not staged
    uint32_t inl = 0;            // the list itself lives in the line (p points into it)
    uint32_t tshift = 10;        // log2 of the distance of the 16-byte pieces in LDS: 10 (LDS-DMA by the lane itself: piece c of lane t
                                 // at base + 1024 c + 16 t) or 4 (QUAD fetch, round 5: the 64 bytes of a line are contiguous)
#if defined(__HIP_DEVICE_COMPILE__)
    __device__ __forceinline__ uint32_t tail_entry(uint32_t i) const {
        const uint32_t o = 8u + (wide ? i << 2 : i << 1);             // (line offset 24 = tail offset 8)
        const uint32_t addr = tail + ((o >> 4) << tshift) + (o & 15u);
        return wide ? *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)addr
                    : (uint32_t) * (const __attribute__((address_space(3))) uint16_t *)(uintptr_t)addr;
    }
#endif
    PW_HD uint32_t at(uint32_t i) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if (tail != 0xffffffffu && inl) return tail_entry(i);
#endif
        return wide ? ((const uint32_t *)p)[i] : (uint32_t)((const uint16_t *)p)[i];
    }
    PW_HD uint32_t pivot(uint32_t k) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if (tail != 0xffffffffu) return tail_entry(k);
#endif
        return wide ? ((const uint32_t *)piv)[k] : (uint32_t)((const uint16_t *)piv)[k];
    }
    // entries [i & ~3, (i & ~3) + 4) in one access (16 / 8 bytes, aligned to 4 entries); entries past the end of
    // the list are garbage the callers never use (the allocations are padded)
    PW_HD ListWin window(uint32_t i) const {
        const uint32_t w0 = i & ~3u;
        ListWin w;
#if defined(__HIP_DEVICE_COMPILE__)
        if (tail != 0xffffffffu && inl) {   // (inline lists are uint16: four entries = one aligned 8-byte LDS read inside a piece)
            const uint32_t o = 8u + (w0 << 1);
            const uint32_t addr = tail + ((o >> 4) << tshift) + (o & 15u);
            struct __attribute__((aligned(8))) Raw2 { uint32_t x, y; };
            const Raw2 raw = *(const __attribute__((address_space(3))) Raw2 *)(uintptr_t)addr;
            w.v[0] = raw.x & 0xffffu; w.v[1] = raw.x >> 16; w.v[2] = raw.y & 0xffffu; w.v[3] = raw.y >> 16;
            return w;
        }
#endif
        if (wide) {
            struct __attribute__((packed, aligned(4))) Raw { uint32_t v[4]; };
            const Raw raw = *(const Raw *)((const uint32_t *)p + w0);
            w.v[0] = raw.v[0]; w.v[1] = raw.v[1]; w.v[2] = raw.v[2]; w.v[3] = raw.v[3];
        } else {
            struct __attribute__((aligned(8))) Raw2 { uint32_t x, y; };
            const Raw2 raw = *(const Raw2 *)((const uint16_t *)p + w0);
            w.v[0] = raw.x & 0xffffu; w.v[1] = raw.x >> 16; w.v[2] = raw.y & 0xffffu; w.v[3] = raw.y >> 16;
        }
        return w;
    }
};

// the view of a list held in one buffer (self tests): entries at cl, pivots (when the list has them) piv_off entries further on
PW_HD ListView list_view_of(const void *cl, uint32_t wide, uint32_t n_cl, uint32_t piv_off) {
    ListView v{cl, wide};
    if (piv_off && list_has_pivots(wide, n_cl)) {
        v.piv = wide ? (const void *)((const uint32_t *)cl + piv_off) : (const void *)((const uint16_t *)cl + piv_off);
        v.npiv = list_pivot_count(wide);
        v.step = list_pivot_step(wide, n_cl);
    }
    return v;
}

// first index i in [0, n) with cl.at(i) >= x (n: none) -- the pivots first (entries (k + 1) * step: probes of the edge line),
// then a bisection of what is left
PW_HD uint32_t list_lower_bound_pos(const ListView &cl, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n;
    if (cl.npiv) {
        uint32_t klo = 0, khi = cl.npiv;
        while (klo < khi) {
            const uint32_t km = (klo + khi) >> 1;
            if (cl.pivot(km) >= x) { khi = km; hi = (km + 1u) * cl.step; }
            else { klo = km + 1u; lo = (km + 1u) * cl.step + 1u; }
        }
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cl.at(mid) < x) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

// floor(a / b) for a, b < 2^52, b > 0, through one float64 division (a 64-bit integer division costs ~200
// instructions on the GPU; this is ~35)
PW_HD uint64_t div_floor_small(uint64_t a, uint64_t b) {
    uint64_t q = (uint64_t)((double)a / (double)b);   // correctly rounded quotient: off by at most one
    if (q * b > a) q--;
    else if ((q + 1u) * b <= a) q++;
    return q;
}

// First index f in [lo_min, n) with ev(f, P_f) >= target (n when none); ev monotone non-decreasing in the index.
// below: entry f - 1 and its value (has_below == false when f == lo_min); at: entry f and its value (p_at ==
// 0xffffffff when f == n).  Plain bisection: ~log2(n) dependent probes.  (Hint tables, galloping from a guessed
// window and sampled skip arrays were measured neutral at RMAT-22 and removed -- DESIGN.md section 9b.)
struct SearchResult {
    uint32_t f;
    uint32_t p_below, p_at;
    uint64_t v_below, v_at;
    bool has_below;
};
// The first levels of either search, over the list's pivots: ordinary bisection steps whose probe index is restricted
// to the pivot entries (k + 1) * step inside [lo, hi) -- the invariants of the search (below = entry lo - 1, at = entry
// hi) hold after every step, so the plain bisection continues from whatever range is left.
template <class Eval>
PW_HD void list_search_pivots(const ListView &cl, const Eval &ev, uint64_t target, uint32_t &lo, uint32_t &hi, SearchResult &r) {
    uint32_t klo = 0, khi = cl.npiv;
    while (klo < khi) {
        const uint32_t km = (klo + khi) >> 1;
        const uint32_t ik = (km + 1u) * cl.step;     // (< hi: hi is n or a pivot further right)
        if (ik < lo) { klo = km + 1u; continue; }    // left of the range asked for: nothing to learn
        const uint32_t P = cl.pivot(km);
        const uint64_t v = ev(ik, P);
        if (v >= target) { khi = km; hi = ik; r.p_at = P; r.v_at = v; }
        else { klo = km + 1u; lo = ik + 1u; r.p_below = P; r.v_below = v; r.has_below = true; }
    }
}

template <class Eval>
PW_HD SearchResult list_search(const ListView &cl, uint32_t lo_min, uint32_t n, const Eval &ev, uint64_t target, uint32_t &reads) {
    SearchResult r;
    r.p_below = 0; r.v_below = 0; r.has_below = false; r.p_at = 0xffffffffu; r.v_at = 0;
    uint32_t lo = lo_min, hi = n;
    if (cl.npiv) list_search_pivots(cl, ev, target, lo, hi, r);
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t P = cl.at(mid);
        reads++;
        const uint64_t v = ev(mid, P);
        if (v >= target) { hi = mid; r.p_at = P; r.v_at = v; }
        else { lo = mid + 1u; r.p_below = P; r.v_below = v; r.has_below = true; }
    }
    r.f = lo;
    return r;
}

// The same search with THREE probes per level (the range is cut into quarters): half the DEPENDENT memory round trips
// of the bisection for 1.5x its probes.  For the float chains of the FLOATS lane kernel (lane_chain<true>): a chain is
// ~17 binades x one search each, every probe a scattered load the next one waits for.  Same result as list_search,
// field by field.
template <class Eval>
PW_HD SearchResult list_search_wide(const ListView &cl, uint32_t lo_min, uint32_t n, const Eval &ev, uint64_t target, uint32_t &reads) {
    SearchResult r;
    r.p_below = 0; r.v_below = 0; r.has_below = false; r.p_at = 0xffffffffu; r.v_at = 0;
    uint32_t lo = lo_min, hi = n;   // invariants: below = entry lo - 1 (when has_below), at = entry hi (when hi < n)
    // (the pivots are NOT used here: their bisection is one dependent probe per level, this search resolves two levels per
    //  round trip -- with pivots the FLOATS form at RMAT-22 went from 963 to 1053 ms per pass)
    while (hi - lo >= 3u && hi > lo) {
        const uint32_t m2 = lo + ((hi - lo) >> 1);
        const uint32_t m1 = lo + ((m2 - lo) >> 1);
        const uint32_t m3 = m2 + 1u + ((hi - m2 - 1u) >> 1);
        const uint32_t P1 = cl.at(m1), P2 = cl.at(m2), P3 = cl.at(m3);   // (independent loads: in flight together)
        reads += 3;
        const uint64_t v1 = ev(m1, P1), v2 = ev(m2, P2), v3 = ev(m3, P3);
        if (v1 >= target) { hi = m1; r.p_at = P1; r.v_at = v1; }
        else if (v2 >= target) { lo = m1 + 1u; r.p_below = P1; r.v_below = v1; r.has_below = true; hi = m2; r.p_at = P2; r.v_at = v2; }
        else if (v3 >= target) { lo = m2 + 1u; r.p_below = P2; r.v_below = v2; r.has_below = true; hi = m3; r.p_at = P3; r.v_at = v3; }
        else { lo = m3 + 1u; r.p_below = P3; r.v_below = v3; r.has_below = true; }
    }
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t P = cl.at(mid);
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
// monotone in i, so the first i with E(P_i) >= lo is found by a bisection; between P_{i-1} and P_i the row
// consists of "out" positions (and possibly prev), where the first position reaching lo is a closed form
// (solve_out_run).
constexpr uint32_t LANE_AMBIGUOUS = 0xfffffffdu;
constexpr uint32_t LANE_REDO = 0xfffffffcu;

struct LaneStep {
    float tot;        // exact row total (float32)
    uint32_t kmax;    // ambiguous steps: leading positions the float chain can need
    uint32_t probes;  // list entries read
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

// exact row total of a unit-weight row: n_in common neighbours (1), prev (w_prev) when pp names it, the rest w_out.
// ONE definition: lane_decide takes the step's total from here, and so does the lane kernel's deferred interval
// decision, which recomputes it instead of carrying it (n_in + [pp present] <= d: lane_decide checked)
PW_HD double lane_row_total(uint32_t d, uint32_t n_in, uint32_t pp, float w_out, float w_prev) {
    const uint32_t n_pv = pp != 0xffffffffu ? 1u : 0u;
    const uint32_t n_out = d - n_in - n_pv;
    return (double)n_in + (double)n_out * (double)w_out + (double)n_pv * (double)w_prev;
}

// lane_decide in three parts -- thresholds / the search over the list / the verdict -- so that the lane kernel can run the
// search its own way (round 5: whole 64-byte sectors of the list fetched by quads of lanes, walk_lanes.hip.h); the search's
// result (first entry whose mass reaches lo_th, its neighbours' positions and masses) does not depend on the probe order.
struct DecideCtx {
    uint32_t lo_th, hi_th;
    uint32_t sh_in, sh_out, sh_prev;
    uint32_t n_pv;
};
// returns LANE_REDO (row outside the exact range) or 0: go on with the search (n_in > 0) and lane_decide_end
PW_HD uint32_t lane_decide_begin(uint32_t d, uint32_t n_in, uint32_t pp, double r, float w_out, float w_prev, LaneStep &ls, DecideCtx &c) {
    const uint32_t n_pv = pp != 0xffffffffu ? 1u : 0u;
    ls.probes = 0;
    if (n_in + n_pv > d) return LANE_REDO;
    const uint32_t n_out = d - n_in - n_pv;
    float u = 1.0f;
    if (n_out && w_out < u) u = w_out;
    if (n_pv && w_prev < u) u = w_prev;
    const double td = lane_row_total(d, n_in, pp, w_out, w_prev);
    if (!(td <= 16777216.0 * (double)u)) return LANE_REDO;   // every partial sum exact: tot = exact sum
    ls.tot = (float)td;
    const uint32_t sh_u = (FloatTraits<float>::bits(u) >> 23) & 0xffu;
    c.sh_in = (127u - sh_u) & 31u;   // classes that do not occur in the row get shift 0
    c.sh_out = n_out ? (((FloatTraits<float>::bits(w_out) >> 23) & 0xffu) - sh_u) & 31u : 0u;
    c.sh_prev = n_pv ? (((FloatTraits<float>::bits(w_prev) >> 23) & 0xffu) - sh_u) & 31u : 0u;
    c.n_pv = n_pv;
    const double units = ldexp(td, (int)(127u - sh_u));   // td / u, exact
    uint32_t sh_max = c.sh_in > c.sh_out ? c.sh_in : c.sh_out;
    if (c.sh_prev > sh_max) sh_max = c.sh_prev;
    const ExactThresholds th = exact_thresholds_f32(r * units, d, 1u << sh_max);
    c.lo_th = th.lo; c.hi_th = th.hi;
    return 0u;
}
// sr: the search for the first common neighbour whose exact mass reaches c.lo_th (nullptr when n_in == 0)
PW_HD uint32_t lane_decide_end(uint32_t d, uint32_t n_in, uint32_t pp, const DecideCtx &c, const SearchResult *sr, LaneStep &ls) {
    const uint32_t lo_th = c.lo_th, hi_th = c.hi_th;
    const uint32_t wp = 1u << c.sh_prev;
    uint32_t s_run = 0, base = 0, p_f = 0xffffffffu, e_f = 0, f_below = 0;
    if (n_in) {
        if (sr->has_below) { s_run = sr->p_below + 1u; base = (uint32_t)sr->v_below; }
        if (sr->f < n_in) { p_f = sr->p_at; e_f = (uint32_t)sr->v_at; }
        f_below = sr->f;   // entries whose mass stays below lo_th: exactly the common neighbours before k1
    }
    uint32_t e1;
    uint32_t k1 = solve_out_run(s_run, base, lo_th, (c.n_pv && pp >= s_run) ? pp : 0xffffffffu, c.sh_out, wp, e1);
    if (p_f != 0xffffffffu && k1 >= p_f) { k1 = p_f; e1 = e_f; }
    if (k1 < d && e1 >= hi_th) return k1;
    // every j < k1 has c_j < r; the chain reaches r at the latest where E >= hi_th, and E grows by >= 1 per element
    const uint64_t km = (uint64_t)k1 + (uint64_t)(hi_th > lo_th ? hi_th - lo_th : 0u) + 2ull;
    ls.kmax = km < d ? (uint32_t)km : d;
    ls.k1 = k1;
    ls.f = f_below;
    ls.shifts = c.sh_in | (c.sh_out << 8) | (c.sh_prev << 16);
    ls.p_next = p_f;
    return LANE_AMBIGUOUS;
}

PW_HD uint32_t lane_decide(uint32_t d, uint32_t n_in, uint32_t pp, double r, float w_out, float w_prev,
                           const ListView &cl, LaneStep &ls) {
    DecideCtx c;
    if (lane_decide_begin(d, n_in, pp, r, w_out, w_prev, ls, c) == LANE_REDO) return LANE_REDO;
    if (!n_in) return lane_decide_end(d, n_in, pp, c, nullptr, ls);
    // first common neighbour whose exact mass reaches lo_th
    const MassEval ev{pp, c.sh_in, c.sh_out, c.sh_prev};
#if defined(PW_LANES_WIDE_DECIDE) && PW_LANES_WIDE_DECIDE
    const SearchResult sr = list_search_wide(cl, 0u, n_in, ev, (uint64_t)c.lo_th, ls.probes);   // (same result, field by field)
#else
    const SearchResult sr = list_search(cl, 0u, n_in, ev, (uint64_t)c.lo_th, ls.probes);
#endif
    return lane_decide_end(d, n_in, pp, c, &sr, ls);
}

// ---- WEIGHTED rows: the decision from float64 prefix sums, with a rigorous bound on the float32 chain (round 4) ------
// Arbitrary float32 weights leave no exact integer form of the partial sums.  What one thread CAN have cheaply is the
// real prefix sum of the step's values: the value of a neighbour that is neither a common neighbour nor prev (its BASE
// value: fl32(f64(w) / q), node2vec+: fl32(f64(w) * alpha_0)) depends on cur alone, so its float64 prefix sums PQ are a
// per-vertex array; the common neighbours of the arriving entry differ from their base value by deltas whose prefix
// sums DL follow the entry's list; prev is one more delta.  S(k) = PQ[k] + DL[#commons <= k] + [pp <= k] dprev is then
// the real sum of the first k + 1 values (float64 evaluation: relative 1e-13), and the reference's float32 chain
//     v_i = fl32(w'_i / tot),  c_k = fl32(c_{k-1} + v_k)            (sparse_rw.py:89, pecanpy.py:556-557)
// obeys | c_k - S(k) / tot | <= (S(k) / tot) * eps(k),  eps(k) = (k + 2) * 2^-24 * (1 + o(1)):  one relative rounding 2^-24
// per division, one per addition applied to a partial sum that never exceeds the last one (the chain is monotone).
// Candidate k1 = first k with S(k) / tot * (1 + eps) >= r (bisection over the list, then inside the run of non-common
// positions that holds it); the chain is monotone, so c_{k1-1} < r <= c_{k1} -- i.e. hi(k1 - 1) < r and lo(k1) >= r --
// makes k1 the reference's answer.  Otherwise LANE_AMBIGUOUS: the step goes to the wave-per-walk scan (RMAT-20 with
// hashed weights: 10 % of the steps; the bound saturates on rows beyond a few thousand entries).
// Relative drift of the float32 chain after element k: v_i = fl32(w'_i / tot) and k float32 additions of non-negative terms
// give |c_k - S(k) / tot| <= ((1 + u)^(k + 1) - 1) S(k) / tot, u = 2^-24, and (1 + u)^n - 1 <= n u / (1 - n u) for n u < 1
// (valid for EVERY k -- round 4's 1.05 (k + 3) u covered the second-order term only up to k ~ 800 000); n = k + 3 and the
// 1e-9 leave room for the float64 evaluation of S and of the bound itself.  Rows beyond 2^22 elements: no bound (the step is
// left to the exact scan).
PW_HD double weighted_eps(uint32_t k) {
    const double nu = ((double)k + 3.0) * (1.0 / 16777216.0);
    return nu < 0.25 ? nu * (1.0 + 1.34 * nu) * 1.000001 + 1e-9 : 1e300;   // (1 / (1 - x) <= 1 + 1.34 x for x < 0.25: no division)
}

struct PrefixPair {     // of a row's base values: p = their inclusive prefix sum at this element, t = the sum of the prefix sums so far
    double p, t;
};
struct WeightedRow {
    const PrefixPair *pq;   // [d] float64 prefix sums of the base values of cur's row (+ their running sum: one 16-byte load)
    const double *dl;   // [n_in] inclusive prefix sums, in list order, of (step value - base value) of the common neighbours
    double dprev;       // (step value - base value) of prev's element (0: prev is not in the row)
    bool dl_pos;        // the common neighbours' differences are all >= 0 (q >= 1) -- else all <= 0
    PW_HD double pq_at(uint32_t k) const { return pq[k].p; }
    PW_HD double dl_at(uint32_t i) const { return dl[i]; }   // (through the common neighbour of index i: i + 1 of them)
    // Half-width of the interval around S(k) (before the division by tot) that holds tot * c_k.  Round 5: the sharper bound of
    // UnitPrefixRow::margin (below) -- u (1 + 2 gamma) times the SUM OF THE PREFIX SUMS T(k) -- with T's base part from the table
    // (pq[k].t) and the common neighbours' part bounded without their positions: their differences share the sign of q - 1
    // (value >= base for q >= 1: sparse_rw.py:84-86 / 119-125, rounding is monotone), so sum_j DL(#commons(j)) <= (k + 1) DL(f) when
    // they are positive and <= 0 otherwise; prev likewise.  Never above the first-order bound (k + 3) u S(k), which remains the
    // fallback (a negative DL(f) under dl_pos cannot happen and is not trusted).
    PW_HD double margin(uint32_t k, uint32_t f, uint32_t pp, double S) const {
        const double DLf = f ? dl[f - 1u] : 0.0;
        if (dl_pos ? DLf < 0.0 : DLf > 0.0) return S * weighted_eps(k);
        const double kd = (double)k;
        double T = pq[k].t + (dl_pos ? (kd + 1.0) * DLf : 0.0);
        if (pp <= k && dprev > 0.0) T += dprev * (kd - (double)pp + 1.0);
        const double u = 1.0 / 16777216.0, ku = (kd + 3.0) * u;
        if (!(ku < 0.25)) return 1e300;
        const double rho = u * (1.0 + 3.0 * ku) * 1.000001;          // (2 / (1 - ku) < 3 for ku < 0.25)
        return S * (u * 1.000001 + 1e-9) + rho * T;                   // (T <= (k + 1) S: never above the first-order bound)
    }
};
// The same row description in CLOSED FORM for unit weights (round 5: unit graphs whose 1/p or 1/q is not a power of two):
// every neighbour weighs b = fl32(1/q) (1 on the first step of a walk) unless it is a common neighbour (1) or prev (fl32(1/p)),
// so the prefix sums need no tables: PQ[k] = (k + 1) b, DL[i] = (i + 1)(1 - b), dprev = fl32(1/p) - b -- products of a
// float32 by an integer below 2^24, exact in float64.
struct UnitPrefixRow {
    double b, db;       // base value, (1 - base value)
    double dprev;
    PW_HD double pq_at(uint32_t k) const { return ((double)k + 1.0) * b; }
    PW_HD double dl_at(uint32_t i) const { return ((double)i + 1.0) * db; }
    // The SHARPER bound this row admits: the rounding of addition j is relative to the partial sum c_j it produces, so
    //   |c_k - s_k| <= u / (1 - u) * sum_{j = 1..k} c_j <= u (1 + 2 gamma_k) * sum_{j <= k} s_j,    s_j = sum of the rounded quotients,
    // and s_j <= S(j) / tot * (1 + u): the drift is bounded by the SUM OF THE PREFIX SUMS T(k) = sum_{j <= k} S(j), about half of
    // (k + 1) S(k) when the sums grow evenly.  T(k) in closed form needs the positions of the common neighbours; with f of them at
    // positions <= k it is bounded from above WITHOUT the list: sum_{j <= k} #commons(j) lies in [f (f + 1) / 2, (k + 1) f - f (f - 1) / 2]
    // (all as late / as early as distinct positions allow), prev contributes dprev (k - pp + 1) when it is in the prefix (0 when
    // dprev < 0: an upper bound).  Monotone in k (b (k + 2) per position outweighs the commons' term), and never above the
    // first-order bound (k + 1) S(k).  f = common neighbours at positions <= k.
    PW_HD double margin(uint32_t k, uint32_t f, uint32_t pp, double S) const {
        const double kd = (double)k, fd = (double)f;
        double T = 0.5 * b * (kd + 1.0) * (kd + 2.0);
        T += db >= 0.0 ? db * ((kd + 1.0) * fd - 0.5 * fd * (fd - 1.0)) : db * (0.5 * fd * (fd + 1.0));
        if (pp <= k && dprev > 0.0) T += dprev * (kd - (double)pp + 1.0);
        const double u = 1.0 / 16777216.0, ku = (kd + 3.0) * u;
        if (!(ku < 0.25)) return 1e300;
        const double rho = u * (1.0 + 3.0 * ku) * 1.000001;          // (2 / (1 - ku) < 3 for ku < 0.25)
        return S * (u * 1.000001 + 1e-9) + rho * T;
    }
};
template <class Row>
struct BoundedEval {   // upper bound of the chain at the common neighbour i (position P), as order-preserving bits
    const Row *wr;
    uint32_t pp;
    double inv;
    PW_HD uint64_t operator()(uint32_t i, uint32_t P) const {
        const double S = wr->pq_at(P) + wr->dl_at(i) + (pp < P ? wr->dprev : 0.0);
        const double hi = (S + wr->margin(P, i + 1u, pp, S)) * inv + 3e-45 * ((double)P + 1.0);
        return FloatTraits<double>::bits(hi > 0.0 ? hi : 0.0);
    }
};
typedef BoundedEval<WeightedRow> WeightedEval;

// What a step left open by the bound knows about itself (round 6: input of the interval decision lane_tight_values):
// f = common neighbours at positions below k_safe, p_next = position of the first common neighbour at or after k_safe
// (0xffffffff: none), z_abs = bound on | c_{k_safe - 1} - (real sum of the chain's values before k_safe) |.
struct BoundedAmb {
    uint32_t f, p_next;
    double z_abs;
};
// k_safe (out): every partial sum before element k_safe is known to stay below r -- a step left open can start its exact
// scan there (from the recorded chain value before it: walk_sparse.hip.h, CHAIN_CKPT) instead of at element 0.
template <class Row>
PW_HD uint32_t lane_decide_bounded(uint32_t d, uint32_t n_in, uint32_t pp, double r, float tot, const Row &wr,
                                   const ListView &cl, uint32_t &probes, uint32_t &k_safe, BoundedAmb *amb = nullptr) {
    k_safe = 0;
    if (amb) { amb->f = 0; amb->p_next = 0xffffffffu; amb->z_abs = 0.0; }
    if (!(tot > 0.0f) || d == 0u) return LANE_REDO;
    const double inv = 1.0 / (double)tot;
    const uint64_t tbits = FloatTraits<double>::bits(r > 0.0 ? r : 0.0);
    uint32_t ks = 0, ke = d, f = 0;
    if (n_in) {
        const BoundedEval<Row> ev{&wr, pp, inv};
        const SearchResult sr = list_search(cl, 0u, n_in, ev, tbits, probes);
        f = sr.f;
        if (sr.has_below) ks = sr.p_below + 1u;
        if (sr.f < n_in) ke = sr.p_at;
    }
    const double dbase = f ? wr.dl_at(f - 1u) : 0.0;
    auto sum_at = [&](uint32_t k, uint32_t commons) -> double {   // S(k) with `commons` common neighbours at positions <= k
        return wr.pq_at(k) + (commons ? wr.dl_at(commons - 1u) : 0.0) + (pp <= k ? wr.dprev : 0.0);
    };
    // (commons: common neighbours at positions <= k)
    auto hi_of = [&](uint32_t k, double S, uint32_t commons) { return (S + wr.margin(k, commons, pp, S)) * inv + 3e-45 * ((double)k + 1.0); };
    auto lo_of = [&](uint32_t k, double S, uint32_t commons) { return (S - wr.margin(k, commons, pp, S)) * inv - 3e-45 * ((double)k + 1.0); };
    // first position of the run [ks, ke) whose upper bound reaches r (none: ke -- the common neighbour there, or d)
    uint32_t lo = ks, hi = ke;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const double S = wr.pq_at(mid) + dbase + (pp <= mid ? wr.dprev : 0.0);
        probes++;
        if (hi_of(mid, S, f) >= r) hi = mid; else lo = mid + 1u;
    }
    const uint32_t k1 = lo;
    // the chain is monotone: c_{k1 - 1} < r <= c_{k1} settles it
    if (k1 > 0u) {
        const uint32_t kp = k1 - 1u;                       // (>= ks - 1: `f` common neighbours at positions <= kp)
        if (!(hi_of(kp, sum_at(kp, f), f) < r)) return LANE_AMBIGUOUS;
    }
    k_safe = k1;                                           // (c_j <= c_{k1 - 1} < r for every j < k1: the chain is monotone)
    if (k1 >= d) return d;                                 // never reached: the mirrored overflow read (choice == degree)
    const uint32_t commons = (k1 == ke && f < n_in) ? f + 1u : f;
    if (!(lo_of(k1, sum_at(k1, commons), commons) >= r)) {
        if (amb && k1 > 0u) {
            // c_{k1 - 1} lies within margin / tot of S(k1 - 1) / tot, and the real sum of the chain's VALUES fl32(w / tot) within
            // 2^-24 relative of that quotient (every value is rounded once, all are positive)
            const uint32_t kp = k1 - 1u;
            const double S = sum_at(kp, f);
            amb->f = f;
            amb->p_next = f < n_in ? ke : 0xffffffffu;
            amb->z_abs = (wr.margin(kp, f, pp, S) + S * (1.001 / 16777216.0)) * inv + 3e-45 * ((double)kp + 1.0);
        }
        return LANE_AMBIGUOUS;
    }
    return k1;
}
PW_HD uint32_t lane_decide_weighted(uint32_t d, uint32_t n_in, uint32_t pp, double r, float tot, const WeightedRow &wr,
                                    const ListView &cl, uint32_t &probes, uint32_t &k_safe) {
    return lane_decide_bounded(d, n_in, pp, r, tot, wr, cl, probes, k_safe);
}
// ... for a unit-weight row: w_out = fl32(1/q) (1.0f on the first step of a walk), w_prev = fl32(1/p), tot = the reference's
// sequential float32 row total (w.sum(), sparse_rw.py:89)
PW_HD uint32_t lane_decide_unit_bounded(uint32_t d, uint32_t n_in, uint32_t pp, double r, float tot, float w_out, float w_prev,
                                        const ListView &cl, uint32_t &probes, uint32_t &k_safe, BoundedAmb *amb = nullptr) {
    const UnitPrefixRow row{(double)w_out, 1.0 - (double)w_out, pp != 0xffffffffu ? (double)w_prev - (double)w_out : 0.0};
    return lane_decide_bounded(d, n_in, pp, r, tot, row, cl, probes, k_safe, amb);
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
constexpr uint32_t LANE_TIE_PENDING = 0xfffffff8u;   // (lane_chain with a ChainResume: stopped in front of a rounding-tie binade)
// State of a chain that stopped in front of a binade with a rounding tie (lanes_chain_kernel, round 6: the wavefront walks that
// binade TOGETHER -- 64 list entries per trip, a parity-function scan -- and the chain goes on behind it): the float32 sum,
// the next element and the number of common neighbours before it.
struct ChainResume {
    float c;
    uint32_t k, i0;
    uint32_t started;   // 0: a fresh chain (the head runs); 1: resume at (c, k, i0)
    uint32_t yield;     // 1: return LANE_YIELD behind every iteration of the binade loop (the wavefront's chains advance in step,
                        //    so that a chain that went on behind its tying binade does not run the rest of its binades alone)
};
constexpr uint32_t LANE_YIELD = 0xfffffff7u;

#if !defined(__HIP_DEVICE_COMPILE__)
// host-side instrumentation of lane_chain (self test): elements added one by one after the head, binade iterations
static thread_local uint64_t g_lane_seq_elems = 0, g_lane_binades = 0;
#define PW_LANE_STAT(x) x
#else
#define PW_LANE_STAT(x)
#endif
#if defined(PW_LANES_WATCHDOG) && defined(__HIPCC__)
__device__ unsigned long long g_wd[32];   // debug builds: state of a wavefront / thread whose loop ran away
#endif
constexpr uint32_t LANE_HEAD = 32;        // leading elements added one by one
#ifndef PW_LANE_TIE_BUDGET
#define PW_LANE_TIE_BUDGET 4096
#endif
constexpr uint32_t LANE_TIE_BUDGET = PW_LANE_TIE_BUDGET;  // runs walked one by one inside binades with a rounding tie

struct ChainEval {   // partial sum (in ulps of the current binade) after common neighbour i at position P
    uint64_t C, ii, io;
    uint32_t k, i0, lim;
    PW_HD uint64_t operator()(uint32_t i, uint32_t P) const {
        if (P >= lim) return ~0ull;   // outside the examined prefix
        const uint32_t cin = i - i0 + 1u;
        return C + (uint64_t)cin * ii + (uint64_t)((P - k + 1u) - cin) * io;
    }
};

// c_end (optional): receives the float32 sum of the kend elements when no partial sum reaches r (LANE_CHAIN_END) -- with
// r = +infinity the routine is the reference's sequential row total, w.sum() (src/pecanpy/rw/sparse_rw.py:89).
// WIDE: the per-binade searches probe three entries per level (list_search_wide) -- for the FLOATS form of the lane
// kernel, where every step is two chains and their dependent probes are what a step waits for (+5 %); the chain
// kernels of the dyadic path run ~60 probes per chain at full occupancy and lose 3 % of a pass to the extra traffic.
template <bool WIDE = false>
// tie_budget_init: runs a binade with a rounding tie may be walked by (0: return LANE_TIE at the first such binade --
// lanes_chain_kernel's first pass, which leaves those chains to a second, densely packed launch).
PW_HD uint32_t lane_chain(uint32_t kend, uint32_t n_in, uint32_t pp, double r, float x_in, float x_out, float x_prev,
                          const ListView &cl, uint32_t &reads, float *c_end = nullptr, uint32_t tie_budget_init = LANE_TIE_BUDGET,
                          ChainResume *rs = nullptr) {
    using B = Binade<float>;
    float c = 0.0f;
    uint32_t k = 0;    // next element to add
    uint32_t i0 = 0;   // number of common neighbours before k
    const bool resumed = rs != nullptr && rs->started != 0u;
    if (resumed) { c = rs->c; k = rs->k; i0 = rs->i0; }
    // cursor over the list: position of common neighbour i0, served from a cached 4-entry window so that walking
    // the list costs one (dependent) load per four entries instead of one per entry
    ListWin cw = {{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}};
    uint32_t cw0 = 0xffffffffu, next_in = 0xffffffffu;
    reads = 0;   // list entries read (statistics)
#define PW_LANE_CURSOR()                                                              \
    do {                                                                              \
        if (i0 >= n_in) next_in = 0xffffffffu;                                        \
        else {                                                                        \
            if ((i0 & ~3u) != cw0) { cw = cl.window(i0); cw0 = i0 & ~3u; reads += 4; } \
            const uint32_t o_ = i0 & 3u;                                              \
            next_in = o_ == 0 ? cw.v[0] : (o_ == 1 ? cw.v[1] : (o_ == 2 ? cw.v[2] : cw.v[3])); \
        }                                                                             \
    } while (0)
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
    if (!resumed) {
        PW_LANE_SEQ(LANE_HEAD, 0);
        if (hit) return k;
    }
    uint32_t tie_budget = tie_budget_init;
#if defined(PW_LANES_WATCHDOG) && defined(__HIP_DEVICE_COMPILE__)
    uint32_t wd_chain = 0;
#endif
    while (k < kend) {
#if defined(PW_LANES_WATCHDOG) && defined(__HIP_DEVICE_COMPILE__)
        if (++wd_chain > 1000000u) {
            if (atomicAdd(&g_wd[16], 1ull) == 0ull) {
                g_wd[17] = k; g_wd[18] = kend; g_wd[19] = n_in; g_wd[20] = pp; g_wd[21] = i0; g_wd[22] = next_in;
                g_wd[23] = (unsigned long long)__float_as_uint(c); g_wd[24] = (unsigned long long)__double_as_longlong(r);
                g_wd[25] = (unsigned long long)__float_as_uint(x_in); g_wd[26] = (unsigned long long)__float_as_uint(x_out);
            }
            return LANE_TIE;
        }
#endif
#define PW_LANE_YIELD() do { if (rs != nullptr && rs->yield) { rs->c = c; rs->k = k; rs->i0 = i0; rs->started = 1u; return LANE_YIELD; } } while (0)
        if (k == pp) {   // prev is a single element: always a real addition (no closed form, no tie question)
            PW_LANE_SEQ(1u, 0);
            if (hit) return k;
            PW_LANE_YIELD();
            continue;
        }
        const uint32_t lim = (pp != 0xffffffffu && pp > k && pp < kend) ? pp : kend;   // closed form over [k, lim)
        const int eb = B::eb_of(c);
        const uint64_t C = B::sig_of(c);
        const uint64_t Tt = B::threshold(r, eb);   // <= TOP
        const Inc<float> qi = B::quantize(x_in, eb), qo = B::quantize(x_out, eb);
        if ((qi.a0 != qi.a1 && next_in < lim) || qo.a0 != qo.a1) {
            if (rs) { rs->c = c; rs->k = k; rs->i0 = i0; rs->started = 1u; return LANE_TIE_PENDING; }   // (the wavefront walks this binade)
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
                    // smallest t in [1, m] with Cc + first + (t - 1) * each >= Tt -- the division only for the ONE run that gets
                    // there (the sums grow: a run whose last element stays below Tt has no such t); round 6: this loop runs once
                    // per common neighbour of a tying binade -- thousands of times on a hub row -- and was 7.7 of the chain
                    // kernels' 17.6 ms per RMAT-22 pass
                    uint64_t t = 0xffffffffull;
                    if (Cc + first + (uint64_t)(m - 1u) * each >= Tt) {
                        t = 1;
                        if (Cc + first < Tt) t = each ? 2ull + div_floor_small(Tt - (Cc + first) - 1ull, each) : 0xffffffffull;
                    }
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
                c = B::make((uint32_t)Cc, eb);
                if (lim == kend) { if (c_end) *c_end = c; return LANE_CHAIN_END; }
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
        // starts at s_run with partial sum `base`
        const ChainEval ev{C, ii, io, k, i0, lim};
        const SearchResult sr = WIDE ? list_search_wide(cl, i0, n_in, ev, Tt, reads) : list_search(cl, i0, n_in, ev, Tt, reads);
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
                c = B::make((uint32_t)(base + (uint64_t)(lim - s_run) * io), eb);
                if (lim == kend) { if (c_end) *c_end = c; return LANE_CHAIN_END; }
                k = lim;
                i0 = lo;
                PW_LANE_YIELD();
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
        if (k < kend) PW_LANE_YIELD();
        PW_LANE_CURSOR();
    }
    if (c_end) *c_end = c;
    return LANE_CHAIN_END;
#undef PW_LANE_YIELD
#undef PW_LANE_SEQ
#undef PW_LANE_CURSOR
}

// ---- interval decision of an ambiguous step: the chain's drift bounded WITHOUT touching the list -------------------
// The chain's roundings are SYSTEMATIC: while the sum stays in binade e, adding a value of class c moves it by
// inc_{e,c} ulps exactly (no tie), i.e. errs by the constant delta_{e,c} = inc_{e,c} * ulp_e - x_c; the only other
// roundings are the one addition per binade that crosses its top (|error| <= ulp/2 of the binade entered).  The
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
PW_HD uint32_t lane_tight_tail(uint32_t pp, double r, float x_in, float x_out, float x_pv, uint32_t k1, uint32_t i1, uint32_t pv0,
                               uint32_t kend, uint32_t p_f, int e_t, uint32_t V, float fr, float voff, float Z, float upu);
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
    return lane_tight_tail(pp, r, x_in, x_out, x_pv, k1, i1, pv0, kend, p_f, e_t, V, fr, voff, Z, upu);
}

// The value-generic part of the interval decision, shared by the dyadic form above (exact integer masses) and the FLOATS form
// below (arbitrary float32 values): given the binade e_t of the exact sum v0 of the values before k1, its integer significand
// V and fraction fr (v0 / ulp_t = V + fr, voff = V - 2^23), the a-priori bound Z on |c_{k1-1} - v0| in ulps of that binade and
// the smallest element value in the same ulps (xmin_u: bounds the number of additions below the evaluated binades).
PW_HD uint32_t lane_tight_tail(uint32_t pp, double r, float x_in, float x_out, float x_pv, uint32_t k1, uint32_t i1, uint32_t pv0,
                               uint32_t kend, uint32_t p_f, int e_t, uint32_t V, float fr, float voff, float Z, float upu) {
    using B = Binade<float>;
    const bool has_pv = pp != 0xffffffffu;
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

// ---- the interval decision for ARBITRARY float32 values (round 6: the FLOATS form -- unit weights, 1/p or 1/q not a power of
// two -- in front of its float chains).  The chain adds x_in = fl32(1 / tot), x_out = fl32(w_out / tot),
// x_pv = fl32(w_prev / tot) (sparse_rw.py:89, pecanpy.py:556-557): three float32 numbers in no exact ratio, so there is no
// integer mass -- but the drift argument above never needed one: it needs the REAL sum v0 of the values before k1 (two products
// of a float32 by a count below 2^24 and one value: each exact in float64, their sum to 2^-52 relative, far below an ulp of the
// float32 chain), the a-priori bound on |c_{k1-1} - v0| (lane_decide_bounded: BoundedAmb::z_abs) and the class counts before k1.
// Everything behind that -- quantised increments per binade, the elimination of the per-binade counts, the continuation
// through the top binade for both ends of the interval -- is lane_tight_tail, the code the dyadic form runs.
PW_HD uint32_t lane_tight_values(uint32_t d, uint32_t pp, double r, float x_in, float x_out, float x_pv, uint32_t k1, uint32_t i1,
                                 uint32_t p_f, double z_abs) {
    if (k1 == 0 || k1 >= d) TIGHT_BAIL(1);
    const bool has_pv = pp != 0xffffffffu;
    const uint32_t pv0 = (has_pv && pp < k1) ? 1u : 0u;
    if (i1 + pv0 > k1) TIGHT_BAIL(2);
    const uint32_t o0 = k1 - i1 - pv0;
    if (o0 == 0) TIGHT_BAIL(2);                             // (the elimination runs over the "out" class)
    if (!(x_in > 0.0f) || !(x_out > 0.0f) || (has_pv && !(x_pv > 0.0f))) TIGHT_BAIL(3);
    const double v0 = (double)i1 * (double)x_in + (double)o0 * (double)x_out + (double)pv0 * (double)x_pv;
    if (!(v0 > 0.0) || !(v0 < 2.0)) TIGHT_BAIL(3);
    // binade of v0 as a float32 number: biased exponent e_t, v0 / ulp_t = V + fr with V in [2^23, 2^24)
    const int e_t = (int)((FloatTraits<double>::bits(v0) >> 52) & 0x7ffu) - 1023 + 127;
    if (e_t < 2 || e_t > 126) TIGHT_BAIL(3);
    const double scaled = v0 * FloatTraits<double>::from_bits((uint64_t)(1023 + 150 - e_t) << 52);   // exact: v0 * 2^(150 - e_t)
    if (!(scaled >= 8388608.0) || !(scaled < 16777216.0)) TIGHT_BAIL(3);
    const uint32_t V = (uint32_t)scaled;
    const float fr = (float)(scaled - (double)V);
    const float voff = (float)(V - (1u << 23));
    const double ulp_inv = FloatTraits<double>::from_bits((uint64_t)(1023 + 150 - e_t) << 52);
    const double zu = z_abs * ulp_inv * 1.01 + 2e-9 * scaled + 1.0;    // (+ the float64 evaluation of v0 itself)
    if (!(zu < 4194304.0)) TIGHT_BAIL(4);
    float xmin = x_in < x_out ? x_in : x_out;
    if (pv0 && x_pv < xmin) xmin = x_pv;
    const float upu = (float)((double)xmin * ulp_inv) * 0.9999f;      // smallest value among the prefix's classes, ulps of the top binade
    if (!(upu > 0.0f)) TIGHT_BAIL(3);
    return lane_tight_tail(pp, r, x_in, x_out, x_pv, k1, i1, pv0, d, p_f, e_t, V, fr, voff, (float)zu * 1.0001f, upu);
}

#undef TIGHT_BAIL
}  // namespace pw
