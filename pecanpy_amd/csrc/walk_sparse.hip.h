// walk_sparse.hip.h -- SparseOTF walk kernel for gfx950 (device code).
//
// One 64-lane wavefront walks one job (start vertex) at a time; waves pull jobs from a global
// counter (persistent grid).  Per step (reference SparseOTF.move_forward, src/pecanpy/pecanpy.py:
// 543-559 + SparseRWGraph.get_normalized_probs, src/pecanpy/rw/sparse_rw.py:51-91):
//
//   1. membership  : which neighbours of `cur` are also neighbours of `prev`
//                    (reference: two-pointer isnotin, sparse_rw.py:142-230).  Done here by
//                    binary-searching the *shorter* of the two sorted rows into the longer one
//                    (min(d_cur,d_prev) * log2(max) probes) and recording the result as one bit
//                    per neighbour of `cur` in an LDS bitmask owned by the wave.
//   2. tot         : sequential float32 sum of the biased weights      (sparse_rw.py:89)
//   3. cdf search  : first k with cumsum(w/tot)[k] >= r, sequential float32 (pecanpy.py:556-557)
//   4. next        : indices[indptr[cur] + k], k == degree mirrored     (pecanpy.py:559, App. D)
//
// 2 and 3 use the binade scan of seqscan.h so that the wave-parallel evaluation is bit-identical
// to the reference's left-to-right float32 loops.
#pragma once
#include "seqscan.h"
#include "wave.h"

namespace pw {

struct CsrDev {
    const uint32_t *indptr;
    const uint32_t *indices;
    const float *data;  // nullptr: every weight is 1.0f
    const float *thr;   // node2vec+ thresholds or nullptr
    uint32_t n_nodes;
    uint32_t nnz;
};

struct WalkArgs {
    CsrDev g;
    double p, q;
    uint32_t L;
    uint64_t n_jobs;
    const uint32_t *starts;
    const uint64_t *stream_off;  // per job: absolute index of its first double in the stream
    const uint32_t *job_list;    // optional: run only these jobs (repair passes); nullptr = all
    uint64_t n_list;
    const double *rng;           // rng[n - rng_base]
    uint64_t rng_base;
    uint32_t *out;               // [n_jobs, L + 2]
    unsigned long long *job_counter;
    unsigned long long *stats;   // [0] steps [1] overflow reads [2] clamped reads [3] dead-end walks
};

constexpr int WAVES_PER_BLOCK = 4;
constexpr int MASK_WORDS = 1024;                 // per wave: 32768 neighbours of `cur` per segment
constexpr uint32_t SEG = MASK_WORDS * 32;
constexpr int EPL = 4;                           // elements per lane per scan pass
constexpr uint32_t NOT_FOUND = 0xffffffffu;

// ---- lower_bound over a sorted global row; every lane searches its own key --------------------
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ base, uint32_t n,
                                                    uint32_t key) {
    uint32_t lo = 0, len = n;
    while (len > 0) {  // n is wave-uniform, so the trip count is too
        uint32_t half = len >> 1;
        uint32_t mid = lo + half;
        uint32_t v = base[mid];
        if (v < key) { lo = mid + 1; len -= half + 1; }
        else len = half;
    }
    return lo;
}

// ---- step 1: membership bitmask of one segment [a, a+len) of cur's row ---------------------------
// mask bit (k - a) = 1  <=>  indices[s0 + k] is a neighbour of prev.   Returns (wave-uniform) the
// position of `prev` itself inside the segment, or NOT_FOUND.
__device__ __forceinline__ uint32_t build_mask(const CsrDev &g, uint32_t *mask, uint32_t s0,
                                               uint32_t a, uint32_t len, uint32_t t0, uint32_t dp,
                                               uint32_t prev) {
    const int lane = lane_id();
    const uint32_t *crow = g.indices + s0 + a;
    const uint32_t *prow = g.indices + t0;
    uint32_t prev_pos = NOT_FOUND;
    const uint32_t nwords = (len + 31) >> 5;
    if (dp <= len) {
        // scatter: every neighbour of prev (plus prev itself) looks itself up in cur's segment
        for (uint32_t w = lane; w < nwords; w += WAVE) mask[w] = 0;
        wave_lds_fence();
        for (uint32_t base = 0; base <= dp; base += WAVE) {
            uint32_t i = base + lane;
            bool valid = i <= dp;
            uint32_t y = (i < dp) ? prow[i] : prev;
            uint32_t pos = lower_bound_u32(crow, len, y);
            bool found = valid && pos < len && crow[pos < len ? pos : 0] == y;
            if (found && i < dp) atomicOr(&mask[pos >> 5], 1u << (pos & 31));
            uint64_t pb = ballot(found && i == dp);
            if (pb) prev_pos = a + readlane_u32(pos, __builtin_ctzll(pb));
        }
    } else {
        // gather: every neighbour of cur in the segment looks itself up in prev's row
        for (uint32_t base = 0; base < len; base += WAVE) {
            uint32_t k = base + lane;
            bool valid = k < len;
            uint32_t x = valid ? crow[k] : 0u;
            uint32_t pos = lower_bound_u32(prow, dp, x);
            bool found = valid && pos < dp && prow[pos < dp ? pos : 0] == x;
            uint64_t fb = ballot(found);
            if (lane == 0) mask[base >> 5] = (uint32_t)fb;
            if (lane == 32) mask[(base >> 5) + 1] = (uint32_t)(fb >> 32);
            uint64_t pb = ballot(valid && x == prev);
            if (pb) prev_pos = a + base + __builtin_ctzll(pb);
        }
    }
    wave_lds_fence();
    return prev_pos;
}

// ---- per-neighbour biased weight / probability ------------------------------------------------------
// Values of one segment of cur's row as the reference computes them:
//   w_k = data[k]; out edges: fl32(f64(w)/q); return edge: fl32(f64(w)/p)  (sparse_rw.py:84-87)
//   normalised: fl32(w_k / tot)                                             (sparse_rw.py:89)
template <bool UNIT> struct RowVals {
    const float *drow;     // data + s0 (unused when UNIT)
    const uint32_t *mask;  // LDS bitmask of the current segment
    uint32_t seg_a;        // first neighbour index covered by mask
    uint32_t prev_pos;     // NOT_FOUND when prev is not a neighbour of cur
    uint32_t kend;         // neighbours >= kend contribute 0
    bool has_prev;
    bool normalize;
    double p, q;
    float tot;
    float u_in, u_out, u_prev;  // UNIT: the three possible values (already normalised if asked)

    __device__ __forceinline__ void setup_unit() {
        float w_in = 1.0f, w_out = (float)(1.0 / q), w_prev = (float)(1.0 / p);
        if (normalize) { u_in = w_in / tot; u_out = w_out / tot; u_prev = w_prev / tot; }
        else { u_in = w_in; u_out = w_out; u_prev = w_prev; }
    }

    __device__ __forceinline__ float value(uint32_t k, uint32_t bit) const {
        if (UNIT) {
            if (!has_prev) return u_in;
            return k == prev_pos ? u_prev : (bit ? u_in : u_out);
        } else {
            float w = drow[k];
            if (has_prev) {
                if (k == prev_pos) w = (float)((double)w / p);
                else if (!bit) w = (float)((double)w / q);
            }
            return normalize ? w / tot : w;
        }
    }

    // one element (k < kend required)
    __device__ __forceinline__ float one(uint32_t k) const {
        uint32_t bit = 0;
        if (has_prev) { uint32_t r = k - seg_a; bit = (mask[r >> 5] >> (r & 31)) & 1u; }
        return value(k, bit);
    }

    // EPL consecutive elements starting at kb (kb multiple of EPL); 0 beyond kend
    __device__ __forceinline__ void vec(uint32_t kb, float (&xs)[EPL]) const {
        uint32_t bits = 0;
        if (has_prev && kb < kend) { uint32_t r = kb - seg_a; bits = mask[r >> 5] >> (r & 31); }
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            uint32_t k = kb + e;
            xs[e] = (k < kend) ? value(k, (bits >> e) & 1u) : 0.0f;
        }
    }
};

// ---- steps 2/3: bit-exact sequential running sum, evaluated wave-parallel -----------------------------
template <typename T> __device__ __forceinline__ Inc<T> wave_scan_inc(Inc<T> f) {
    using U = typename FloatTraits<T>::UInt;
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        Inc<T> g;
        g.a0 = shfl_up_uint<U>(f.a0, off);
        g.a1 = shfl_up_uint<U>(f.a1, off);
        Inc<T> h = Binade<T>::compose(g, f);
        if (lane >= off) f = h;
    }
    return f;
}

// Continues the running sum `c` over elements [kbeg, kend) of `vals`.
//   HAS_TARGET: returns the first k with (double)c_k >= r, or NOT_FOUND; c is updated either way.
// head: number of leading elements to add one by one (cheap while the sum doubles every few
// elements and would otherwise leave its binade on almost every pass).
template <typename T, bool HAS_TARGET, typename Vals>
__device__ __forceinline__ uint32_t seq_scan(T &c, uint32_t kbeg, uint32_t kend, double r,
                                             const Vals &vals, uint32_t head) {
    using B = Binade<T>;
    using U = typename B::UInt;
    const int lane = lane_id();
    uint32_t k = kbeg;
    if (head) {
        uint32_t n = kend - kbeg < head ? kend - kbeg : head;
        T v = (lane < (int)n) ? (T)vals.one(kbeg + lane) : (T)0;
        for (uint32_t j = 0; j < n; j++) {
            c = c + readlane_fp<T>(v, (int)j);
            if (HAS_TARGET && (double)c >= r) return kbeg + j;
        }
        k += n;
    }
    while (k < kend) {
        const int eb = B::eb_of(c);
        const U C = B::sig_of(c);
        const U Tt = HAS_TARGET ? B::threshold(r, eb) : B::TOP;
        const uint32_t kb = k & ~(uint32_t)(EPL - 1);
        const uint32_t kl = kb + (uint32_t)lane * EPL;
        T xs[EPL];
        vals.vec(kl, xs);
        Inc<T> f[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            f[e] = B::quantize(xs[e], eb);
            if (kl + e < k) { f[e].a0 = 0; f[e].a1 = 0; }
        }
        Inc<T> agg = f[0];
#pragma unroll
        for (int e = 1; e < EPL; e++) agg = B::compose(agg, f[e]);
        Inc<T> incl = wave_scan_inc<T>(agg);
        U Cincl = B::apply(C, incl);
        uint64_t hit = ballot(Cincl >= Tt);
        if (!hit) {
            c = B::make(readlane_uint<U>(Cincl, WAVE - 1), eb);
            k = kb + WAVE * EPL;
            continue;
        }
        const int fl = __builtin_ctzll(hit);
        U Cprev = fl ? readlane_uint<U>(Cincl, fl - 1) : C;
        U Cn = Cprev;
        int ef = EPL - 1;
        T xf = (T)0;
        bool done = false;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            if (!done) {
                Inc<T> fe;
                fe.a0 = readlane_uint<U>(f[e].a0, fl);
                fe.a1 = readlane_uint<U>(f[e].a1, fl);
                Cn = B::apply(Cprev, fe);
                if (Cn >= Tt) { done = true; ef = e; xf = readlane_fp<T>(xs[e], fl); }
                else Cprev = Cn;
            }
        }
        const uint32_t kf = kb + (uint32_t)fl * EPL + (uint32_t)ef;
        if (Cn < B::TOP) { c = B::make(Cn, eb); return kf; }  // only reachable with a target
        // the sum leaves the binade at element kf: one real floating-point add, then rescan
        c = B::make(Cprev, eb) + xf;
        if (HAS_TARGET && (double)c >= r) return kf;
        k = kf + 1;
    }
    return NOT_FOUND;
}

// ---- one transition --------------------------------------------------------------------------------
// Returns the sampled neighbour *position* k in [0, d] (d == "CDF never reached r").
template <bool UNIT>
__device__ __forceinline__ uint32_t sample_step(const WalkArgs &a, uint32_t *mask, uint32_t cur,
                                                bool has_prev, uint32_t prev, double r,
                                                uint32_t s0, uint32_t d) {
    const CsrDev &g = a.g;
    uint32_t t0 = 0, dp = 0;
    if (has_prev) { t0 = g.indptr[prev]; dp = g.indptr[prev + 1] - t0; }

    RowVals<UNIT> rv;
    rv.drow = UNIT ? nullptr : g.data + s0;
    rv.mask = mask;
    rv.has_prev = has_prev;
    rv.p = a.p;
    rv.q = a.q;
    rv.tot = 1.0f;
    rv.prev_pos = NOT_FOUND;

    const bool multi = has_prev && d > SEG;
    // prev's position is needed by every segment: find it once when the row is segmented
    if (multi) {
        uint32_t pos = lower_bound_u32(g.indices + s0, d, prev);
        pos = readfirst_u32(pos);
        if (pos < d && g.indices[s0 + pos] == prev) rv.prev_pos = pos;
    }

    // pass 1: tot
    float tot = 0.0f;
    rv.normalize = false;
    if (UNIT) rv.setup_unit();
    for (uint32_t sa = 0; sa < d; sa += SEG) {
        uint32_t len = d - sa < SEG ? d - sa : SEG;
        if (has_prev) {
            uint32_t pp = build_mask(g, mask, s0, sa, len, t0, dp, prev);
            if (!multi) rv.prev_pos = pp;
        }
        rv.seg_a = sa;
        rv.kend = sa + len;
        seq_scan<float, false>(tot, sa, sa + len, 0.0, rv, sa == 0 ? WAVE : 0);
    }

    // pass 2: cdf search
    rv.normalize = true;
    rv.tot = tot;
    if (UNIT) rv.setup_unit();
    float c = 0.0f;
    uint32_t choice = NOT_FOUND;
    for (uint32_t sa = 0; sa < d && choice == NOT_FOUND; sa += SEG) {
        uint32_t len = d - sa < SEG ? d - sa : SEG;
        if (multi) (void)build_mask(g, mask, s0, sa, len, t0, dp, prev);  // single segment: still valid
        rv.seg_a = sa;
        rv.kend = sa + len;
        choice = seq_scan<float, true>(c, sa, sa + len, r, rv, sa == 0 ? WAVE : 0);
    }
    return choice == NOT_FOUND ? d : choice;
}

template <bool UNIT>
__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE)
walk_sparse_kernel(WalkArgs a) {
    __shared__ uint32_t s_mask[WAVES_PER_BLOCK][MASK_WORDS];
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    uint32_t *mask = s_mask[wave];
    const CsrDev &g = a.g;
    const uint32_t L = a.L;
    const uint64_t W = (uint64_t)L + 2;
    const uint64_t n_work = a.job_list ? a.n_list : a.n_jobs;

    unsigned long long st_steps = 0, st_over = 0, st_clamp = 0, st_dead = 0;

    for (;;) {
        unsigned long long widx = 0;
        if (lane == 0) widx = atomicAdd(a.job_counter, 1ull);
        widx = readfirst_u64(widx);
        if (widx >= n_work) break;
        const uint64_t job = a.job_list ? (uint64_t)a.job_list[widx] : (uint64_t)widx;
        uint32_t *row = a.out + job * W;
        const uint32_t start = a.starts[job];
        const uint64_t soff = a.stream_off[job] - a.rng_base;

        uint32_t cur = start, prev = 0;
        uint32_t len_out = L + 1;
        uint32_t j = 1;
        for (; j <= L; j++) {
            const uint32_t s0 = g.indptr[cur];
            const uint32_t d = g.indptr[cur + 1] - s0;
            if (d == 0) { len_out = j; if (j > 1) st_dead++; break; }
            const double r = a.rng[soff + (j - 1)];
            const uint32_t choice = sample_step<UNIT>(a, mask, cur, j >= 2, prev, r, s0, d);
            uint64_t pos = (uint64_t)s0 + choice;
            if (choice >= d) {
                st_over++;
                if (pos >= g.nnz) { pos = g.nnz - 1; st_clamp++; }
            }
            const uint32_t nxt = g.indices[pos];
            if (lane == 0) row[j] = nxt;
            prev = cur;
            cur = nxt;
            st_steps++;
        }
        // header, tail zeros and length cell (cells j..L stay 0 after an early stop)
        if (lane == 0) { row[0] = start; row[L + 1] = len_out; }
        for (uint32_t z = j + lane; z <= L; z += WAVE) row[z] = 0;
    }
    if (lane == 0) {
        if (st_steps) atomicAdd(&a.stats[0], st_steps);
        if (st_over) atomicAdd(&a.stats[1], st_over);
        if (st_clamp) atomicAdd(&a.stats[2], st_clamp);
        if (st_dead) atomicAdd(&a.stats[3], st_dead);
    }
}

}  // namespace pw
