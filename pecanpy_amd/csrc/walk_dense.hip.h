// walk_dense.hip.h -- DenseOTF on a bit-packed adjacency matrix, unweighted graphs (gfx950).
//
// The reference keeps a float64[N,N] matrix and a bool[N,N] mask and touches ~1 MB per step at
// N = 100k (rw/dense_rw.py:34-72, pecanpy.py:597-612).  For an unweighted graph all of that is one
// bit per pair: row u = WPR 64-bit words.  A step here works in COLUMN space:
//   in  = row(cur) &  row(prev)          common neighbours           value 1
//   out = row(cur) & ~row(prev)          other neighbours            value 1/q      (float64)
//   prev itself (if adjacent to cur)                                 value 1/p
// are three bitmasks obtained with word-wide AND/ANDN from two coalesced row reads; the exact
// sequential float64 sum / CDF search of the reference is evaluated with the same closed-form
// binade chain as the sparse unit path (seqscan.h, walk_sparse.hip.h: unit_chain) over the columns
// of a 16384-column segment at a time, class counts coming from prefix popcounts (rank arrays) of
// the two masks in LDS.  The sampled element is directly the next vertex (a column index).
// Two kernels: walk_dense_fast_kernel (dyadic 1/p, 1/q, rows <= 131072 columns: the exact-arithmetic decision alone,
// entirely in registers, ONE row read per step -- the row of cur is kept as the next step's prev row) and
// walk_dense_bits_kernel (the complete step: the same decision + the float64 chain behind it; any p, q, any width; also
// walks the jobs the fast kernel hands over).  HBM per step: N/8 bytes (fast) .. 3 N/8 (complete, wide rows) instead of
// ~10 N bytes.
#pragma once
#include <type_traits>
#include "walk_sparse.hip.h"

namespace pw {

constexpr int DW = 512;                  // 32-bit mask words per segment
constexpr uint32_t DSEG = DW * 32;       // columns per segment
constexpr int DQW = DW / 2;              // 64-bit adjacency words per segment
constexpr uint32_t MAX_SEG_COUNTS = 64;  // rows up to 64 segments (1 M columns) keep per-segment class counts

struct DenseArgs {
    const uint64_t *__restrict__ adjbits;  // [n][wpr]
    const uint32_t *__restrict__ deg;      // [n]
    uint32_t n, wpr;
    double p, q;
    uint32_t L;
    uint64_t n_jobs;
    const uint32_t *__restrict__ starts;
    const uint64_t *__restrict__ stream_off;
    const uint32_t *__restrict__ job_list;
    uint64_t n_list;
    const double *__restrict__ rng;
    uint64_t rng_base;
    uint32_t *out;
    unsigned long long *job_counter;
    unsigned long long *stats;
};

struct ColRow {
    const uint32_t *mi, *mo;   // LDS class masks of the segment (prev's own bit cleared in both)
    const uint16_t *ri, *ro;   // LDS rank arrays
    uint32_t seg_lo, seg_len;
    uint32_t prev_col;         // column of prev if it is a neighbour of cur, else NOT_FOUND
    __device__ __forceinline__ uint32_t rank(const uint32_t *m, const uint16_t *rk, uint32_t k) const {
        uint32_t r = k - seg_lo, w = r >> 5, b = r & 31;
        uint32_t base = rk[w];
        return b ? base + (uint32_t)__popc(m[w] & ((1u << b) - 1u)) : base;
    }
    __device__ __forceinline__ uint32_t rank_in(uint32_t k) const { return rank(mi, ri, k); }
    __device__ __forceinline__ uint32_t rank_out(uint32_t k) const { return rank(mo, ro, k); }
    __device__ __forceinline__ uint32_t bit(const uint32_t *m, uint32_t k) const {
        uint32_t r = k - seg_lo;
        return (m[r >> 5] >> (r & 31)) & 1u;
    }
};

// element view for the generic (tie) fallback: one element per COLUMN, zeros for non-neighbours
struct ColVals {
    ColRow cr;
    uint32_t kend;
    double x_in, x_out, x_prev;
    __device__ __forceinline__ double one(uint32_t k) const {
        if (k >= kend) return 0.0;
        if (k == cr.prev_col) return x_prev;
        if (cr.bit(cr.mi, k)) return x_in;
        return cr.bit(cr.mo, k) ? x_out : 0.0;
    }
    __device__ __forceinline__ void vec(uint32_t kb, double (&xs)[EPL]) const {
#pragma unroll
        for (int e = 0; e < EPL; e++) xs[e] = one(kb + e);
    }
};

template <bool HAS_TARGET>
__device__ __forceinline__ int dense_chain(double &c, uint32_t &k, uint32_t kend, double r, const ColRow &cr,
                                           const ColVals &cv, double x_in, double x_out, double x_prev,
                                           uint32_t &found) {
    using B = Binade<double>;
    using U = uint64_t;
    const int lane = lane_id();
    c = uni(c);
    while (k < kend) {
        const int eb = B::eb_of(c);
        const U C = B::sig_of(c);
        const U Tt = HAS_TARGET ? uni(B::threshold(r, eb)) : B::TOP;
        const Inc<double> qi = B::quantize(x_in, eb), qo = B::quantize(x_out, eb), qp = B::quantize(x_prev, eb);
        const bool prev_in = cr.prev_col != NOT_FOUND && cr.prev_col >= k && cr.prev_col < kend;
        if (qi.a0 != qi.a1 || qo.a0 != qo.a1 || (prev_in && qp.a0 != qp.a1)) {
            int rc = seq_scan_binade<double, HAS_TARGET>(c, k, kend, r, cv, found);
            if (rc == SCAN_FOUND) return SCAN_FOUND;
            c = uni(c);
            continue;
        }
        const U ii = qi.a0, io = qo.a0, ipv = qp.a0;
        const uint32_t ri0 = uni(cr.rank_in(k)), ro0 = uni(cr.rank_out(k));
        uint32_t lo = k, hi = kend - 1, kf = 0;
        uint64_t Cf = 0;
        bool crossed = true;
        for (;;) {
            const uint32_t n = hi - lo + 1;
            const uint32_t step = (n + WAVE - 1) / WAVE;
            uint64_t kp64 = (uint64_t)lo + (uint64_t)(lane + 1) * step - 1;
            const uint32_t kp = kp64 > hi ? hi : (uint32_t)kp64;
            const uint32_t cin = cr.rank_in(kp + 1) - ri0;
            const uint32_t cout = cr.rank_out(kp + 1) - ro0;
            const uint32_t cpv = (prev_in && cr.prev_col <= kp) ? 1u : 0u;
            const uint64_t G = C + chain_term<U>(cin, ii) + chain_term<U>(cout, io) + chain_term<U>(cpv, ipv);
            const uint64_t hitm = ballot(G >= Tt);
            if (!hitm) {
                Cf = readlane_u64(G, WAVE - 1);
                crossed = false;
                break;
            }
            const int first = __builtin_ctzll(hitm);
            if (step == 1) { kf = lo + (uint32_t)first; Cf = readlane_u64(G, first); break; }
            uint64_t nhi = (uint64_t)lo + (uint64_t)(first + 1) * step - 1;
            lo = lo + (uint32_t)first * step;
            if (nhi < hi) hi = (uint32_t)nhi;
        }
        if (!crossed) {
            c = uni(B::make((U)Cf, eb));
            k = kend;
            break;
        }
        const bool f_prev = prev_in && kf == cr.prev_col;
        const bool f_in = !f_prev && uni(cr.bit(cr.mi, kf)) != 0u;
        const double xf = f_prev ? x_prev : (f_in ? x_in : x_out);
        const uint32_t cin0 = uni(cr.rank_in(kf)) - ri0;
        const uint32_t cout0 = uni(cr.rank_out(kf)) - ro0;
        const uint32_t cpv0 = (prev_in && cr.prev_col < kf) ? 1u : 0u;
        const uint64_t Cprev = C + chain_term<U>(cin0, ii) + chain_term<U>(cout0, io) + chain_term<U>(cpv0, ipv);
        if (Cf < B::TOP) { c = uni(B::make((U)Cf, eb)); found = kf; return SCAN_FOUND; }
        c = uni(B::make((U)Cprev, eb) + xf);
        k = kf + 1;
        if (HAS_TARGET && (double)c >= r) { found = kf; return SCAN_FOUND; }
    }
    return SCAN_END;
}

// The first neighbours of a row are added one by one (the sum leaves its binade every few elements
// there); all operands are wave uniform, so this runs mostly on the scalar unit.
template <bool HAS_TARGET>
__device__ __forceinline__ bool dense_head(double &c, uint32_t &k, uint32_t kend, double r, const ColRow &cr,
                                           double x_in, double x_out, double x_prev, uint32_t &found) {
    uint32_t cnt = 0;
    uint32_t next_k = k;
    const uint32_t w_end = (kend - cr.seg_lo + 31) >> 5;
    for (uint32_t w = (k - cr.seg_lo) >> 5; w < w_end && cnt < WAVE; w++) {
        const uint32_t bi = uni(cr.mi[w]), bo = uni(cr.mo[w]);
        uint32_t bp = 0;
        if (cr.prev_col != NOT_FOUND && cr.prev_col >= cr.seg_lo && ((cr.prev_col - cr.seg_lo) >> 5) == w)
            bp = 1u << ((cr.prev_col - cr.seg_lo) & 31);
        uint32_t all = bi | bo | bp;
        while (all && cnt < WAVE) {
            const uint32_t b = (uint32_t)__builtin_ctz(all);
            all &= all - 1u;
            const uint32_t col = cr.seg_lo + w * 32u + b;
            const double x = ((bp >> b) & 1u) ? x_prev : (((bi >> b) & 1u) ? x_in : x_out);
            c = uni(c + x);
            cnt++;
            next_k = col + 1;
            if (HAS_TARGET && c >= r) { found = col; return true; }
        }
        if (!all) next_k = cr.seg_lo + (w + 1) * 32u;
    }
    k = next_k > kend ? kend : next_k;
    return false;
}

// Class masks + ranks of one segment of columns [seg_lo, seg_lo + seg_len).
__device__ __forceinline__ void prepare_dense_segment(const uint64_t *__restrict__ crow,
                                                      const uint64_t *__restrict__ prow, bool has_prev,
                                                      uint32_t wpr, uint32_t n, uint32_t seg, uint32_t prev,
                                                      uint32_t *mi, uint32_t *mo, uint16_t *ri, uint16_t *ro) {
    const int lane = lane_id();
    const uint32_t w0 = seg * DQW;
#pragma unroll
    for (int j = 0; j < DQW / WAVE; j++) {
        const uint32_t wl = (uint32_t)j * WAVE + lane;   // 64-bit word inside the segment
        const uint32_t w = w0 + wl;
        uint64_t cw = 0, pw = 0;
        if (w < wpr) {
            cw = crow[w];
            if (has_prev) pw = prow[w];
            const uint64_t col0 = (uint64_t)w * 64;
            if (col0 + 64 > n) cw &= (n > col0) ? ((1ull << (n - col0)) - 1ull) : 0ull;   // columns >= n
            if (has_prev && (prev >> 6) == w) {           // prev forms its own class
                const uint64_t pb = 1ull << (prev & 63);
                cw &= ~pb;
            }
        }
        const uint64_t in = cw & pw, out = cw & ~pw;
        mi[2 * wl] = (uint32_t)in;
        mi[2 * wl + 1] = (uint32_t)(in >> 32);
        mo[2 * wl] = (uint32_t)out;
        mo[2 * wl + 1] = (uint32_t)(out >> 32);
    }
    wave_lds_fence();
    build_rank(mi, ri, DW);
    build_rank(mo, ro, DW);
}

// prepare_dense_segment with the segment's words of both rows already in registers (c4[j], p4[j] = word
// seg * DQW + j * 64 + lane of cur's / prev's row; has_prev)
__device__ __forceinline__ void prepare_dense_segment_regs(const uint64_t (&c4)[4], const uint64_t (&p4)[4], uint32_t wpr, uint32_t n,
                                                           uint32_t seg, uint32_t prev, uint32_t *mi, uint32_t *mo, uint16_t *ri,
                                                           uint16_t *ro) {
    const int lane = lane_id();
    const uint32_t w0 = seg * DQW;
#pragma unroll
    for (int j = 0; j < DQW / WAVE; j++) {
        const uint32_t wl = (uint32_t)j * WAVE + lane;
        const uint32_t w = w0 + wl;
        uint64_t cw = 0, pw = 0;
        if (w < wpr) {
            cw = c4[j];
            pw = p4[j];
            const uint64_t col0 = (uint64_t)w * 64;
            if (col0 + 64 > n) cw &= (n > col0) ? ((1ull << (n - col0)) - 1ull) : 0ull;   // columns >= n
            if ((prev >> 6) == w) cw &= ~(1ull << (prev & 63));                            // prev forms its own class
        }
        const uint64_t in = cw & pw, out = cw & ~pw;
        mi[2 * wl] = (uint32_t)in;
        mi[2 * wl + 1] = (uint32_t)(in >> 32);
        mo[2 * wl] = (uint32_t)out;
        mo[2 * wl + 1] = (uint32_t)(out >> 32);
    }
    wave_lds_fence();
    build_rank(mi, ri, DW);
    build_rank(mo, ro, DW);
}

// WPL > 0 (rows of at most 64 * WPL words): the row of `cur` stays in REGISTERS (WPL 64-bit words per lane) and is the
// row of `prev` one step later -- the count pass then reads ONE row from HBM per step instead of two (round 3; the
// kernel is HBM bound: FETCH_SIZE 1.13 TB per ER-100k pass under PMC, a wide coalesced stream that the counter
// under-counts by 2, i.e. ~5 TB/s).  WPL == 0: both rows from memory (rows beyond 131 072 columns).
template <int WPL>
__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE)
walk_dense_bits_kernel(DenseArgs a) {
    __shared__ uint32_t s_mi[WAVES_PER_BLOCK][DW], s_mo[WAVES_PER_BLOCK][DW];
    __shared__ uint16_t s_ri[WAVES_PER_BLOCK][DW + 2], s_ro[WAVES_PER_BLOCK][DW + 2];
    __shared__ uint32_t s_seg_in[WAVES_PER_BLOCK][MAX_SEG_COUNTS], s_seg_all[WAVES_PER_BLOCK][MAX_SEG_COUNTS];
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    uint32_t *mi = s_mi[wave], *mo = s_mo[wave];
    uint16_t *ri = s_ri[wave], *ro = s_ro[wave];
    uint32_t *seg_in = s_seg_in[wave], *seg_all = s_seg_all[wave];
    const uint32_t L = a.L, n = a.n, wpr = a.wpr;
    const uint64_t W = (uint64_t)L + 2;
    const uint64_t n_work = a.job_list ? a.n_list : a.n_jobs;
    const uint32_t n_seg = (n + DSEG - 1) / DSEG;
    const double w_in = 1.0, w_outq = 1.0 / a.q, w_prevp = 1.0 / a.p;
    unsigned long long st_steps = 0, st_over = 0, st_clamp = 0, st_dead = 0;

    for (;;) {
        unsigned long long widx = 0;
        if (lane == 0) widx = atomicAdd(a.job_counter, 1ull);
        widx = readfirst_u64(widx);
        if (widx >= n_work) break;
        const uint64_t job = a.job_list ? (uint64_t)uni(a.job_list[widx]) : (uint64_t)widx;
        uint32_t *row = a.out + job * W;
        const uint32_t start = uni(a.starts[job]);
        const uint64_t soff = readfirst_u64(a.stream_off[job]) - a.rng_base;
        uint32_t cur = start, prev = 0;
        uint32_t len_out = L + 1;
        double rbuf = 0.0;
        uint64_t keep[WPL > 0 ? WPL : 1];   // row of the vertex the walk was at one step ago (word i * 64 + lane)
        uint64_t cws[WPL > 0 ? WPL : 1];    // row of cur, this step
        uint32_t j = 1;
        for (; j <= L; j++) {
            const uint32_t d = uni(a.deg[cur]);
            if (d == 0) { len_out = j; if (j > 1) st_dead++; break; }
            const uint32_t jr = (j - 1) & (WAVE - 1);
            if (jr == 0) {
                uint32_t idx = (j - 1) + (uint32_t)lane;
                rbuf = idx < L ? a.rng[soff + idx] : 0.0;
            }
            const double r = readlane_f64(rbuf, (int)jr);
            const bool has_prev = j >= 2;
            const uint64_t *__restrict__ crow = a.adjbits + (uint64_t)cur * wpr;
            const uint64_t *__restrict__ prow = a.adjbits + (uint64_t)prev * wpr;

            // class counts over the whole row (for tot)
            uint32_t n_in = 0, n_pv = 0;
            // per-segment class counts (in / all neighbours, prev excluded) for the exact-arithmetic search
            const bool seg_counts = has_prev && n_seg <= MAX_SEG_COUNTS;
            if (WPL > 0 && !has_prev) {   // first step of a walk: its row is the next step's prev row
#pragma unroll
                for (int i = 0; i < (WPL > 0 ? WPL : 1); i++) {
                    const uint32_t w = (uint32_t)i * WAVE + lane;
                    cws[i] = w < wpr ? crow[w] : 0ull;
                }
            }
            if (has_prev) {
                uint32_t acc = 0, acc_in_seg = 0, acc_all_seg = 0;
                if (WPL > 0) {
#pragma unroll
                    for (int i = 0; i < (WPL > 0 ? WPL : 1); i++) {   // every load of the row in flight at once
                        const uint32_t w = (uint32_t)i * WAVE + lane;
                        cws[i] = w < wpr ? crow[w] : 0ull;
                    }
#pragma unroll
                    for (int i = 0; i < (WPL > 0 ? WPL : 1); i++) {
                        const uint32_t w = (uint32_t)i * WAVE + lane;
                        uint64_t cw = cws[i];
                        const uint64_t pw = keep[i];
                        if ((prev >> 6) == w) cw &= ~(1ull << (prev & 63));
                        const uint32_t ci = (uint32_t)__popcll(cw & pw);
                        acc += ci;
                        acc_in_seg += ci;
                        acc_all_seg += (uint32_t)__popcll(cw);
                        // a segment = DQW 64-bit words = DQW / WAVE words per lane
                        if (seg_counts && ((i + 1) % (DQW / WAVE) == 0 || (uint32_t)(i + 1) * WAVE >= wpr)) {
                            uint32_t si = acc_in_seg, sa = acc_all_seg;
#pragma unroll
                            for (int off = 32; off >= 1; off >>= 1) {
                                si += (uint32_t)__shfl_xor((int)si, off, WAVE);
                                sa += (uint32_t)__shfl_xor((int)sa, off, WAVE);
                            }
                            if (lane == 0 && (uint32_t)i * WAVE < wpr) { seg_in[i / (DQW / WAVE)] = si; seg_all[i / (DQW / WAVE)] = sa; }
                            acc_in_seg = acc_all_seg = 0;
                        }
                    }
                } else {
                uint32_t it = 0;
                for (uint32_t w0 = 0; w0 < wpr; w0 += WAVE, it++) {
                    const uint32_t w = w0 + lane;
                    if (w < wpr) {
                        uint64_t cw = crow[w], pw = prow[w];
                        if ((prev >> 6) == w) cw &= ~(1ull << (prev & 63));
                        const uint32_t ci = (uint32_t)__popcll(cw & pw);
                        acc += ci;
                        acc_in_seg += ci;
                        acc_all_seg += (uint32_t)__popcll(cw);
                    }
                    // a segment = DQW 64-bit words = DQW / WAVE iterations of this loop
                    if (seg_counts && ((it + 1) % (DQW / WAVE) == 0 || w0 + WAVE >= wpr)) {
                        uint32_t si = acc_in_seg, sa = acc_all_seg;
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) {
                            si += (uint32_t)__shfl_xor((int)si, off, WAVE);
                            sa += (uint32_t)__shfl_xor((int)sa, off, WAVE);
                        }
                        if (lane == 0) { seg_in[it / (DQW / WAVE)] = si; seg_all[it / (DQW / WAVE)] = sa; }
                        acc_in_seg = acc_all_seg = 0;
                    }
                }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) acc += (uint32_t)__shfl_xor((int)acc, off, WAVE);
                n_in = uni(acc);
                n_pv = (uint32_t)((uni(crow[prev >> 6]) >> (prev & 63)) & 1ull);
                wave_lds_fence();
            }
            const uint32_t n_out = d - n_in - n_pv;
            const uint32_t prev_col = n_pv ? prev : NOT_FOUND;
            const double w_out = has_prev ? w_outq : 1.0;

            double tot = 0.0;
            bool have_tot = false;
            if ((n_out == 0 || is_pow2_fp<double>(w_out)) && (n_pv == 0 || is_pow2_fp<double>(w_prevp))) {
                double u = 1.0;
                if (n_out && w_out < u) u = w_out;
                if (n_pv && w_prevp < u) u = w_prevp;
                double td = (double)n_in + (double)n_out * w_out + (double)n_pv * w_prevp;
                if (td / u <= 9007199254740992.0) { tot = td; have_tot = true; }
            }
            if (!have_tot) {
                for (uint32_t seg = 0; seg < n_seg; seg++) {
                    prepare_dense_segment(crow, prow, has_prev, wpr, n, seg, prev, mi, mo, ri, ro);
                    const uint32_t lo = seg * DSEG, len = n - lo < DSEG ? n - lo : DSEG;
                    const ColRow cr{mi, mo, ri, ro, lo, len, prev_col};
                    const ColVals cv{cr, lo + len, w_in, w_out, w_prevp};
                    uint32_t k = lo, found = NOT_FOUND;
                    if (seg == 0) (void)dense_head<false>(tot, k, lo + len, 0.0, cr, w_in, w_out, w_prevp, found);
                    (void)dense_chain<false>(tot, k, lo + len, 0.0, cr, cv, w_in, w_out, w_prevp, found);
                }
            }
            tot = uni(tot);
            const double x_in = uni(w_in / tot), x_out = uni(w_out / tot), x_prev = uni(w_prevp / tot);

            uint32_t nxt = NOT_FOUND;
            double c = 0.0;
            if (!(r > 0.0)) {
                // (double)c_0 >= r holds at the first neighbour: lowest set bit of the row
                uint32_t best = NOT_FOUND;
                for (uint32_t w = lane; w < wpr && best == NOT_FOUND; w += WAVE) {
                    uint64_t cw = crow[w];
                    if (cw) best = w * 64 + (uint32_t)__builtin_ctzll(cw);
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    uint32_t o = (uint32_t)__shfl_xor((int)best, off, WAVE);
                    best = o < best ? o : best;
                }
                nxt = uni(best);
            } else {
                // Exact-arithmetic decision (same argument as the sparse lazy step, with 2^-53): partial sums of
                // the exact CDF are E(k) / S in units of the smallest weight; the float64 chain cannot differ
                // from them by more than zr units, so when no partial sum lies within zr of R = r * S the answer
                // is the first column with E >= R -- found from the per-segment counts and ONE segment's ranks.
                if (have_tot && has_prev && seg_counts) {
                    double u = 1.0;
                    if (n_out && w_out < u) u = w_out;
                    if (n_pv && w_prevp < u) u = w_prevp;
                    const double S = tot / u, wi = 1.0 / u, wo = w_out / u, wp = w_prevp / u;   // exact: powers of two
                    const double wmax = fmax(wi, fmax(wo, wp)) + 2.0;
                    if (S <= 1099511627776.0 && wmax <= 1048576.0) {
                        const ExactThresholds64 th = exact_thresholds_f64(r * S, (double)d, wmax - 2.0);   // seqscan.h
                        const uint64_t lo_th = th.lo, hi_th = th.hi;
                        const uint64_t Wi = (uint64_t)wi, Wo = (uint64_t)wo, Wp = (uint64_t)wp;
                        const uint32_t pseg = prev_col != NOT_FOUND ? prev_col / DSEG : NOT_FOUND;
                        // segment holding the first column with E >= lo_th
                        uint64_t e0 = 0;
                        uint32_t sx = NOT_FOUND;
                        for (uint32_t sg = 0; sg < n_seg; sg++) {
                            const uint32_t ci = uni(seg_in[sg]), ca = uni(seg_all[sg]);
                            const uint64_t e1 = e0 + (uint64_t)ci * Wi + (uint64_t)(ca - ci) * Wo + (pseg == sg ? Wp : 0ull);
                            if (e1 >= lo_th) { sx = sg; break; }
                            e0 = e1;
                        }
                        if (sx != NOT_FOUND) {
                            if (WPL > 0) {
                                // both rows of the segment are in registers (words 4 sx .. 4 sx + 3 of every lane): no
                                // second trip to memory between the count pass and the search
                                uint64_t c4[4] = {0, 0, 0, 0}, p4[4] = {0, 0, 0, 0};
#define PW_DSEG(S)                                                                                        \
    case S:                                                                                                \
        _Pragma("unroll") for (int jj = 0; jj < 4; jj++) {                                                 \
            c4[jj] = cws[(4 * S + jj) < (WPL > 0 ? WPL : 1) ? (4 * S + jj) : 0];                           \
            p4[jj] = keep[(4 * S + jj) < (WPL > 0 ? WPL : 1) ? (4 * S + jj) : 0];                          \
            if ((4 * S + jj) >= (WPL > 0 ? WPL : 1)) c4[jj] = p4[jj] = 0;                                  \
        }                                                                                                  \
        break;
                                switch (sx) { PW_DSEG(0) PW_DSEG(1) PW_DSEG(2) PW_DSEG(3) PW_DSEG(4) PW_DSEG(5) PW_DSEG(6) PW_DSEG(7) default: break; }
#undef PW_DSEG
                                prepare_dense_segment_regs(c4, p4, wpr, n, sx, prev, mi, mo, ri, ro);
                            } else
                            prepare_dense_segment(crow, prow, has_prev, wpr, n, sx, prev, mi, mo, ri, ro);
                            const uint32_t lo = sx * DSEG, len = n - lo < DSEG ? n - lo : DSEG;
                            const ColRow cr{mi, mo, ri, ro, lo, len, prev_col};
                            uint32_t a0 = lo, a1 = lo + len - 1, kf = NOT_FOUND;
                            uint64_t ef = 0;
                            for (;;) {   // 64-ary search for the first column with E >= lo_th
                                const uint32_t cnt = a1 - a0 + 1, step = (cnt + WAVE - 1) / WAVE;
                                const uint64_t kp64 = (uint64_t)a0 + (uint64_t)(lane + 1) * step - 1;
                                const uint32_t kp = kp64 > a1 ? a1 : (uint32_t)kp64;
                                const uint64_t G = e0 + (uint64_t)cr.rank_in(kp + 1) * Wi + (uint64_t)cr.rank_out(kp + 1) * Wo +
                                                   ((pseg == sx && prev_col <= kp) ? Wp : 0ull);
                                const uint64_t hitm = ballot(G >= lo_th);
                                if (!hitm) break;   // cannot happen: the segment total reaches lo_th
                                const int first = __builtin_ctzll(hitm);
                                if (step == 1) { kf = a0 + (uint32_t)first; ef = readlane_u64(G, first); break; }
                                const uint64_t nhi = (uint64_t)a0 + (uint64_t)(first + 1) * step - 1;
                                a0 += (uint32_t)first * step;
                                if (nhi < a1) a1 = (uint32_t)nhi;
                            }
                            if (kf != NOT_FOUND && ef >= hi_th) nxt = kf;   // decisive; else the float chain below
                        }
                    }
                }
                for (uint32_t seg = 0; seg < n_seg && nxt == NOT_FOUND; seg++) {
                    prepare_dense_segment(crow, prow, has_prev, wpr, n, seg, prev, mi, mo, ri, ro);
                    const uint32_t lo = seg * DSEG, len = n - lo < DSEG ? n - lo : DSEG;
                    const ColRow cr{mi, mo, ri, ro, lo, len, prev_col};
                    const ColVals cv{cr, lo + len, x_in, x_out, x_prev};
                    uint32_t k = lo, found = NOT_FOUND;
                    if (seg == 0 && dense_head<true>(c, k, lo + len, r, cr, x_in, x_out, x_prev, found)) { nxt = found; break; }
                    if (dense_chain<true>(c, k, lo + len, r, cr, cv, x_in, x_out, x_prev, found) == SCAN_FOUND) nxt = found;
                }
            }
            if (nxt == NOT_FOUND) {
                // CDF never reached r: the reference reads past a temporary; clamp to the last neighbour
                st_over++;
                st_clamp++;
                uint32_t best = 0;
                for (uint32_t w = lane; w < wpr; w += WAVE) {
                    uint64_t cw = crow[w];
                    if (cw) { uint32_t col = w * 64 + 63u - (uint32_t)__builtin_clzll(cw); best = col > best ? col : best; }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    uint32_t o = (uint32_t)__shfl_xor((int)best, off, WAVE);
                    best = o > best ? o : best;
                }
                nxt = uni(best);
            }
            if (lane == 0) row[j] = nxt;
            if (WPL > 0) {
#pragma unroll
                for (int i = 0; i < (WPL > 0 ? WPL : 1); i++) keep[i] = cws[i];
            }
            prev = cur;
            cur = nxt;
            st_steps++;
        }
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

// ---- the FAST form ---------------------------------------------------------------------------------------------
// walk_dense_bits_kernel carries the float64 chain (masks + rank arrays in LDS, seq_scan_binade, ...) for the steps the
// exact-arithmetic decision cannot settle -- 2 of 80 M steps at ER-100k -- and for non-dyadic p, q; that code sets the
// kernel's register count (254 VGPRs at WPL = 25: two wavefronts per SIMD, and the walk is a chain of dependent
// operations: occupancy is what hides it).  This kernel is the decisive path ALONE, entirely in registers:
//   count pass   per-lane class counts per 16384-column segment, packed in | out << 16, one interleaved wave sum each
//   tot          from the counts (dyadic weights: exact), thresholds of the exact decision (seqscan.h)
//   search       segment from the sums; inside it the columns are ordered (word group, lane, bit): group by four wave
//                sums, lane by one inclusive scan, bit by a 64-way evaluation of the chosen lane's two words
// No LDS, no masks, no rank arrays.  The first step of a walk (no prev: every neighbour weighs 1) is the same decision
// with an empty prev row.  A walk that meets a step this kernel does not settle (not decisive, r == 0, weights beyond
// the exact range) is put on the redo list and walked again, from its start, by walk_dense_bits_kernel.
// (Register budget: left to the compiler -- 211 VGPRs at WPL = 25, two wavefronts per SIMD.  Forcing three (168) or four
// (128) spills in the step loop: 199 / 57 M steps/s against 396 at ER-100k.  At 396 M steps/s x 12.5 KB per row the kernel
// moves 5 TB/s, four fifths of the achievable HBM rate.)
// FULL: the first FULL word groups lie inside every row this instantiation is launched for (wpr > 64 * FULL): no bounds test.
// LDSK (round 6): the row of the vertex the walk came from waits in LDS (WPL x 512 bytes per wavefront; every lane reads and
// writes its own slots only: no barrier) and is REPLACED word by word by the row of cur inside the count pass, so no row is
// live in registers across the decision; the four words per lane of the target segment are then read back -- cur's from LDS,
// prev's from memory again (2 KB, read one step ago: cache hits).  Three wavefronts per SIMD instead of two.
// BOUNDED (round 6): 1/p or 1/q NOT a power of two.  The class counts are the same; the masses are float64 values
// count_in + count_out * fl(1/q) + [prev] * fl(1/p) (a handful of roundings each, accumulated level by level of the search) and the
// decision is the float64-BOUNDED one of walk_dense_w.hip.h: with T = fl(r * TOT~) and E = (2 d + 64) u, the first column whose
// mass reaches T (1 - E) is np.searchsorted's answer when its mass also reaches T (1 + E) (the reference's tot carries at most
// d - 1 factors (1 +- u), its chain k + 1, the masses here fewer than 24); otherwise -- ~2 d^2 u of the steps -- the walk goes to
// walk_dense_bits_kernel through the redo list like any undecided step.
template <int WPL, int FULL, bool LDSK = false, bool BOUNDED = false>
__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE, LDSK ? 3 : 1)
walk_dense_fast_kernel(DenseArgs a, uint32_t *redo_list, unsigned long long *redo_count, uint32_t redo_every) {
    using M = typename std::conditional<BOUNDED, double, uint64_t>::type;   // a mass: float64 value / integer units of the smallest weight
    constexpr int NSEG = (WPL + 3) / 4;   // a segment = DQW 64-bit words = 4 words per lane
    static_assert(DQW / WAVE == 4, "segment = four words per lane");
    const int lane = lane_id();
    __shared__ uint64_t s_keep[LDSK ? WAVES_PER_BLOCK * WPL * WAVE : 1];
    uint64_t *const lk = s_keep + (LDSK ? (threadIdx.x / WAVE) * (WPL * WAVE) + lane : 0);   // slot i of this lane: lk[i * WAVE]
    const uint32_t L = a.L, n = a.n, wpr = a.wpr;
    const uint64_t W = (uint64_t)L + 2;
    const uint64_t n_work = a.job_list ? a.n_list : a.n_jobs;
    const double w_outq = 1.0 / a.q, w_prevp = 1.0 / a.p;
    unsigned long long st_steps = 0, st_dead = 0;

    for (;;) {
        unsigned long long widx = 0;
        if (lane == 0) widx = atomicAdd(a.job_counter, 1ull);
        widx = readfirst_u64(widx);
        if (widx >= n_work) break;
        const uint64_t job = a.job_list ? (uint64_t)uni(a.job_list[widx]) : (uint64_t)widx;
        uint32_t *row = a.out + job * W;
        const uint32_t start = uni(a.starts[job]);
        const uint64_t soff = readfirst_u64(a.stream_off[job]) - a.rng_base;
        uint32_t cur = start, prev = 0;
        uint32_t len_out = L + 1;
        double rbuf = 0.0;
        uint64_t keep[LDSK ? 1 : WPL];   // row of the vertex the walk was at one step ago (word i * 64 + lane)
        uint64_t cws[WPL];    // row of cur
        if (!LDSK) {
#pragma unroll
            for (int i = 0; i < WPL; i++) keep[LDSK ? 0 : i] = 0ull;
        }
        bool redo = false, dead = false;
        uint32_t j = 1;
        for (; j <= L; j++) {
            const uint32_t d = uni(a.deg[cur]);
            if (d == 0) { len_out = j; dead = j > 1; break; }
            const uint32_t jr = (j - 1) & (WAVE - 1);
            if (jr == 0) {
                const uint32_t idx = (j - 1) + (uint32_t)lane;
                rbuf = idx < L ? a.rng[soff + idx] : 0.0;
            }
            const double r = readlane_f64(rbuf, (int)jr);
            const bool has_prev = j >= 2;
            const uint64_t *__restrict__ crow = a.adjbits + (uint64_t)cur * wpr;
            // (scalar base + this lane's 32-bit byte offset + a constant: ONE address register for the whole row instead of a
            //  64-bit address per load)
            const char *const crow_lane = (const char *)crow + (uint64_t)((uint32_t)lane * 8u);
#pragma unroll
            for (int i = 0; i < WPL; i++) {   // every load of the row in flight at once
                const uint32_t w = (uint32_t)i * WAVE + lane;
                cws[i] = (i < FULL || w < wpr) ? *(const uint64_t *)(crow_lane + (size_t)i * (WAVE * 8)) : 0ull;
            }
            uint32_t n_pv = 0;
            if (has_prev) n_pv = (uint32_t)((uni(crow[prev >> 6]) >> (prev & 63)) & 1ull);
            // count pass (keep == 0 at the first step: everything is "out")
            uint32_t pks[NSEG];
#pragma unroll
            for (int sg = 0; sg < NSEG; sg++) pks[sg] = 0;
#pragma unroll
            for (int i = 0; i < WPL; i++) {
                const uint32_t w = (uint32_t)i * WAVE + lane;
                uint64_t cw = cws[i];
                if (has_prev && (prev >> 6) == w) cw &= ~(1ull << (prev & 63));   // prev forms its own class
                const uint64_t kw = LDSK ? (has_prev ? lk[i * WAVE] : 0ull) : keep[LDSK ? 0 : i];
                const uint32_t ci = (uint32_t)__popcll(cw & kw);
                pks[i / 4] += ci | (((uint32_t)__popcll(cw) - ci) << 16);           // (<= 256 per lane and class)
                if (LDSK) lk[i * WAVE] = cws[i];
            }
            uint32_t n_in = 0, n_out = 0;
#pragma unroll
            for (int sg = 0; sg < NSEG; sg++) {   // (DPP sums: wave.h; round 6 -- 78 ds_bpermute trips per step before)
                pks[sg] = wave_sum_u32(pks[sg]);
                n_in += pks[sg] & 0xffffu;
                n_out += pks[sg] >> 16;
            }
            const uint32_t prev_col = n_pv ? prev : NOT_FOUND;
            const double w_out = has_prev ? w_outq : 1.0;
            // exact total and the exact-arithmetic decision (see walk_dense_bits_kernel)
            double u = 1.0;
            if (n_out && w_out < u) u = w_out;
            if (n_pv && w_prevp < u) u = w_prevp;
            const double td = (double)n_in + (double)n_out * w_out + (double)n_pv * w_prevp;
            const double S = td / u, wi = 1.0 / u, wo = w_out / u, wp = w_prevp / u;   // exact: powers of two
            const double wmax = fmax(wi, fmax(wo, wp)) + 2.0;
            bool ok = BOUNDED ? (n_in + n_out + n_pv == d && r > 0.0 && td > 0.0 && td < 0x1p1000)
                              : ((n_out == 0 || is_pow2_fp<double>(w_out)) && (n_pv == 0 || is_pow2_fp<double>(w_prevp)) &&
                                 n_in + n_out + n_pv == d && S <= 1099511627776.0 && wmax <= 1048576.0 && r > 0.0);
            uint32_t nxt = NOT_FOUND;
            M lo_th = (M)0, hi_th = (M)0, Wi = (M)0, Wo = (M)0, Wp = (M)0, e0 = (M)0;
            uint32_t sx = NOT_FOUND;
            const uint32_t pw_word = prev_col != NOT_FOUND ? (prev >> 6) : NOT_FOUND;   // word of prev's own class
            if (ok) {
                if (BOUNDED) {
                    const double E = ((2.0 * (double)d + 64.0) * 0x1p-53) * (1.0 + 0x1p-20) + 8.0 * 0x1p-53;
                    const double T = r * td;
                    lo_th = (M)(T - T * E); hi_th = (M)(T + T * E);
                    Wi = (M)1.0; Wo = (M)w_out; Wp = (M)w_prevp;
                } else {
                    const ExactThresholds64 th = exact_thresholds_f64(r * S, (double)d, wmax - 2.0);
                    lo_th = (M)th.lo; hi_th = (M)th.hi;
                    Wi = (M)(uint64_t)wi; Wo = (M)(uint64_t)wo; Wp = (M)(uint64_t)wp;
                }
                const uint32_t pseg = pw_word != NOT_FOUND ? pw_word / DQW : NOT_FOUND;
#pragma unroll
                for (int sg = 0; sg < NSEG; sg++) {
                    if (sx == NOT_FOUND) {
                        const M e1 = e0 + (M)(pks[sg] & 0xffffu) * Wi + (M)(pks[sg] >> 16) * Wo +
                                     (pseg == (uint32_t)sg ? Wp : (M)0);
                        if (e1 >= lo_th) sx = (uint32_t)sg; else e0 = e1;
                    }
                }
            }
            // the eight words of the target segment; then the row of cur BECOMES the prev row of the next step: from here
            // on one copy of a row is live (the search works on the words just taken out)
            uint64_t c4[4] = {0, 0, 0, 0}, p4[4] = {0, 0, 0, 0};
            if (LDSK) {
                if (sx != NOT_FOUND) {
                    const uint64_t *__restrict__ prow = a.adjbits + (uint64_t)prev * wpr;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const uint32_t wi = 4u * sx + (uint32_t)jj;
                        const uint32_t w = wi * WAVE + (uint32_t)lane;
                        if (wi < (uint32_t)WPL) {
                            c4[jj] = lk[wi * WAVE];
                            if (has_prev && w < wpr) p4[jj] = prow[w];
                        }
                    }
                }
            } else {
#define PW_DSEG(SG)                                                                  \
    case SG:                                                                         \
        _Pragma("unroll") for (int jj = 0; jj < 4; jj++) {                           \
            if (4 * SG + jj < WPL) { c4[jj] = cws[4 * SG + jj < WPL ? 4 * SG + jj : 0]; p4[jj] = keep[4 * SG + jj < WPL ? 4 * SG + jj : 0]; } \
        }                                                                            \
        break;
                    switch (sx) { PW_DSEG(0) PW_DSEG(1) PW_DSEG(2) PW_DSEG(3) PW_DSEG(4) PW_DSEG(5) PW_DSEG(6) PW_DSEG(7) default: break; }
#undef PW_DSEG
            }
            if (!LDSK) {
#pragma unroll
                for (int i = 0; i < WPL; i++) keep[LDSK ? 0 : i] = cws[i];
            }
            if (sx != NOT_FOUND) {
                {
                    uint64_t in4[4], out4[4];
                    uint32_t pk[4];
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const uint32_t w = (4u * sx + (uint32_t)jj) * WAVE + (uint32_t)lane;
                        uint64_t cw = c4[jj];
                        if (has_prev && (prev >> 6) == w) cw &= ~(1ull << (prev & 63));
                        in4[jj] = cw & p4[jj];
                        out4[jj] = cw & ~p4[jj];
                        pk[jj] = (uint32_t)__popcll(in4[jj]) | ((uint32_t)__popcll(out4[jj]) << 16);
                    }
                    uint32_t sm4[4];   // four independent wave sums
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) sm4[jj] = wave_sum_u32(pk[jj]);
                    uint64_t in_s = 0, out_s = 0;
                    M eg = e0;
                    uint32_t pk_s = 0, g_sel = NOT_FOUND;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        if (g_sel == NOT_FOUND) {   // (wave uniform)
                            const uint32_t sm = uni(sm4[jj]);
                            const uint32_t grp = 4u * sx + (uint32_t)jj;
                            const M e_next = eg + (M)(sm & 0xffffu) * Wi + (M)(sm >> 16) * Wo +
                                             ((pw_word != NOT_FOUND && (pw_word >> 6) == grp) ? Wp : (M)0);
                            if (e_next >= lo_th) { g_sel = grp; in_s = in4[jj]; out_s = out4[jj]; pk_s = pk[jj]; }
                            else eg = e_next;
                        }
                    }
                    if (g_sel != NOT_FOUND) {
                        const uint32_t incl = wave_incl_scan_u32(pk_s);   // inclusive scan over the lanes (the halves cannot carry: sums <= 4096)
                        const bool pv_here = pw_word != NOT_FOUND && (pw_word >> 6) == g_sel;
                        const uint32_t pv_lane = pw_word & 63u;
                        const M G = eg + (M)(incl & 0xffffu) * Wi + (M)(incl >> 16) * Wo +
                                    ((pv_here && pv_lane <= (uint32_t)lane) ? Wp : (M)0);
                        const uint64_t hm = ballot(G >= lo_th);
                        if (hm) {
                            const int l = __builtin_ctzll(hm);
                            // the mass before lane l's word from the EXCLUSIVE counts (no subtraction: a float64 difference of two
                            // masses would carry the rounding of the larger one into the small ones at the start of a row)
                            const uint32_t ex_l = readlane_u32(incl, l) - readlane_u32(pk_s, l);
                            const bool pv_word = pv_here && pv_lane == (uint32_t)l;
                            const M e2 = eg + (M)(ex_l & 0xffffu) * Wi + (M)(ex_l >> 16) * Wo + ((pv_here && pv_lane < (uint32_t)l) ? Wp : (M)0);
                            const uint64_t inw = readlane_u64(in_s, l), outw = readlane_u64(out_s, l);
                            const uint64_t mb = lane == WAVE - 1 ? ~0ull : ((2ull << lane) - 1ull);   // bits 0 .. lane
                            const M Gb = e2 + (M)(uint32_t)__popcll(inw & mb) * Wi + (M)(uint32_t)__popcll(outw & mb) * Wo +
                                         ((pv_word && (prev & 63u) <= (uint32_t)lane) ? Wp : (M)0);
                            const uint64_t hb = ballot(Gb >= lo_th);
                            if (hb) {
                                const int b = __builtin_ctzll(hb);
                                const M Gsel = BOUNDED ? (M)readlane_f64((double)Gb, b) : (M)readlane_u64((uint64_t)Gb, b);
                                if (Gsel >= hi_th) nxt = (g_sel * WAVE + (uint32_t)l) * 64u + (uint32_t)b;   // decisive
                            }
                        }
                    }
                }
            }
            if (nxt == NOT_FOUND || nxt >= n) { redo = true; break; }
            if (redo_every && j == 3 && job % redo_every == 0) { redo = true; break; }   // (test switch: exercises the hand-over)
            if (lane == 0) row[j] = nxt;
            prev = cur;
            cur = uni(nxt);   // (a scalar: the row loads take their base from scalar registers)
        }
        if (redo) {   // walk_dense_bits_kernel walks this job again (and writes the whole row)
            if (lane == 0) redo_list[atomicAdd(redo_count, 1ull)] = (uint32_t)job;
            continue;
        }
        st_steps += (unsigned long long)(j <= L ? j - 1 : L);
        if (dead) st_dead++;
        if (lane == 0) { row[0] = start; row[L + 1] = len_out; }
        for (uint32_t z = j + lane; z <= L; z += WAVE) row[z] = 0;
    }
    if (lane == 0) {
        if (st_steps) atomicAdd(&a.stats[0], st_steps);
        if (st_dead) atomicAdd(&a.stats[3], st_dead);
    }
}

// degrees from the packed rows (one wavefront per row).  The bits of the last word beyond column n - 1 are CLEARED in the
// handle's copy first: the register-only kernels count whole words, and a caller's padding bits would be phantom
// neighbours.
__global__ void __launch_bounds__(256)
dense_degree_kernel(uint64_t *__restrict__ adjbits, uint32_t n, uint32_t wpr, uint32_t *deg) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const uint32_t n_waves = (gridDim.x * blockDim.x) / WAVE;
    const int lane = lane_id();
    const uint64_t tail_mask = (n & 63u) ? ((1ull << (n & 63u)) - 1ull) : ~0ull;
    for (uint32_t u = wave; u < n; u += n_waves) {
        if (lane == 0 && tail_mask != ~0ull) adjbits[(uint64_t)u * wpr + (wpr - 1u)] &= tail_mask;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t acc = 0;
        for (uint32_t w = lane; w < wpr; w += WAVE) {
            const uint64_t v = adjbits[(uint64_t)u * wpr + w];
            acc += (uint32_t)__popcll(w == wpr - 1u ? v & tail_mask : v);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += (uint32_t)__shfl_xor((int)acc, off, WAVE);
        if (lane == 0) deg[u] = acc;
    }
}

}  // namespace pw
