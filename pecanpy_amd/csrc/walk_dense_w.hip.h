// walk_dense_w.hip.h -- DenseOTF on a WEIGHTED dense graph: the float64-bounded decision, one row stream per step (gfx950).
//
// The reference's step (rw/dense_rw.py:34-118 + pecanpy.py:597-612): bias the float64 row of `cur` (node2vec: / q for the
// columns that are not neighbours of `prev`, / p for `prev` itself; node2vec+: * alpha(t), t = data[prev, x] / thr[x]), then
//   tot = w.sum();  cdf = np.cumsum(w / tot);  k = np.searchsorted(cdf, r)
// with both reductions naive left-to-right float64 loops under Numba.  The complete kernel (walk_kernel<double, true, ...>,
// walk_sparse.hip.h) reproduces that chain rounding by rounding: two passes over the row, parity-function scans per binade.
// This kernel does what seqscan.h's lane_decide_bounded does for float32 rows, in float64 and by a whole wavefront:
//
//   * ONE pass over the compressed row of cur (uint32 column + float64 weight per non-zero, coalesced) computes every biased
//     value e_j EXACTLY as the reference does (same divisions / products, -ffp-contract=off) and sums them in ANY order
//     (per lane, per block of 256 elements -- 128 for node2vec+ --, block prefixes P[b] kept in LDS).  Membership of a column in prev's row is one bit of
//     prev's packed row (N/8 bytes, staged in LDS once per step); node2vec+ finds data[prev, x] at the RANK of bit x in that
//     row (prefix popcounts per word in LDS): prev's float64 weights are gathered in ascending order, its columns never read.
//   * Every value is non-negative, so a sum in ANY order in which no term passes through more than m additions errs by at
//     most (1+u)^m - 1 relative, u = 2^-53.  With S(k) the real prefix sums and TOT = S(n-1):  the reference's
//     tot_ref = TOT (1+b), 1+b a product of at most n-1 factors (1 +- u);  its chain c_k = S(k) / tot_ref * (1+a_k), at most
//     k+1 factors (one rounding per quotient, one per addition applied to a partial sum that only grows);  this kernel's
//     S~(k) = S(k)(1+s_k) and TOT~ = TOT (1+t), at most nblk + 2 B + 12 and nblk + B + 6 factors (B additions per lane and
//     block of B x 64 elements, 6 levels of the wave sum, nblk block prefixes; in the block scan B - 1 + 6 + 1 more);
//     T = fl(r * TOT~), one.  Hence
//         c_k >= r  <=>  S~(k) >= T * F,   |F - 1| <= E = (2 n + 2 nblk + 3 B + 20) u (1 + 2^-20)
//     The chain is monotone, so with k1 the first element of the block
//     scan whose S~ reaches T (1 - E):  S~(k1 - 1) < T (1 - E)  and  S~(k1) >= T (1 + E)  prove  c_{k1-1} < r <= c_{k1}, i.e.
//     k1 is what np.searchsorted returns.  A partial sum inside [T (1 - E), T (1 + E)) -- probability ~ 2 n^2 u per step: 10^-8
//     at n = 5 000 non-zeros, one walk in 10^6 -- a negative / non-finite value, or a draw no partial sum reaches (the
//     reference then reads past the row) hands the WALK to the complete kernel through the redo list, as
//     walk_dense_fast_kernel does.  Round 6, second half: a step whose partial sum falls inside the interval is decided IN PLACE
//     by the reference's two loops themselves (values 64 at a time, the float64 additions one after the other in row order:
//     ~40 us of one wavefront) -- one redone walk cost ~10 ms of the complete kernel behind a 140 ms pass; only negative /
//     non-finite values and the read past the row still go to the redo list.
//   * the block that holds k1 (256 elements, 3 KB) is read a second time and scanned; nothing else is read twice.
//
// Declared bytes per step: 12 d(cur) + N / 8 (+ 8 d(prev) for node2vec+) + 8 (draw) + 4 (output).
#pragma once
#include "walk_sparse.hip.h"

namespace pw {

#ifndef PW_DWB
#define PW_DWB 4       // 64-element iterations per block, node2vec (HBM bound at 2 .. 8: ER-20k 149.0 / 149.5 / 151.3 ms per pass)
#endif
#ifndef PW_DWB_EXT
#define PW_DWB_EXT 2   // ... node2vec+ (the gathers behind the column load make it latency bound: 70 VGPRs and seven wavefronts per
#endif                 // SIMD at 2, 84 / five at 4, 112 / four at 8: ER-20k 274 / 286 / 321 ms per pass)
constexpr uint32_t DWBLK_MIN = WAVE;          // smallest block any instantiation uses (host: capacity of the block-prefix array)

struct DenseWArgs {
    const uint32_t *__restrict__ indptr;
    const uint32_t *__restrict__ indices;
    const double *__restrict__ data;
    const uint64_t *__restrict__ adjbits;     // [n][wpr]
    const float *__restrict__ thr;            // node2vec+ thresholds (EXTEND)
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
    uint32_t *redo_list;
    unsigned long long *redo_count;
    uint32_t redo_every;                      // tests: every k-th walk is handed over at its third step
    uint32_t exact_every;                     // tests: steps with (job + step) % k == 0 skip the bounded decision (the in-kernel chain decides)
    uint32_t lds_blocks;                      // capacity of the block-prefix array (>= max degree / DWBLK_MIN + 2)
};

__device__ __forceinline__ double dw_wave_sum(double v) {
    return readlane_f64(wave_incl_scan_f64(v), WAVE - 1);   // ONE value for the whole wavefront: the scan's last lane (6 additions deep)
}

// The biased value of one non-zero of cur's row, statement by statement the reference's arithmetic.
template <bool EXTEND> struct DenseWStep {
    bool has_prev;
    uint32_t prev;
    const uint64_t *pb;              // LDS: prev's packed row
    const uint32_t *pr;              // LDS: set bits before each word of it (EXTEND)
    const double *__restrict__ pdata;   // prev's compressed weights (EXTEND)
    const float *__restrict__ thr;
    double thr_cur;
    double p, q, inv_p, inv_q, one_minus_inv_q, alpha_noisy;
    bool p_pow2, q_pow2;

    __device__ __forceinline__ double div_p(double w) const { return p_pow2 ? w * inv_p : w / p; }
    __device__ __forceinline__ double div_q(double w) const { return q_pow2 ? w * inv_q : w / q; }

    // (col, w) of an element beyond the row's end are (0, 0.0): every branch maps a zero weight to zero
    __device__ __forceinline__ double value(uint32_t col, double w) const {
        if (!has_prev) return w;
        const uint64_t word = pb[col >> 6];
        const bool bit = (word >> (col & 63u)) & 1ull;
        if (!EXTEND) {
            if (col == prev) return div_p(w);                 // dense_rw.py:64
            return bit ? w : div_q(w);                        // dense_rw.py:60-63
        } else {
            double w_px = 0.0;                                // data[prev, col]: zero for a non-neighbour (dense_rw.py:89)
            if (bit) w_px = pdata[pr[col >> 6] + (uint32_t)__popcll(word & ((1ull << (col & 63u)) - 1ull))];
            const double thx = (double)thr[col];
            if (col == prev) return div_p(w);                 // dense_rw.py:94, 111
            if (w_px < thx) {                                 // out edge (dense_rw.py:93)
                const double t = w_px / thx;                  // dense_rw.py:100
                double alpha = inv_q + one_minus_inv_q * t;   // dense_rw.py:105
                if (w < thr_cur) alpha = alpha_noisy;         // dense_rw.py:108-110
                return w * alpha;
            }
            return w;
        }
    }
};

template <bool EXTEND>
__global__ void __launch_bounds__(WAVE)
walk_dense_weighted_kernel(DenseWArgs a) {
    constexpr int DWB = EXTEND ? PW_DWB_EXT : PW_DWB;
    constexpr uint32_t DWBLK = DWB * WAVE;        // elements per block
    extern __shared__ uint64_t dw_lds[];
    uint64_t *pb = dw_lds;                               // [wpr]
    double *P = (double *)(pb + a.wpr);                  // [lds_blocks]
    uint32_t *pr = (uint32_t *)(P + a.lds_blocks);       // [wpr] (EXTEND)
    const int lane = lane_id();
    const uint32_t L = a.L, n = a.n, wpr = a.wpr;
    const uint64_t W = (uint64_t)L + 2;
    const uint64_t n_work = a.job_list ? a.n_list : a.n_jobs;
    unsigned long long st_steps = 0, st_dead = 0, st_exact = 0;

    DenseWStep<EXTEND> sv;
    sv.pb = pb;
    sv.pr = pr;
    sv.thr = a.thr;
    sv.p = a.p;
    sv.q = a.q;
    sv.inv_p = 1.0 / a.p;
    sv.inv_q = 1.0 / a.q;
    sv.one_minus_inv_q = 1.0 - sv.inv_q;
    sv.alpha_noisy = sv.inv_q < 1.0 ? sv.inv_q : 1.0;
    {
        const uint64_t qb = (uint64_t)__double_as_longlong(a.q), pbits = (uint64_t)__double_as_longlong(a.p);
        sv.q_pow2 = (qb & 0xfffffffffffffull) == 0 && a.q > 0x1p-100 && a.q < 0x1p100;
        sv.p_pow2 = (pbits & 0xfffffffffffffull) == 0 && a.p > 0x1p-100 && a.p < 0x1p100;
    }

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
        bool redo = false, dead = false;
        uint32_t j = 1;
        for (; j <= L; j++) {
            const uint32_t rs = uni(a.indptr[cur]), re = uni(a.indptr[cur + 1]);
            const uint32_t d = re - rs;
            if (d == 0) { len_out = j; dead = j > 1; break; }
            const uint32_t jr = (j - 1) & (WAVE - 1);
            if (jr == 0) {
                const uint32_t idx = (j - 1) + (uint32_t)lane;
                rbuf = idx < L ? a.rng[soff + idx] : 0.0;
            }
            const double r = readlane_f64(rbuf, (int)jr);
            const bool has_prev = j >= 2;
            const uint32_t *__restrict__ cols = a.indices + rs;
            const double *__restrict__ wts = a.data + rs;
            const uint32_t nblk = (d + DWBLK - 1) / DWBLK;

            // first block's loads in flight while prev's packed row is staged
            uint32_t c_nx[DWB];
            double w_nx[DWB];
#pragma unroll
            for (int i = 0; i < DWB; i++) {
                const uint32_t k = (uint32_t)i * WAVE + (uint32_t)lane;
                c_nx[i] = k < d ? cols[k] : 0u;
                w_nx[i] = k < d ? wts[k] : 0.0;
            }
            sv.has_prev = has_prev;
            sv.prev = prev;
            if (has_prev) {
                wave_lds_fence();   // (the previous step's readers of pb / pr / P are done)
                const uint64_t *__restrict__ prow = a.adjbits + (uint64_t)prev * wpr;
                uint32_t carry = 0;
                for (uint32_t w0 = 0; w0 < wpr; w0 += WAVE) {
                    const uint32_t w = w0 + (uint32_t)lane;
                    const uint64_t v = w < wpr ? prow[w] : 0ull;
                    if (w < wpr) pb[w] = v;
                    if (EXTEND) {
                        const uint32_t own = (uint32_t)__popcll(v);
                        const uint32_t incl = wave_incl_scan_u32(own);
                        if (w < wpr) pr[w] = carry + incl - own;
                        carry += readlane_u32(incl, WAVE - 1);
                    }
                }
                if (EXTEND) {
                    sv.pdata = a.data + uni(a.indptr[prev]);
                    sv.thr_cur = (double)uni(a.thr[cur]);
                }
            } else {
                wave_lds_fence();
            }
            if (lane == 0) P[0] = 0.0;
            wave_lds_fence();

            // ---- the one pass: block sums, block prefixes into LDS ----
            double run = 0.0;
            bool bad = false;
            for (uint32_t blk = 0; blk < nblk; blk++) {
                uint32_t c_cu[DWB];
                double w_cu[DWB];
#pragma unroll
                for (int i = 0; i < DWB; i++) { c_cu[i] = c_nx[i]; w_cu[i] = w_nx[i]; }
                if (blk + 1 < nblk) {
#pragma unroll
                    for (int i = 0; i < DWB; i++) {
                        const uint32_t k = (blk + 1) * DWBLK + (uint32_t)i * WAVE + (uint32_t)lane;
                        c_nx[i] = k < d ? cols[k] : 0u;
                        w_nx[i] = k < d ? wts[k] : 0.0;
                    }
                }
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < DWB; i++) {
                    const double e = sv.value(c_cu[i], w_cu[i]);
                    bad |= !(e >= 0.0);
                    acc += e;
                }
                run += dw_wave_sum(acc);
                if (lane == 0) P[blk + 1] = run;
            }
            wave_lds_fence();
            const double TOT = run;
            // ---- thresholds of the bounded decision (header) ----
            // (2 d + 2 nblk + 3 DWB + 20 factors (1 +- u) at most: header; + 8 u for the two thresholds' own roundings)
            const double E = ((2.0 * (double)d + 2.0 * (double)nblk + (double)(3 * DWB + 20)) * 0x1p-53) * (1.0 + 0x1p-20) + 8.0 * 0x1p-53;
            const double T = r * TOT;
            const double Tl = T - T * E, Th = T + T * E;
            const bool ok = ballot(bad) == 0ull && TOT > 0.0 && TOT < 0x1p1000;
            uint32_t nxt = NOT_FOUND;
            const bool force_exact = a.exact_every && (job + j) % a.exact_every == 0;   // (test switch)
            if (ok && !force_exact) {
                uint32_t tb = NOT_FOUND;
                for (uint32_t b0 = 0; b0 < nblk && tb == NOT_FOUND; b0 += WAVE) {
                    const uint32_t b = b0 + (uint32_t)lane;
                    const uint64_t m = ballot(b < nblk && P[b + 1] >= Tl);
                    if (m) tb = b0 + (uint32_t)__builtin_ctzll(m);
                }
                if (tb != NOT_FOUND) {
                    double base = P[tb];
#pragma unroll 1
                    for (int i = 0; i < DWB; i++) {
                        const uint32_t k = tb * DWBLK + (uint32_t)i * WAVE + (uint32_t)lane;
                        const uint32_t col = k < d ? cols[k] : 0u;
                        const double w = k < d ? wts[k] : 0.0;
                        const double sc = wave_incl_scan_f64(sv.value(col, w));
                        const double S = base + sc;
                        const uint64_t m = ballot(k < d && S >= Tl);
                        if (m) {
                            const int l = __builtin_ctzll(m);
                            const double Sk = readlane_f64(S, l);
                            const bool first = tb == 0 && i == 0 && l == 0;
                            const bool low_ok = l > 0 || first || base < Tl;
                            if (low_ok && Sk >= Th) nxt = readlane_u32(col, l);
                            break;
                        }
                        base = base + readlane_f64(sc, WAVE - 1);
                    }
                }
            }
            if (nxt == NOT_FOUND && ok) {
                // ---- a partial sum inside the bound's interval: the reference's two loops themselves, in their order ----
                // (values 64 at a time in parallel, the additions one after the other: ~40 us of one wavefront, once per ~10^8
                //  steps; the complete kernel would walk the whole walk again: ~10 ms)
                double tot = 0.0;
                for (uint32_t k0 = 0; k0 < d; k0 += WAVE) {                       // tot = w.sum()  (dense_rw.py:69 / 116)
                    const uint32_t k = k0 + (uint32_t)lane;
                    const double e = sv.value(k < d ? cols[k] : 0u, k < d ? wts[k] : 0.0);
                    const uint32_t m = d - k0 < (uint32_t)WAVE ? d - k0 : (uint32_t)WAVE;
                    for (uint32_t l = 0; l < m; l++) tot = tot + readlane_f64(e, (int)l);
                }
                double c = 0.0;
                for (uint32_t k0 = 0; k0 < d && nxt == NOT_FOUND; k0 += WAVE) {   // cdf = np.cumsum(w / tot); searchsorted (pecanpy.py:609-610)
                    const uint32_t k = k0 + (uint32_t)lane;
                    const uint32_t col = k < d ? cols[k] : 0u;
                    const double v = sv.value(col, k < d ? wts[k] : 0.0) / tot;
                    const uint32_t m = d - k0 < (uint32_t)WAVE ? d - k0 : (uint32_t)WAVE;
                    for (uint32_t l = 0; l < m; l++) {
                        c = c + readlane_f64(v, (int)l);
                        if (c >= r) { nxt = readlane_u32(col, (int)l); break; }
                    }
                }
                st_exact++;
            }
            if (nxt == NOT_FOUND || nxt >= n) { redo = true; break; }   // (no partial sum reaches r: the reference reads past the row)
            if (a.redo_every && j == 3 && job % a.redo_every == 0) { redo = true; break; }
            if (lane == 0) row[j] = nxt;
            prev = cur;
            cur = nxt;
        }
        if (redo) {   // the complete kernel walks this job again (and writes the whole row)
            if (lane == 0) a.redo_list[atomicAdd(a.redo_count, 1ull)] = (uint32_t)job;
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
        if (st_exact) atomicAdd(&a.stats[7], st_exact);   // (pw_stats.ambiguous_steps: steps decided by the float64 chain itself)
    }
}

}  // namespace pw
