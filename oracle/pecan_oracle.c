/*
 * oracle/pecan_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, single stream) of the reference's walk-generation path,
 * written from the behaviour described in SURVEY.md App. A/B/D.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library, and only as the checker.
 * The product path (pecanpy_amd + libpecanpy_amd.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 *   (1) the reference's own known-answer walks (test/test_walk.py:21-82), and
 *   (2) fixtures produced by importing the Python reference in the build container
 *       (tests/golden/make_golden.py, stub-import recipe of SURVEY.md section 8(c)).
 *
 * Semantics restated (Numba rules, not NumPy's -- SURVEY.md App. A.0):
 *   arr.sum()/np.cumsum   : sequential, accumulator in the array dtype
 *   f32_array op= f64     : computed in float64, stored back as float32
 *   np.searchsorted(left) : first index with a[idx] >= v, float32 widened to float64
 *   no bounds checking on array reads (App. D quirk 1)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "mt19937.h"

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* Sparse (CSR) transition probabilities                                                      */
/* ------------------------------------------------------------------------------------------ */

/* Two-pointer scan over two ascending id arrays: flag[k] = 1 when a1[k] does not occur in a2.
 * Follows src/pecanpy/rw/sparse_rw.py:142-230 (isnotin). */
static void orc_isnotin(const uint32_t *a1, uint32_t n1, const uint32_t *a2, uint32_t n2,
                        uint8_t *flag) {
    for (uint32_t k = 0; k < n1; k++) flag[k] = 1;
    uint32_t i2 = 0;
    for (uint32_t i1 = 0; i1 < n1; i1++) {
        if (i2 == n2) break;
        uint32_t v1 = a1[i1], v2 = a2[i2];
        if (v1 < v2) continue;
        if (v1 == v2) {
            flag[i1] = 0;
            i2++;
        } else {
            for (uint32_t j = i2; j < n2; j++) {
                v2 = a2[j];
                if (v2 == v1) { flag[i1] = 0; i2 = j + 1; break; }
                if (v2 > v1) { i2 = j; break; }
            }
        }
    }
}

/* node2vec+ variant: also classifies a common neighbour as an out edge when the edge
 * prev->x is below x's noise threshold, and returns t = w(prev,x)/thr[x] (float32) for it.
 * Follows src/pecanpy/rw/sparse_rw.py:233-295 (isnotin_extended). */
static void orc_isnotin_extended(const uint32_t *a1, uint32_t n1, const uint32_t *a2, uint32_t n2,
                                 const float *w2, const float *thr, uint8_t *flag, float *t) {
    for (uint32_t k = 0; k < n1; k++) { flag[k] = 1; t[k] = 0.0f; }
    uint32_t i2 = 0;
    for (uint32_t i1 = 0; i1 < n1; i1++) {
        if (i2 >= n2) break;
        uint32_t v1 = a1[i1], v2 = a2[i2];
        if (v1 < v2) continue;
        if (v1 == v2) {
            if (w2[i2] >= thr[v2]) flag[i1] = 0;
            else t[i1] = w2[i2] / thr[v2];
            i2++;
        } else {
            for (uint32_t j = i2 + 1; j < n2; j++) {
                v2 = a2[j];
                if (v2 == v1) {
                    if (w2[j] >= thr[v2]) flag[i1] = 0;
                    else t[i1] = w2[j] / thr[v2];
                    i2 = j + 1;
                    break;
                }
                if (v2 > v1) { i2 = j; break; }
            }
        }
    }
}

/* Normalised transition probabilities out of `cur` given `prev` (has_prev=0: first order).
 * thr == NULL : node2vec   (src/pecanpy/rw/sparse_rw.py:51-91)
 * thr != NULL : node2vec+  (src/pecanpy/rw/sparse_rw.py:93-130)
 * Writes d float32 values to w, returns d.  scratch must hold d bytes + d floats. */
static uint32_t orc_sparse_probs_impl(const uint32_t *indptr, const uint32_t *indices,
                                      const float *data, double p, double q, uint32_t cur,
                                      int has_prev, uint32_t prev, const float *thr, float *w,
                                      uint8_t *flag, float *t) {
    uint32_t s0 = indptr[cur], d = indptr[cur + 1] - s0;
    const uint32_t *nb = indices + s0;
    for (uint32_t k = 0; k < d; k++) w[k] = data[s0 + k]; /* get_nbrs copy, sparse_rw.py:133-139 */
    if (has_prev) {
        uint32_t t0 = indptr[prev], dp = indptr[prev + 1] - t0;
        if (thr == NULL) {
            orc_isnotin(nb, d, indices + t0, dp, flag);
            for (uint32_t k = 0; k < d; k++)
                if (nb[k] == prev) flag[k] = 0; /* sparse_rw.py:84 */
            for (uint32_t k = 0; k < d; k++)
                if (flag[k]) w[k] = (float)((double)w[k] / q); /* :86 */
            for (uint32_t k = 0; k < d; k++)
                if (nb[k] == prev) w[k] = (float)((double)w[k] / p); /* :87 */
        } else {
            orc_isnotin_extended(nb, d, indices + t0, dp, data + t0, thr, flag, t);
            for (uint32_t k = 0; k < d; k++)
                if (nb[k] == prev) flag[k] = 0; /* sparse_rw.py:116 */
            double inv_q = 1.0 / q;
            double noisy = inv_q < 1.0 ? inv_q : 1.0; /* np.minimum(1, 1/q), :124 */
            float thr_cur = thr[cur];
            for (uint32_t k = 0; k < d; k++) {
                if (!flag[k]) continue;
                double alpha = inv_q + (1.0 - inv_q) * (double)t[k]; /* :119 */
                if (w[k] < thr_cur) alpha = noisy;                   /* :122-124 */
                w[k] = (float)((double)w[k] * alpha);                /* :125 */
            }
            for (uint32_t k = 0; k < d; k++)
                if (nb[k] == prev) w[k] = (float)((double)w[k] / p); /* :126 */
        }
    }
    float tot = 0.0f;
    for (uint32_t k = 0; k < d; k++) tot += w[k]; /* sequential float32 .sum() */
    for (uint32_t k = 0; k < d; k++) w[k] = w[k] / tot;
    return d;
}

/* cumsum (sequential, float32) + left bisect against a float64 draw; returns d when the CDF
 * never reaches r (pecanpy.py:556-557). */
static uint32_t orc_cdf_search_f32(const float *pr, uint32_t d, double r) {
    float c = 0.0f;
    for (uint32_t k = 0; k < d; k++) {
        c += pr[k];
        if ((double)c >= r) return k;
    }
    return d;
}

typedef struct {
    uint64_t overflow_reads;   /* steps where choice == degree (App. D quirk 1) */
    uint64_t clamped_reads;    /* of those, reads that would have left the buffer (clamped) */
    uint64_t total_steps;      /* sampled transitions */
} orc_stats_t;

ORC_API uint32_t orc_sparse_probs(const uint32_t *indptr, const uint32_t *indices,
                                  const float *data, double p, double q, uint32_t cur,
                                  int has_prev, uint32_t prev, const float *thr, float *out) {
    uint32_t d = indptr[cur + 1] - indptr[cur];
    uint8_t *flag = (uint8_t *)malloc(d + 1);
    float *t = (float *)malloc(sizeof(float) * (d + 1));
    orc_sparse_probs_impl(indptr, indices, data, p, q, cur, has_prev, prev, thr, out, flag, t);
    free(flag);
    free(t);
    return d;
}

static uint32_t orc_max_degree(const uint32_t *indptr, uint32_t n) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t d = indptr[i + 1] - indptr[i];
        if (d > m) m = d;
    }
    return m;
}

static inline uint32_t orc_read_choice(const uint32_t *indptr, const uint32_t *indices,
                                       uint32_t nnz, uint32_t cur, uint32_t choice, uint32_t d,
                                       orc_stats_t *st) {
    uint64_t pos = (uint64_t)indptr[cur] + choice;
    if (choice >= d) {
        st->overflow_reads++;
        if (pos >= nnz) { pos = nnz - 1; st->clamped_reads++; }
    }
    return indices[pos];
}

/*
 * SparseOTF walks, one sequential MT19937 stream in job order
 * (Base._random_walks, src/pecanpy/pecanpy.py:164-210; SparseOTF.move_forward :543-559).
 * out is uint32[n_jobs, L+2]; row = [start, n_1..n_L, len] with unused cells 0 (App. A.6).
 * stream_skip: number of doubles to discard first (lets tests/bench emulate a shard that
 * starts in the middle of the single stream).
 */
ORC_API int orc_walks_sparse_otf(const uint32_t *indptr, const uint32_t *indices,
                                 const float *data, uint32_t n_nodes, double p, double q,
                                 const float *thr, const uint32_t *starts, uint64_t n_jobs,
                                 uint32_t L, uint32_t seed, uint64_t stream_skip, uint32_t *out,
                                 orc_stats_t *stats) {
    orc_stats_t st = {0, 0, 0};
    uint32_t nnz = indptr[n_nodes];
    uint32_t md = orc_max_degree(indptr, n_nodes);
    float *w = (float *)malloc(sizeof(float) * (md + 1));
    float *t = (float *)malloc(sizeof(float) * (md + 1));
    uint8_t *flag = (uint8_t *)malloc(md + 1);
    orc_mt_t rng;
    orc_mt_seed(&rng, seed);
    for (uint64_t s = 0; s < stream_skip; s++) (void)orc_mt_random(&rng);
    const uint64_t W = (uint64_t)L + 2;
    for (uint64_t i = 0; i < n_jobs; i++) {
        uint32_t *row = out + i * W;
        memset(row, 0, sizeof(uint32_t) * W);
        row[0] = starts[i];
        row[L + 1] = L + 1;
        uint32_t cur = row[0];
        if (indptr[cur] == indptr[cur + 1]) { row[L + 1] = 1; continue; }
        for (uint32_t j = 1; j <= L; j++) {
            cur = row[j - 1];
            uint32_t d = indptr[cur + 1] - indptr[cur];
            if (d == 0) { row[L + 1] = j; break; }
            orc_sparse_probs_impl(indptr, indices, data, p, q, cur, j >= 2, j >= 2 ? row[j - 2] : 0,
                                  thr, w, flag, t);
            double r = orc_mt_random(&rng);
            uint32_t choice = orc_cdf_search_f32(w, d, r);
            row[j] = orc_read_choice(indptr, indices, nnz, cur, choice, d, &st);
            st.total_steps++;
        }
    }
    free(w); free(t); free(flag);
    if (stats) *stats = st;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Dense path (float64 N x N + bool mask)                                                     */
/* ------------------------------------------------------------------------------------------ */

/* Dense normalised probabilities over the neighbours of cur (ascending column order).
 * node2vec : src/pecanpy/rw/dense_rw.py:34-72 ; node2vec+ : :74-118.
 * Writes the compressed float64 probabilities to pr and the neighbour columns to cols; returns d. */
static uint32_t orc_dense_probs_impl(const double *data, const uint8_t *nonzero, uint32_t n,
                                     double p, double q, uint32_t cur, int has_prev, uint32_t prev,
                                     const float *thr, double *wfull, double *pr, uint32_t *cols) {
    const double *rc = data + (uint64_t)cur * n;
    const uint8_t *mc = nonzero + (uint64_t)cur * n;
    for (uint32_t x = 0; x < n; x++) wfull[x] = rc[x];
    if (has_prev) {
        const double *rp = data + (uint64_t)prev * n;
        const uint8_t *mp = nonzero + (uint64_t)prev * n;
        if (thr == NULL) {
            for (uint32_t x = 0; x < n; x++)
                if (mc[x] && !mp[x] && x != prev) wfull[x] /= q; /* dense_rw.py:63-66 */
            wfull[prev] /= p;                                    /* :67 */
        } else {
            double inv_q = 1.0 / q;
            double noisy = inv_q < 1.0 ? inv_q : 1.0;
            float thr_cur = thr[cur];
            for (uint32_t x = 0; x < n; x++) {
                /* out_ind = cur_nbrs & (data[prev] < thr)  (:95), prev excluded (:96) */
                if (!(mc[x] && rp[x] < (double)thr[x]) || x == prev) continue;
                double t = rp[x] / (double)thr[x];               /* :102 */
                double alpha = inv_q + (1.0 - inv_q) * t;        /* :107 */
                if (rc[x] < (double)thr_cur) alpha = noisy;      /* :110-112 */
                wfull[x] *= alpha;                               /* :113 */
            }
            wfull[prev] /= p;                                    /* :114 */
        }
    }
    uint32_t d = 0;
    for (uint32_t x = 0; x < n; x++)
        if (mc[x]) { pr[d] = wfull[x]; cols[d] = x; d++; }
    double tot = 0.0;
    for (uint32_t k = 0; k < d; k++) tot += pr[k];
    for (uint32_t k = 0; k < d; k++) pr[k] = pr[k] / tot;
    return d;
}

ORC_API uint32_t orc_dense_probs(const double *data, const uint8_t *nonzero, uint32_t n, double p,
                                 double q, uint32_t cur, int has_prev, uint32_t prev,
                                 const float *thr, double *out_pr, uint32_t *out_cols) {
    double *wfull = (double *)malloc(sizeof(double) * n);
    uint32_t d = orc_dense_probs_impl(data, nonzero, n, p, q, cur, has_prev, prev, thr, wfull,
                                      out_pr, out_cols);
    free(wfull);
    return d;
}

/* DenseOTF walks (DenseOTF.move_forward, src/pecanpy/pecanpy.py:597-612;
 * has_nbrs = any True in the mask row, dense_rw.py:21-32). */
ORC_API int orc_walks_dense_otf(const double *data, const uint8_t *nonzero, uint32_t n, double p,
                                double q, const float *thr, const uint32_t *starts,
                                uint64_t n_jobs, uint32_t L, uint32_t seed, uint64_t stream_skip,
                                uint32_t *out, orc_stats_t *stats) {
    orc_stats_t st = {0, 0, 0};
    double *wfull = (double *)malloc(sizeof(double) * n);
    double *pr = (double *)malloc(sizeof(double) * n);
    uint32_t *cols = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint8_t *has = (uint8_t *)calloc(n, 1);
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t x = 0; x < n; x++)
            if (nonzero[(uint64_t)i * n + x]) { has[i] = 1; break; }
    orc_mt_t rng;
    orc_mt_seed(&rng, seed);
    for (uint64_t s = 0; s < stream_skip; s++) (void)orc_mt_random(&rng);
    const uint64_t W = (uint64_t)L + 2;
    for (uint64_t i = 0; i < n_jobs; i++) {
        uint32_t *row = out + i * W;
        memset(row, 0, sizeof(uint32_t) * W);
        row[0] = starts[i];
        row[L + 1] = L + 1;
        if (!has[row[0]]) { row[L + 1] = 1; continue; }
        for (uint32_t j = 1; j <= L; j++) {
            uint32_t cur = row[j - 1];
            if (!has[cur]) { row[L + 1] = j; break; }
            uint32_t d = orc_dense_probs_impl(data, nonzero, n, p, q, cur, j >= 2,
                                              j >= 2 ? row[j - 2] : 0, thr, wfull, pr, cols);
            double r = orc_mt_random(&rng);
            double c = 0.0;
            uint32_t choice = d;
            for (uint32_t k = 0; k < d; k++) {
                c += pr[k];
                if (c >= r) { choice = k; break; }
            }
            if (choice >= d) { /* reference reads past a temporary: deviation D, clamp */
                st.overflow_reads++;
                st.clamped_reads++;
                choice = d - 1;
            }
            row[j] = cols[choice];
            st.total_steps++;
        }
    }
    free(wfull); free(pr); free(cols); free(has);
    if (stats) *stats = st;
    return 0;
}

/* DenseOTF walks over an UNWEIGHTED dense graph given as bit-packed adjacency rows (bit x of word
 * x / 64 of row i set <=> data[i][x] == 1.0, nonzero[i][x] == True): the same statements as
 * orc_dense_probs_impl / orc_walks_dense_otf above (dense_rw.py:34-72, pecanpy.py:597-612) with the
 * float64 row never materialised -- every stored value is 1.0, so w[x] is 1.0, 1.0 / q (x a neighbour of
 * cur but not of prev, x != prev) or 1.0 / p (x == prev), in ascending column order.  Exists so that the
 * BASELINE C4 shape (N = 100 000, density 0.25: 80 GB as float64) can be checked on the host; pinned to
 * the dense goldens through the float64 entry (tests/test_oracle_golden.py packs their matrices). */
ORC_API int orc_walks_dense_otf_bits(const uint64_t *bits, uint32_t n, uint32_t wpr, double p, double q,
                                     const uint32_t *starts, uint64_t n_jobs, uint32_t L, uint32_t seed,
                                     uint64_t stream_skip, uint32_t *out, orc_stats_t *stats) {
    orc_stats_t st = {0, 0, 0};
    double *pr = (double *)malloc(sizeof(double) * (n + 1));
    uint32_t *cols = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1));
    orc_mt_t rng;
    orc_mt_seed(&rng, seed);
    for (uint64_t s = 0; s < stream_skip; s++) (void)orc_mt_random(&rng);
    const uint64_t W = (uint64_t)L + 2;
    for (uint64_t i = 0; i < n_jobs; i++) {
        uint32_t *row = out + i * W;
        memset(row, 0, sizeof(uint32_t) * W);
        row[0] = starts[i];
        row[L + 1] = L + 1;
        for (uint32_t j = 1; j <= L; j++) {
            uint32_t cur = row[j - 1];
            const uint64_t *mc = bits + (uint64_t)cur * wpr;
            const int has_prev = j >= 2;
            const uint32_t prev = has_prev ? row[j - 2] : 0;
            const uint64_t *mp = bits + (uint64_t)prev * wpr;
            uint32_t d = 0;
            for (uint32_t wd = 0; wd < wpr; wd++) {
                uint64_t m = mc[wd];
                while (m) {
                    const uint32_t b = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t x = wd * 64u + b;
                    if (x >= n) continue;
                    double w = 1.0;                                     /* copy of data[cur], dense_rw.py:57 */
                    if (has_prev) {
                        if (!((mp[wd] >> b) & 1ull) && x != prev) w /= q;  /* :63-66 */
                        if (x == prev) w /= p;                          /* :67 */
                    }
                    pr[d] = w;
                    cols[d] = x;
                    d++;
                }
            }
            if (d == 0) { row[L + 1] = j; break; }                      /* has_nbrs, dense_rw.py:21-32 */
            double tot = 0.0;
            for (uint32_t k = 0; k < d; k++) tot += pr[k];
            double r = orc_mt_random(&rng);
            double c = 0.0;
            uint32_t choice = d;
            for (uint32_t k = 0; k < d; k++) {
                c += pr[k] / tot;
                if (c >= r) { choice = k; break; }
            }
            if (choice >= d) { st.overflow_reads++; st.clamped_reads++; choice = d - 1; }
            row[j] = cols[choice];
            st.total_steps++;
        }
    }
    free(pr); free(cols);
    if (stats) *stats = st;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Alias tables (PreComp / PreCompFirstOrder) and first-order modes                           */
/* ------------------------------------------------------------------------------------------ */

/* alias_setup, src/pecanpy/pecanpy.py:617-665.  q[kk] = k*probs[kk] is int64*float32 -> the
 * product is formed in float64 and stored as float32; q[large] + q[small] - 1.0 is a float32
 * add followed by a float64 subtract stored as float32. */
static void orc_alias_setup(const float *probs, uint32_t k, uint32_t *aj, float *aq,
                            uint32_t *smaller, uint32_t *larger) {
    uint32_t sp = 0, lp = 0;
    for (uint32_t kk = 0; kk < k; kk++) {
        aj[kk] = 0;
        aq[kk] = (float)((double)k * (double)probs[kk]);
        if ((double)aq[kk] < 1.0) smaller[sp++] = kk;
        else larger[lp++] = kk;
    }
    while (sp > 0 && lp > 0) {
        uint32_t small = smaller[--sp];
        uint32_t large = larger[--lp];
        aj[small] = large;
        aq[large] = (float)((double)(aq[large] + aq[small]) - 1.0);
        if ((double)aq[large] < 1.0) smaller[sp++] = large;
        else larger[lp++] = large;
    }
}

/* alias_draw, src/pecanpy/pecanpy.py:668-677 */
static inline uint32_t orc_alias_draw(orc_mt_t *rng, const uint32_t *aj, const float *aq,
                                      uint32_t k) {
    uint32_t kk = (uint32_t)orc_mt_randint(rng, k);
    double u = orc_mt_random(rng);
    return (u < (double)aq[kk]) ? kk : aj[kk];
}

/* PreComp.preprocess_transition_probs, src/pecanpy/pecanpy.py:442-507.
 * alias_indptr is uint64[n+1] = cumsum(deg^2); alias_j/alias_q hold alias_indptr[n] entries. */
ORC_API int orc_precomp_tables(const uint32_t *indptr, const uint32_t *indices, const float *data,
                               uint32_t n_nodes, double p, double q, const float *thr,
                               const uint64_t *alias_indptr, uint32_t *alias_j, float *alias_q) {
    uint32_t md = orc_max_degree(indptr, n_nodes);
    float *w = (float *)malloc(sizeof(float) * (md + 1));
    float *t = (float *)malloc(sizeof(float) * (md + 1));
    uint8_t *flag = (uint8_t *)malloc(md + 1);
    uint32_t *sm = (uint32_t *)malloc(sizeof(uint32_t) * (md + 1));
    uint32_t *lg = (uint32_t *)malloc(sizeof(uint32_t) * (md + 1));
    for (uint32_t v = 0; v < n_nodes; v++) {
        uint32_t s0 = indptr[v], d = indptr[v + 1] - s0;
        for (uint32_t nb = 0; nb < d; nb++) {
            orc_sparse_probs_impl(indptr, indices, data, p, q, v, 1, indices[s0 + nb], thr, w, flag, t);
            uint64_t off = alias_indptr[v] + (uint64_t)d * nb;
            orc_alias_setup(w, d, alias_j + off, alias_q + off, sm, lg);
        }
    }
    free(w); free(t); free(flag); free(sm); free(lg);
    return 0;
}

/* PreComp walks (PreComp.move_forward, src/pecanpy/pecanpy.py:409-438). Sequential stream with a
 * variable number of words per step (App. B). */
ORC_API int orc_walks_precomp(const uint32_t *indptr, const uint32_t *indices, const float *data,
                              uint32_t n_nodes, double p, double q, const uint64_t *alias_indptr,
                              const uint32_t *alias_j, const float *alias_q,
                              const uint32_t *starts, uint64_t n_jobs, uint32_t L, uint32_t seed,
                              uint32_t *out, orc_stats_t *stats) {
    orc_stats_t st = {0, 0, 0};
    uint32_t nnz = indptr[n_nodes];
    uint64_t n_alias = alias_indptr[n_nodes];
    uint32_t md = orc_max_degree(indptr, n_nodes);
    float *w = (float *)malloc(sizeof(float) * (md + 1));
    float *t = (float *)malloc(sizeof(float) * (md + 1));
    uint8_t *flag = (uint8_t *)malloc(md + 1);
    orc_mt_t rng;
    orc_mt_seed(&rng, seed);
    const uint64_t W = (uint64_t)L + 2;
    for (uint64_t i = 0; i < n_jobs; i++) {
        uint32_t *row = out + i * W;
        memset(row, 0, sizeof(uint32_t) * W);
        row[0] = starts[i];
        row[L + 1] = L + 1;
        if (indptr[row[0]] == indptr[row[0] + 1]) { row[L + 1] = 1; continue; }
        for (uint32_t j = 1; j <= L; j++) {
            uint32_t cur = row[j - 1];
            uint32_t s0 = indptr[cur], d = indptr[cur + 1] - s0;
            if (d == 0) { row[L + 1] = j; break; }
            uint32_t choice;
            if (j == 1) {
                orc_sparse_probs_impl(indptr, indices, data, p, q, cur, 0, 0, NULL, w, flag, t);
                choice = orc_cdf_search_f32(w, d, orc_mt_random(&rng));
            } else {
                uint32_t prev = row[j - 2];
                uint32_t lo = 0, hi = d; /* np.searchsorted(row(cur), prev), left */
                while (hi > lo) {
                    uint32_t mid = (lo + hi) >> 1;
                    if (indices[s0 + mid] < prev) lo = mid + 1; else hi = mid;
                }
                uint64_t off = alias_indptr[cur] + (uint64_t)d * lo;
                if (off + d > n_alias) off = n_alias - d; /* would be a wild read: clamp */
                choice = orc_alias_draw(&rng, alias_j + off, alias_q + off, d);
            }
            row[j] = orc_read_choice(indptr, indices, nnz, cur, choice, d, &st);
            st.total_steps++;
        }
    }
    free(w); free(t); free(flag);
    if (stats) *stats = st;
    return 0;
}

/* PreCompFirstOrder tables (src/pecanpy/pecanpy.py:336-361; probs = w / sum(w), sparse_rw.py:37-49) */
ORC_API int orc_first_order_tables(const uint32_t *indptr, const float *data, uint32_t n_nodes,
                                   uint32_t *alias_j, float *alias_q) {
    uint32_t md = orc_max_degree(indptr, n_nodes);
    float *w = (float *)malloc(sizeof(float) * (md + 1));
    uint32_t *sm = (uint32_t *)malloc(sizeof(uint32_t) * (md + 1));
    uint32_t *lg = (uint32_t *)malloc(sizeof(uint32_t) * (md + 1));
    for (uint32_t v = 0; v < n_nodes; v++) {
        uint32_t s0 = indptr[v], d = indptr[v + 1] - s0;
        float tot = 0.0f;
        for (uint32_t k = 0; k < d; k++) tot += data[s0 + k];
        for (uint32_t k = 0; k < d; k++) w[k] = data[s0 + k] / tot;
        orc_alias_setup(w, d, alias_j + s0, alias_q + s0, sm, lg);
    }
    free(w); free(sm); free(lg);
    return 0;
}

/* mode 0: FirstOrderUnweighted (pecanpy.py:299-309) ; mode 1: PreCompFirstOrder (:319-334) */
ORC_API int orc_walks_first_order(const uint32_t *indptr, const uint32_t *indices,
                                  uint32_t n_nodes, int mode, const uint32_t *alias_j,
                                  const float *alias_q, const uint32_t *starts, uint64_t n_jobs,
                                  uint32_t L, uint32_t seed, uint32_t *out, orc_stats_t *stats) {
    orc_stats_t st = {0, 0, 0};
    (void)n_nodes;
    orc_mt_t rng;
    orc_mt_seed(&rng, seed);
    const uint64_t W = (uint64_t)L + 2;
    for (uint64_t i = 0; i < n_jobs; i++) {
        uint32_t *row = out + i * W;
        memset(row, 0, sizeof(uint32_t) * W);
        row[0] = starts[i];
        row[L + 1] = L + 1;
        if (indptr[row[0]] == indptr[row[0] + 1]) { row[L + 1] = 1; continue; }
        for (uint32_t j = 1; j <= L; j++) {
            uint32_t cur = row[j - 1];
            uint32_t s0 = indptr[cur], d = indptr[cur + 1] - s0;
            if (d == 0) { row[L + 1] = j; break; }
            uint32_t choice;
            if (mode == 0) choice = (uint32_t)orc_mt_randint(&rng, d);
            else choice = orc_alias_draw(&rng, alias_j + s0, alias_q + s0, d);
            row[j] = indices[s0 + choice];
            st.total_steps++;
        }
    }
    if (stats) *stats = st;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Stream helpers                                                                             */
/* ------------------------------------------------------------------------------------------ */

/* doubles #offset .. #offset+n-1 of RandomState(seed).random_sample (App. B stream addressing) */
ORC_API void orc_random_sample(uint32_t seed, uint64_t offset, uint64_t n, double *out) {
    orc_mt_t rng;
    orc_mt_seed(&rng, seed);
    for (uint64_t s = 0; s < offset; s++) (void)orc_mt_random(&rng);
    for (uint64_t s = 0; s < n; s++) out[s] = orc_mt_random(&rng);
}

/* raw tempered words #offset.. (for checking the product's jump-ahead) */
ORC_API void orc_random_words(uint32_t seed, uint64_t offset, uint64_t n, uint32_t *out) {
    orc_mt_t rng;
    orc_mt_seed(&rng, seed);
    for (uint64_t s = 0; s < offset; s++) (void)orc_mt_next32(&rng);
    for (uint64_t s = 0; s < n; s++) out[s] = orc_mt_next32(&rng);
}
