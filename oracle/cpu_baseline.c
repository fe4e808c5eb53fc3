/*
 * oracle/cpu_baseline.c -- TEST / BENCH INFRASTRUCTURE ONLY (the timed CPU baseline of bench.py).
 *
 * Multi-threaded port of the reference's Numba-parallel SparseOTF path, used as the
 * "cpu_baseline" leg (kind = "port"; PecanPy itself cannot run on the GPU box: Numba is not
 * installed and reference code never travels).  Same per-step algorithm as the reference:
 *   get_nbrs copy            src/pecanpy/rw/sparse_rw.py:133-139
 *   two-pointer isnotin      src/pecanpy/rw/sparse_rw.py:142-230
 *   /q, /p, sum, normalise   src/pecanpy/rw/sparse_rw.py:77-91
 *   cumsum + searchsorted    src/pecanpy/pecanpy.py:556-557
 *   prange over jobs, thread-local MT19937   src/pecanpy/pecanpy.py:177-178,189
 * faithful = 1 reproduces Numba's per-step heap temporaries (one malloc/free per array the njit
 * code materialises: weights copy, prev weights copy, bool mask, probs, cdf ...);
 * faithful = 0 is a tuned variant with per-thread scratch buffers.
 * Static chunking of the (already shuffled) job array over OpenMP threads, like prange.
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mt19937.h"

#define CB_API __attribute__((visibility("default")))

static inline uint32_t step_faithful(const uint32_t *indptr, const uint32_t *indices,
                                     const float *data, double p, double q, uint32_t cur,
                                     int has_prev, uint32_t prev, double r) {
    uint32_t s0 = indptr[cur], d = indptr[cur + 1] - s0;
    const uint32_t *nb = indices + s0;
    float *w = (float *)malloc(sizeof(float) * d); /* get_nbrs(...).copy() */
    memcpy(w, data + s0, sizeof(float) * d);
    if (has_prev) {
        uint32_t t0 = indptr[prev], dp = indptr[prev + 1] - t0;
        const uint32_t *pb = indices + t0;
        uint8_t *eq = (uint8_t *)malloc(d); /* nbrs_idx == prev_idx */
        for (uint32_t k = 0; k < d; k++) eq[k] = nb[k] == prev;
        float *pw = (float *)malloc(sizeof(float) * (dp ? dp : 1)); /* prev weights copy (unused) */
        memcpy(pw, data + t0, sizeof(float) * dp);
        uint8_t *flag = (uint8_t *)malloc(d); /* isnotin indicator */
        memset(flag, 1, d);
        uint32_t i2 = 0;
        for (uint32_t i1 = 0; i1 < d && i2 < dp; i1++) {
            uint32_t v1 = nb[i1], v2 = pb[i2];
            if (v1 < v2) continue;
            if (v1 == v2) { flag[i1] = 0; i2++; }
            else {
                uint32_t j = i2;
                for (; j < dp; j++) {
                    v2 = pb[j];
                    if (v2 == v1) { flag[i1] = 0; i2 = j + 1; break; }
                    if (v2 > v1) { i2 = j; break; }
                }
            }
        }
        for (uint32_t k = 0; k < d; k++) if (eq[k]) flag[k] = 0;
        for (uint32_t k = 0; k < d; k++) if (flag[k]) w[k] = (float)((double)w[k] / q);
        for (uint32_t k = 0; k < d; k++) if (eq[k]) w[k] = (float)((double)w[k] / p);
        free(flag); free(pw); free(eq);
    }
    float tot = 0.0f;
    for (uint32_t k = 0; k < d; k++) tot += w[k];
    float *pr = (float *)malloc(sizeof(float) * d); /* w / w.sum() */
    for (uint32_t k = 0; k < d; k++) pr[k] = w[k] / tot;
    float *cdf = (float *)malloc(sizeof(float) * d); /* np.cumsum */
    float c = 0.0f;
    for (uint32_t k = 0; k < d; k++) { c += pr[k]; cdf[k] = c; }
    uint32_t lo = 0, hi = d; /* np.searchsorted, left */
    while (hi > lo) {
        uint32_t mid = (lo + hi) >> 1;
        if ((double)cdf[mid] < r) lo = mid + 1; else hi = mid;
    }
    free(cdf); free(pr); free(w);
    return lo;
}

static inline uint32_t step_tuned(const uint32_t *indptr, const uint32_t *indices, const float *data,
                                  double p, double q, uint32_t cur, int has_prev, uint32_t prev,
                                  double r, float *w) {
    uint32_t s0 = indptr[cur], d = indptr[cur + 1] - s0;
    const uint32_t *nb = indices + s0;
    if (has_prev) {
        uint32_t t0 = indptr[prev], dp = indptr[prev + 1] - t0;
        const uint32_t *pb = indices + t0;
        uint32_t i2 = 0;
        for (uint32_t k = 0; k < d; k++) {
            uint32_t v = nb[k];
            while (i2 < dp && pb[i2] < v) i2++;
            float x = data[s0 + k];
            if (v == prev) x = (float)((double)x / p);
            else if (!(i2 < dp && pb[i2] == v)) x = (float)((double)x / q);
            w[k] = x;
        }
    } else {
        memcpy(w, data + s0, sizeof(float) * d);
    }
    float tot = 0.0f;
    for (uint32_t k = 0; k < d; k++) tot += w[k];
    float c = 0.0f;
    for (uint32_t k = 0; k < d; k++) {
        c += w[k] / tot;
        if ((double)c >= r) return k;
    }
    return d;
}

/* Returns 0; *steps_out = sampled transitions; walks written to out (may be NULL to skip). */
CB_API int cpub_walks_sparse(const uint32_t *indptr, const uint32_t *indices, const float *data,
                             uint32_t n_nodes, double p, double q, const uint32_t *starts,
                             uint64_t n_jobs, uint32_t L, uint32_t seed, int n_threads,
                             int faithful, uint32_t *out, uint64_t *steps_out) {
    uint32_t nnz = indptr[n_nodes];
    uint32_t md = 0;
    for (uint32_t i = 0; i < n_nodes; i++) {
        uint32_t d = indptr[i + 1] - indptr[i];
        if (d > md) md = d;
    }
    omp_set_num_threads(n_threads > 0 ? n_threads : omp_get_num_procs());
    uint64_t steps = 0;
    const uint64_t W = (uint64_t)L + 2;
#pragma omp parallel reduction(+ : steps)
    {
        orc_mt_t rng;
        orc_mt_seed(&rng, seed + (uint32_t)omp_get_thread_num());
        float *w = (float *)malloc(sizeof(float) * (md + 1));
        uint32_t *rowbuf = (uint32_t *)malloc(sizeof(uint32_t) * W);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n_jobs; i++) {
            uint32_t *row = out ? out + (uint64_t)i * W : rowbuf;
            memset(row, 0, sizeof(uint32_t) * W);
            row[0] = starts[i];
            row[L + 1] = L + 1;
            if (indptr[row[0]] == indptr[row[0] + 1]) { row[L + 1] = 1; continue; }
            for (uint32_t j = 1; j <= L; j++) {
                uint32_t cur = row[j - 1];
                uint32_t d = indptr[cur + 1] - indptr[cur];
                if (d == 0) { row[L + 1] = j; break; }
                double r = orc_mt_random(&rng);
                uint32_t prev = j >= 2 ? row[j - 2] : 0;
                uint32_t choice = faithful
                    ? step_faithful(indptr, indices, data, p, q, cur, j >= 2, prev, r)
                    : step_tuned(indptr, indices, data, p, q, cur, j >= 2, prev, r, w);
                uint64_t pos = (uint64_t)indptr[cur] + choice;
                if (pos >= nnz) pos = nnz - 1;
                row[j] = indices[pos];
                steps++;
            }
        }
        free(w);
        free(rowbuf);
    }
    if (steps_out) *steps_out = steps;
    return 0;
}

CB_API int cpub_max_threads(void) { return omp_get_num_procs(); }
