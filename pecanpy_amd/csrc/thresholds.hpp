// thresholds.hpp -- node2vec+ noisy-edge thresholds on the host (no GPU involved).
//
// The reference computes thr[i] = max(mean(row_i) + gamma * std(row_i), 0) with plain NumPy, one Python-level
// call per row (src/pecanpy/rw/sparse_rw.py:22-35, rw/dense_rw.py:11-19).  The thresholds feed float32
// comparisons inside the walk (sparse_rw.py:262-264), so they have to come out bit for bit as NumPy
// produces them.  NumPy's float reductions are well defined: np.add.reduce on a contiguous array adds, per
// buffer of 8192 elements, `pairwise_sum(chunk)` (the 8-accumulator / 128-element-block scheme of
// numpy/core/src/umath/loops_utils.h.src) to the running result, mean = sum / n, var = sum((a - mean)^2) / n, std = sqrt(var),
// every step in the array's own precision.  This file restates exactly that.
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

namespace pw {

template <typename T> inline T numpy_pairwise_sum(const T *a, uint64_t n) {
    if (n < 8) {
        T res = (T)0;
        for (uint64_t i = 0; i < n; i++) res = res + a[i];
        return res;
    }
    if (n <= 128) {
        T r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        uint64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
        T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res = res + a[i];
        return res;
    }
    uint64_t n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_pairwise_sum(a, n2) + numpy_pairwise_sum(a + n2, n - n2);
}

// np.add.reduce over a contiguous 1-d array: the ufunc machinery hands the inner loop at most
// np.getbufsize() = 8192 elements at a time, out = out + pairwise_sum(chunk), starting from the identity 0
template <typename T> inline T numpy_add_reduce(const T *a, uint64_t n) {
    const uint64_t BUF = 8192;
    T acc = (T)0;
    for (uint64_t i = 0; i < n; i += BUF) acc = acc + numpy_pairwise_sum(a + i, n - i < BUF ? n - i : BUF);
    return acc;
}

// mean and std (ddof = 0) of one row as ndarray.mean() / ndarray.std() return them; n == 0 gives NaN
template <typename T> inline void numpy_mean_std(const T *a, uint64_t n, std::vector<T> &scratch, T &mean, T &stdev) {
    const T cnt = (T)n;
    mean = numpy_add_reduce(a, n) / cnt;
    scratch.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        const T x = a[i] - mean;
        scratch[i] = x * x;
    }
    const T var = numpy_add_reduce(scratch.data(), n) / cnt;
    stdev = (T)sqrt((double)var);   // correctly rounded sqrt of a T value, rounded to T: same as sqrtf / sqrt
}

// CSR rows of float32 weights.  How `row.mean() + gamma * row.std()` (two float32 scalars and a Python float) is
// evaluated depends on the NumPy the reference runs under:
//   numpy1 = false: NumPy >= 2 (NEP 50): the Python float is "weak", everything stays float32 (two roundings);
//   numpy1 = true : NumPy 1.x (the reference pins 1.23.2): float32 scalar * Python float -> float64, the sum is
//                   float64 and is rounded once when stored into the float32 threshold array.
// Both agree whenever gamma * std is exact in float32 (gamma = 0, 0.5, ... as in the fixtures).
inline void noise_thresholds_csr(const uint32_t *indptr, const float *data, uint32_t n_nodes, double gamma, float *thr,
                                 bool numpy1 = false) {
    std::vector<float> scratch;
    const float g = (float)gamma;
    for (uint32_t i = 0; i < n_nodes; i++) {
        float m, s;
        numpy_mean_std<float>(data + indptr[i], (uint64_t)indptr[i + 1] - indptr[i], scratch, m, s);
        const float t = numpy1 ? (float)((double)m + gamma * (double)s) : m + g * s;
        thr[i] = (t != t) ? t : (t > 0.0f ? t : 0.0f);   // np.maximum(t, 0): NaN propagates
    }
}

// dense float64 matrix, row i restricted to its non-zero entries; float64 arithmetic, stored as float32
inline void noise_thresholds_dense(const double *mat, uint32_t n, double gamma, float *thr) {
    std::vector<double> row, scratch;
    for (uint32_t i = 0; i < n; i++) {
        row.clear();
        const double *r = mat + (uint64_t)i * n;
        for (uint32_t j = 0; j < n; j++)
            if (r[j] != 0.0) row.push_back(r[j]);
        double m, s;
        numpy_mean_std<double>(row.data(), row.size(), scratch, m, s);
        const float t = (float)(m + gamma * s);
        thr[i] = (t != t) ? t : (t > 0.0f ? t : 0.0f);
    }
}

}  // namespace pw
