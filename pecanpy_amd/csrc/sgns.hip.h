// sgns.hip.h -- skip-gram with negative sampling over a walk matrix (gfx950): the stage that follows the walks in
// the reference's pipeline (Base.embed / cli.learn_embeddings: gensim Word2Vec(walks, sg=1, negative=5, window, epochs),
// src/pecanpy/pecanpy.py:276-290, cli.py:307-325).  SURVEY.md section 8(f) rank 4.  The model and update rule are
// word2vec.c's (gensim's sg / negative path):
//   a walk is thinned by word2vec's frequent-word subsampling, windows are taken over what is left; for every centre
//   position a window shrunk by a random amount, and for every context word c in it: input vector syn0[c], targets =
//   the centre (label 1) and `negative` words drawn from the unigram^0.75 table (label 0),
//   g = (label - sigmoid(v.u)) * lr, u += g v, v += sum g u; learning rate decaying linearly over the run.
// Every random choice is a hash of (seed, epoch, walk, position[, context, draw]) -- no generator state -- so the set
// of updates is a function of the seed alone; what the hardware adds is their ORDER: one wavefront per (walk,
// position), hogwild (unsynchronised) like gensim's worker threads.  With ONE wavefront (pw_sgns_train: workers = 1)
// the updates run in sentence order and the result is compared with the sequential CPU restatement
// (oracle/sgns_ref.c, tests/test_gpu_sgns.py) within float tolerance.
// The `dim` components of a vector are spread over the 64 lanes, dot products are wave reductions.
#pragma once
#include "wave.h"

namespace pw {

struct SgnsArgs {
    const uint32_t *__restrict__ walks;   // [n_walks, L + 2], last cell = number of nodes in the walk
    uint64_t n_walks;
    uint32_t L;
    uint32_t dim, window, negative;
    float *syn0, *syn1;                   // [n_nodes, dim]
    const uint32_t *__restrict__ table;   // negative-sampling table (word2vec's unigram^0.75 table)
    uint32_t table_size;
    const float *__restrict__ keep;       // per word: probability of keeping an occurrence (subsampling), or nullptr
    float alpha, min_alpha;
    uint64_t item_base, item_total;       // position of this launch in the whole run (learning-rate schedule)
    uint64_t seed;
};

__device__ __forceinline__ uint64_t sgns_mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

constexpr int SGNS_MAX_PER_LANE = 8;   // dim <= 512

__global__ void __launch_bounds__(256)
sgns_kernel(SgnsArgs a) {
    const int lane = lane_id();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) / WAVE;
    const uint32_t W = a.L + 2, per = (a.dim + WAVE - 1) / WAVE;
    const uint64_t n_items = a.n_walks * (uint64_t)(a.L + 1);
    for (uint64_t item = wave; item < n_items; item += n_waves) {
        const uint64_t wk = item / (a.L + 1);
        const uint32_t pos = (uint32_t)(item - wk * (a.L + 1));
        const uint32_t *row = a.walks + wk * W;
        const uint32_t len = row[a.L + 1];
        if (pos >= len) continue;
        // occurrence (walk, p) of this epoch survives the subsampling?  (every wavefront that meets it agrees)
        const uint64_t occ0 = a.item_base + wk * (uint64_t)(a.L + 1);
#define occ(p) sgns_mix(a.seed ^ (occ0 + (p)) * 0x9E3779B97F4A7C15ull)
#define kept(p) (!a.keep || (float)(occ(p) >> 40) * (1.0f / 16777216.0f) < a.keep[row[p]])
        if (!kept(pos)) continue;
        const uint32_t centre = row[pos];
        uint64_t rs = sgns_mix(occ(pos));
        const uint32_t eff = a.window - (uint32_t)(rs % a.window);                              // shrunk window, 1 .. window
        const float lr = fmaxf(a.min_alpha, a.alpha - (a.alpha - a.min_alpha) * (float)((double)(a.item_base + item) / (double)a.item_total));
        // the window over the thinned walk: up to eff surviving positions on either side
        uint32_t lo = pos, hi = pos, got = 0;
        for (uint32_t c = pos; c-- > 0 && got < eff;) if (kept(c)) { lo = c; got++; }
        got = 0;
        for (uint32_t c = pos + 1; c < len && got < eff; c++) if (kept(c)) { hi = c; got++; }
        for (uint32_t c = lo; c <= hi; c++) {
            if (c == pos || !kept(c)) continue;
            const uint32_t ctx = row[c];
            rs = sgns_mix(rs + c);
            float *v = a.syn0 + (uint64_t)ctx * a.dim;
            float vin[SGNS_MAX_PER_LANE], acc[SGNS_MAX_PER_LANE];
#pragma unroll
            for (int t = 0; t < SGNS_MAX_PER_LANE; t++) {
                const uint32_t k = (uint32_t)t * WAVE + lane;
                vin[t] = (t < (int)per && k < a.dim) ? v[k] : 0.0f;
                acc[t] = 0.0f;
            }
            for (uint32_t ng = 0; ng <= a.negative; ng++) {
                uint32_t target = centre;
                if (ng) {
                    rs = sgns_mix(rs + ng);
                    target = a.table[(uint32_t)(rs >> 16) % a.table_size];
                    if (target == centre) continue;
                }
                float *u = a.syn1 + (uint64_t)target * a.dim;
                float uu[SGNS_MAX_PER_LANE], dot = 0.0f;
#pragma unroll
                for (int t = 0; t < SGNS_MAX_PER_LANE; t++) {
                    const uint32_t k = (uint32_t)t * WAVE + lane;
                    uu[t] = (t < (int)per && k < a.dim) ? u[k] : 0.0f;
                    dot += vin[t] * uu[t];
                }
                dot = wave_sum(dot);
                const float sig = dot > 6.0f ? 1.0f : (dot < -6.0f ? 0.0f : 1.0f / (1.0f + __expf(-dot)));
                const float g = ((ng == 0 ? 1.0f : 0.0f) - sig) * lr;
#pragma unroll
                for (int t = 0; t < SGNS_MAX_PER_LANE; t++) {
                    const uint32_t k = (uint32_t)t * WAVE + lane;
                    if (t < (int)per && k < a.dim) {
                        acc[t] += g * uu[t];
                        u[k] = uu[t] + g * vin[t];
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < SGNS_MAX_PER_LANE; t++) {
                const uint32_t k = (uint32_t)t * WAVE + lane;
                if (t < (int)per && k < a.dim) v[k] = vin[t] + acc[t];
            }
        }
#undef occ
#undef kept
    }
}

// word counts of a walk matrix (vocabulary statistics for the sampling table and the subsampling probabilities);
// bad[0] = first walk whose length cell exceeds L + 1 or that names a node >= n_nodes (atomicMin, ~0: none)
__global__ void __launch_bounds__(256)
sgns_count_kernel(const uint32_t *__restrict__ walks, uint64_t n_walks, uint32_t L, uint32_t n_nodes, unsigned long long *counts,
                  unsigned long long *bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t wk = i / (L + 1);
    if (wk >= n_walks) return;
    const uint32_t pos = (uint32_t)(i - wk * (L + 1));
    const uint32_t *row = walks + wk * ((uint64_t)L + 2);
    const uint32_t len = row[L + 1];
    if (len > L + 1) { atomicMin(bad, (unsigned long long)wk); return; }
    if (pos < len) {
        if (row[pos] >= n_nodes) { atomicMin(bad, (unsigned long long)wk); return; }
        atomicAdd(&counts[row[pos]], 1ull);
    }
}

}  // namespace pw
