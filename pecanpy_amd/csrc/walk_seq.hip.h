// walk_seq.hip.h -- alias-table modes (PreComp, PreCompFirstOrder) and FirstOrderUnweighted (gfx950).
//
// These reference modes draw a *variable* number of MT19937 words per step (masked rejection in
// np.random.randint, numba:cpython/randomimpl.py:320-387; call sites src/pecanpy/pecanpy.py:307,
// 673-674), so the stream position of walk i depends on every earlier draw: a seeded run that must
// reproduce the reference bit for bit is inherently sequential.
// The exact path below therefore walks all jobs in order on ONE lane with the generator state in
// LDS.  It exists for API / CLI compatibility on the small graphs these modes are meant for
// (reference README: PreComp for < 10k nodes); the throughput path of this engine is SparseOTF.
//
// Alias tables are built on the device, one thread per (vertex, previous-neighbour) pair, with
// scalar code that follows the reference statement by statement (probabilities: sparse_rw.py:51-130,
// alias_setup: pecanpy.py:617-665) -- sequential float32 sums come out exact for free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aux_kernels.hip.h"
#include "walk_sparse.hip.h"

namespace pw {

// ---- scalar transition probabilities of (cur | prev) into w[0..d) ------------------------------------
__device__ inline void seq_probs(const CsrDev &g, double p, double q, bool extend, uint32_t cur, bool has_prev,
                                 uint32_t prev, float *w) {
    const float *__restrict__ data = (const float *)g.data;
    const uint32_t s0 = g.indptr[cur], d = g.indptr[cur + 1] - s0;
    const uint32_t *nb = g.indices + s0;
    for (uint32_t k = 0; k < d; k++) w[k] = data ? data[s0 + k] : 1.0f;
    if (has_prev) {
        const uint32_t t0 = g.indptr[prev], dp = g.indptr[prev + 1] - t0;
        const uint32_t *pb = g.indices + t0;
        const double inv_q = 1.0 / q;
        const double noisy = inv_q < 1.0 ? inv_q : 1.0;
        const float thr_cur = extend ? g.thr[cur] : 0.0f;
        uint32_t i2 = 0;
        for (uint32_t k = 0; k < d; k++) {
            const uint32_t x = nb[k];
            while (i2 < dp && pb[i2] < x) i2++;
            const bool common = i2 < dp && pb[i2] == x;
            if (x == prev) { w[k] = (float)((double)w[k] / p); continue; }
            if (!extend) {
                if (!common) w[k] = (float)((double)w[k] / q);
            } else {
                float t = 0.0f;
                bool in_edge = false;
                if (common) {
                    const float u = data ? data[t0 + i2] : 1.0f;
                    if (u >= g.thr[x]) in_edge = true;
                    else t = u / g.thr[x];
                }
                if (!in_edge) {
                    double alpha = inv_q + (1.0 - inv_q) * (double)t;
                    if (w[k] < thr_cur) alpha = noisy;
                    w[k] = (float)((double)w[k] * alpha);
                }
            }
        }
    }
    float tot = 0.0f;
    for (uint32_t k = 0; k < d; k++) tot += w[k];
    for (uint32_t k = 0; k < d; k++) w[k] = w[k] / tot;
}

// alias_setup (pecanpy.py:617-665) in place: probs arrive in aq[0..k), leave as the q table.
__device__ inline void seq_alias_setup(uint32_t k, uint32_t *aj, float *aq, uint32_t *smaller, uint32_t *larger) {
    uint32_t sp = 0, lp = 0;
    for (uint32_t kk = 0; kk < k; kk++) {
        aj[kk] = 0;
        aq[kk] = (float)((double)k * (double)aq[kk]);
        if ((double)aq[kk] < 1.0) smaller[sp++] = kk;
        else larger[lp++] = kk;
    }
    while (sp > 0 && lp > 0) {
        const uint32_t small = smaller[--sp];
        const uint32_t large = larger[--lp];
        aj[small] = large;
        aq[large] = (float)((double)(aq[large] + aq[small]) - 1.0);
        if ((double)aq[large] < 1.0) smaller[sp++] = large;
        else larger[lp++] = large;
    }
}

// second order: one thread per (v, nb) pair = per CSR entry e; first order: one thread per vertex
__global__ void __launch_bounds__(256)
alias_tables_kernel(CsrDev g, double p, double q, int extend, int first_order, const uint32_t *__restrict__ edge_row,
                    const uint64_t *__restrict__ alias_indptr, uint32_t *alias_j, float *alias_q,
                    uint32_t *scratch_s, uint32_t *scratch_l) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (first_order) {
        if (tid >= g.n_nodes) return;
        const uint32_t v = (uint32_t)tid;
        const uint32_t s0 = g.indptr[v], d = g.indptr[v + 1] - s0;
        if (d == 0) return;
        seq_probs(g, 1.0, 1.0, false, v, false, 0, alias_q + s0);
        seq_alias_setup(d, alias_j + s0, alias_q + s0, scratch_s + s0, scratch_l + s0);
        return;
    }
    if (tid >= g.nnz) return;
    const uint32_t e = (uint32_t)tid;
    const uint32_t v = edge_row[e];
    const uint32_t s0 = g.indptr[v], d = g.indptr[v + 1] - s0;
    const uint32_t nb_idx = e - s0;
    const uint64_t off = alias_indptr[v] + (uint64_t)d * nb_idx;
    seq_probs(g, p, q, extend != 0, v, true, g.indices[e], alias_q + off);
    seq_alias_setup(d, alias_j + off, alias_q + off, scratch_s + off, scratch_l + off);
}

__global__ void edge_rows_kernel(const uint32_t *__restrict__ indptr, uint32_t n_nodes, uint32_t *edge_row) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes) return;
    for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) edge_row[e] = v;
}

// ---- sequential generator in LDS ------------------------------------------------------------------------
struct SeqMt {
    uint32_t *mt;  // 624 words (LDS)
    int idx;
    __device__ inline uint32_t next32() {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) mt[i] = mt_mix_dev(mt[i], mt[(i + 1) % 624], mt[(i + 397) % 624]);
            idx = 0;
        }
        return mt_temper_dev(mt[idx++]);
    }
    __device__ inline double random() {
        const uint32_t a = next32() >> 5, b = next32() >> 6;
        return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
    }
    // np.random.randint(n): n == 1 -> no words; else low bit_length(n-1) bits, reject >= n
    __device__ inline uint32_t randint(uint32_t n) {
        if (n == 1) return 0;
        const int nbits = 32 - __clz(n - 1);
        const uint32_t mask = nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u);
        for (;;) {
            const uint32_t r = next32() & mask;
            if (r < n) return r;
        }
    }
};

struct SeqArgs {
    CsrDev g;
    double p, q;
    int mode;  // pw_mode: 2 PreComp, 3 FirstOrderUnweighted, 4 PreCompFirstOrder
    uint32_t L;
    uint64_t n_jobs;
    const uint32_t *__restrict__ starts;
    const uint32_t *__restrict__ mt_seed_state;  // 624 words
    const uint64_t *__restrict__ alias_indptr;
    const uint32_t *__restrict__ alias_j;
    const float *__restrict__ alias_q;
    uint64_t n_alias;
    float *probs_scratch;  // max_degree floats (PreComp first steps)
    uint32_t *out;
    unsigned long long *stats;
};

__device__ inline uint32_t seq_alias_draw(SeqMt &rng, const uint32_t *aj, const float *aq, uint32_t k) {
    const uint32_t kk = rng.randint(k);
    const double u = rng.random();
    return (u < (double)aq[kk]) ? kk : aj[kk];
}

__global__ void __launch_bounds__(64)
walk_seq_kernel(SeqArgs a) {
    __shared__ uint32_t s_mt[624];
    if (threadIdx.x != 0) return;
    for (int i = 0; i < 624; i++) s_mt[i] = a.mt_seed_state[i];
    SeqMt rng{s_mt, 624};
    const CsrDev &g = a.g;
    const uint32_t L = a.L;
    const uint64_t W = (uint64_t)L + 2;
    unsigned long long steps = 0, over = 0, clamp = 0, dead = 0;
    for (uint64_t i = 0; i < a.n_jobs; i++) {
        uint32_t *row = a.out + i * W;
        for (uint32_t z = 0; z < W; z++) row[z] = 0;
        row[0] = a.starts[i];
        row[L + 1] = L + 1;
        uint32_t cur = row[0], prev = 0;
        for (uint32_t j = 1; j <= L; j++) {
            const uint32_t s0 = g.indptr[cur], d = g.indptr[cur + 1] - s0;
            if (d == 0) { row[L + 1] = j; if (j > 1) dead++; break; }
            uint32_t choice;
            if (a.mode == 3) {
                choice = rng.randint(d);
            } else if (a.mode == 4) {
                choice = seq_alias_draw(rng, a.alias_j + s0, a.alias_q + s0, d);
            } else if (j == 1) {
                seq_probs(g, a.p, a.q, false, cur, false, 0, a.probs_scratch);
                const double r = rng.random();
                float c = 0.0f;
                choice = d;
                for (uint32_t k = 0; k < d; k++) {
                    c += a.probs_scratch[k];
                    if ((double)c >= r) { choice = k; break; }
                }
            } else {
                uint32_t lo = 0, hi = d;  // np.searchsorted(row(cur), prev), left
                while (hi > lo) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (g.indices[s0 + mid] < prev) lo = mid + 1; else hi = mid;
                }
                uint64_t off = a.alias_indptr[cur] + (uint64_t)d * lo;
                if (off + d > a.n_alias) off = a.n_alias - d;  // reference would read past the table
                choice = seq_alias_draw(rng, a.alias_j + off, a.alias_q + off, d);
            }
            uint64_t pos = (uint64_t)s0 + choice;
            if (choice >= d) {
                over++;
                if (pos >= g.nnz) { pos = g.nnz - 1; clamp++; }
            }
            const uint32_t nxt = g.indices[pos];
            row[j] = nxt;
            prev = cur;
            cur = nxt;
            steps++;
        }
    }
    a.stats[0] = steps;
    a.stats[1] = over;
    a.stats[2] = clamp;
    a.stats[3] = dead;
}

// ---- unseeded runs of the alias / first-order modes: one lane per walk, counter-based draws -------------
// With random_state = None the reference is not reproducible either (NumPy/Numba seed themselves from
// the OS, pecanpy.py:139-140, 177), so there is no stream to match: every walk draws from its own
// counter-based generator (splitmix64 of (seed, job, draw index)) and all walks run in parallel.
// The sampled distribution is the same (same alias tables, same alias_draw / randint semantics).
struct CounterRng {
    uint64_t key, ctr;
    __device__ inline uint64_t next64() {
        uint64_t z = (key + (++ctr) * 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    __device__ inline uint32_t next32() { return (uint32_t)(next64() >> 32); }
    __device__ inline double random() { return (double)(next64() >> 11) * (1.0 / 9007199254740992.0); }
    __device__ inline uint32_t randint(uint32_t n) {  // unbiased: masked rejection like the reference
        if (n == 1) return 0;
        const int nbits = 32 - __clz(n - 1);
        const uint32_t mask = nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u);
        for (;;) {
            const uint32_t r = next32() & mask;
            if (r < n) return r;
        }
    }
};

__global__ void __launch_bounds__(256)
walk_alias_parallel_kernel(SeqArgs a, uint64_t seed) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_jobs) return;
    const CsrDev &g = a.g;
    const float *__restrict__ data = (const float *)g.data;
    const uint32_t L = a.L;
    const uint64_t W = (uint64_t)L + 2;
    CounterRng rng{seed * 0xD6E8FEB86659FD93ull + i * 0xA24BAED4963EE407ull + 0x9FB21C651E98DF25ull, 0};
    uint32_t *row = a.out + i * W;
    for (uint32_t z = 0; z < W; z++) row[z] = 0;
    row[0] = a.starts[i];
    row[L + 1] = L + 1;
    uint32_t cur = row[0], prev = 0;
    unsigned long long steps = 0, dead = 0;
    for (uint32_t j = 1; j <= L; j++) {
        const uint32_t s0 = g.indptr[cur], d = g.indptr[cur + 1] - s0;
        if (d == 0) { row[L + 1] = j; if (j > 1) dead++; break; }
        uint32_t choice;
        if (a.mode == 3) {
            choice = rng.randint(d);
        } else if (a.mode == 4) {
            const uint32_t kk = rng.randint(d);
            choice = (rng.random() < (double)a.alias_q[s0 + kk]) ? kk : a.alias_j[s0 + kk];
        } else if (j == 1) {
            // first-order step: sequential float32 sum / cumsum over the row, as the reference
            float tot = 0.0f;
            for (uint32_t k = 0; k < d; k++) tot += data ? data[s0 + k] : 1.0f;
            const double r = rng.random();
            float c = 0.0f;
            choice = d - 1;
            for (uint32_t k = 0; k < d; k++) {
                c += (data ? data[s0 + k] : 1.0f) / tot;
                if ((double)c >= r) { choice = k; break; }
            }
        } else {
            uint32_t lo = 0, hi = d;
            while (hi > lo) {
                const uint32_t mid = (lo + hi) >> 1;
                if (g.indices[s0 + mid] < prev) lo = mid + 1; else hi = mid;
            }
            uint64_t off = a.alias_indptr[cur] + (uint64_t)d * lo;
            if (off + d > a.n_alias) off = a.n_alias - d;
            const uint32_t kk = rng.randint(d);
            choice = (rng.random() < (double)a.alias_q[off + kk]) ? kk : a.alias_j[off + kk];
        }
        const uint32_t nxt = g.indices[s0 + (choice < d ? choice : d - 1)];
        row[j] = nxt;
        prev = cur;
        cur = nxt;
        steps++;
    }
    if (steps) atomicAdd(&a.stats[0], steps);
    if (dead) atomicAdd(&a.stats[3], dead);
}

}  // namespace pw
