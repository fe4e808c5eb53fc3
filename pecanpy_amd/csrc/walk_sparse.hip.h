// walk_sparse.hip.h -- SparseOTF walk kernel for gfx950 (device code).
//
// One 64-lane wavefront walks one job (start vertex) at a time; waves pull jobs from a global
// counter (persistent grid).  Per step (reference SparseOTF.move_forward, src/pecanpy/pecanpy.py:
// 543-559 + SparseRWGraph.get_normalized_probs, src/pecanpy/rw/sparse_rw.py:51-91):
//
//   1. membership  : which neighbours of `cur` are also neighbours of `prev`
//                    (reference: two-pointer isnotin, sparse_rw.py:142-230).  Done here by
//                    streaming the *shorter* of the two sorted rows and testing each entry
//                    against the longer row's Bloom filter; survivors are looked up in that
//                    row's adjacency hash index.  The result is one bit per neighbour of `cur`
//                    in an LDS bitmask owned by the wave.
//   2. tot         : sequential float32 sum of the biased weights      (sparse_rw.py:89)
//   3. cdf search  : first k with cumsum(w/tot)[k] >= r, sequential float32 (pecanpy.py:556-557)
//   4. next        : indices[indptr[cur] + k], k == degree mirrored     (pecanpy.py:559, App. D)
//
// 2 and 3 use the binade scan of seqscan.h so that the wave-parallel evaluation is bit-identical
// to the reference's left-to-right float32 loops.  Unweighted graphs take a closed-form variant
// of the same chain (run lengths of equal values, see "unit-weight fast path").
//
// Two implementations of a step:
//   eager (sample_step_unit / sample_step_weighted): 1-4 as listed, the full mask of the row first.
//         Weighted graphs, node2vec+, non-dyadic p/q, the first step of a walk, graphs with self loops.
//   lazy  (sample_step_unit_lazy, the headline case: unit weights, 1/p and 1/q powers of two):
//         tot from the per-edge common-neighbour count, membership 64 keys at a time and only as far
//         as the search needs it, the search decided in exact integer arithmetic whenever no partial
//         sum lies within the float32 drift bound of the target (else by the same binade chain).
//
// Everything a wave decides on is wave-uniform and lives in SGPRs; read-only arrays are read with
// scalar loads and kernel arguments are re-read at the point of use (wave.h) because the 80 SGPRs
// a wave gets at 8 waves/SIMD are the scarcest resource of this kernel.
#pragma once
#include "seqscan.h"
#include "wave.h"

namespace pw {

constexpr uint32_t NOT_FOUND = 0xffffffffu;

// Edge line of CSR entry e = (u -> v): one 64-byte aligned record -- everything a step needs about the edge it
// arrives by and, for the short lists that most steps meet, the list itself: ONE sector per step for both.
struct ELine {
    uint32_t nxt;       // v
    uint32_t n_in;      // |N(u) & N(v)|
    uint32_t rev_pos;   // position of u in row v, NOT_FOUND when (v -> u) is not an edge
    uint32_t deg;       // degree(v)       (these four words: the record walk_kernel's lazy step reads)
    uint32_t s0;        // indptr[v]
    uint32_t coff;      // list in the overflow array: offset in 16-byte units (n_in > EL_INLINE or degree(v) > 65536)
    uint16_t inl[20];   // the list itself when n_in <= EL_INLINE and degree(v) <= 65536: uint16 positions in row v;
                        // otherwise the list's PIVOTS (seqscan.h): every (n_in / 21)-th entry, 20 of them (10 uint32 for wider rows)
};
static_assert(sizeof(ELine) == 64, "edge line is one 64-byte sector");
constexpr uint32_t EL_INLINE = 20;
// PARTIAL INDEX: a list the byte budget of the index left out (pw_csr_create: PECANPY_AMD_INDEX_BUDGET / half of the free
// device memory; the LONGEST lists go first) has this offset.  Its line still holds the record -- count, reverse
// position, degree -- so only the step that ARRIVES by such an entry needs the membership of the two rows established
// the slow way (lanes_eager_kernel: one wavefront, walk_kernel's eager step), the walk itself stays in the lane kernel.
constexpr uint32_t EL_NO_LIST = 0xffffffffu;
__device__ __forceinline__ bool edge_list_stored(uint32_t d, uint32_t n_in, uint32_t coff) {
    return (d <= 65536u && n_in <= EL_INLINE) || coff != EL_NO_LIST;
}

// the list of common-neighbour positions of the edge a walk arrived by (n_in == 0: never dereferenced)
__device__ __forceinline__ ListView edge_list(const ELine *lines, const uint8_t *clist, uint32_t e, uint32_t d, uint32_t n_in,
                                              uint32_t coff) {
    const bool narrow = d <= 65536u;
    const bool inl = narrow && n_in <= EL_INLINE;
    const uint8_t *p = inl ? (const uint8_t *)(lines + e) + 24 : clist + (uint64_t)coff * 16u;
    ListView v{p, narrow ? 0u : 1u};
    if (!inl && list_has_pivots(v.wide, n_in)) {   // the unused inline area holds the list's pivots (eline_pivots_kernel)
        v.piv = (const uint8_t *)(lines + e) + 24;
        v.npiv = list_pivot_count(v.wide);
        v.step = list_pivot_step(v.wide, n_in);
    }
    return v;
}

struct CsrDev {
    const uint32_t *__restrict__ indptr;
    const uint32_t *__restrict__ indices;
    const void *__restrict__ data;   // float32 (SparseOTF) / float64 (DenseOTF); nullptr: all 1.0
    const float *__restrict__ thr;   // node2vec+ thresholds or nullptr
    // DenseOTF only: bit-packed adjacency, row u = words [u * words_per_row, ...), bit x of the
    // row set <=> nonzero[u, x].  Membership of x in N(prev) is one bit test instead of a search.
    const uint64_t *__restrict__ adjbits;
    uint32_t words_per_row;
    // per-row membership filter (blocked Bloom, 2 bits per neighbour inside one 64-bit word, 4-8
    // filter bits per neighbour): row u owns words [foff[u], foff[u+1]) of fbits (a power of two).
    // "x is a neighbour of u" is answered negatively with ONE cache-line access for ~90 % of the
    // non-neighbours; only the survivors pay the log2(d) probes of the exact search.
    // The word a neighbour id maps to is ORDER PRESERVING: word = floor(W * F(v)) with F the degree CDF of
    // the graph (F(v) = indptr[v] / nnz, stored next to every CSR entry in `kf`, so it arrives with
    // the coalesced key load).  A row's neighbours are spread roughly uniformly by F
    // (neighbours are drawn ~ proportionally to degree), and the sorted keys of one wavefront load
    // hit consecutive filter words: a handful of cache lines per 64 keys instead of 64.
    const uint32_t *__restrict__ foff;
    const uint64_t *__restrict__ fbits;
    // kf[e] = { indices[e], fw } with fw = the top 22 bits of floor(2^32 * indptr[indices[e]] / nnz) (word
    // selector) | 10 hash bits of the neighbour id (its two filter bit positions): one 8-byte load per key,
    // no hashing in the walk kernel.
    const uint2 *__restrict__ kf;
    // exact adjacency index for the filter survivors: row u owns slots [tab_off[u], tab_off[u+1]) of an
    // open-addressing table (size next_pow2(2*degree)); slot = (position in row u) << 32 | neighbour id,
    // all ones = empty.  One probe (rarely two) replaces the ~log2(d) dependent probes of a binary
    // search -- the per-step critical path is a chain of memory latencies, not bandwidth.
    const uint64_t *__restrict__ tab_off;
    const uint64_t *__restrict__ slots;
    // vrec[v] = { indptr[v], degree(v), foff[v], tab_off[v] / 2 }: everything the walk needs to know about
    // a vertex in ONE 16-byte scalar load (filter and index sizes follow from the degree).
    const uint4 *__restrict__ vrec;
    // tri[4 e] (the first 16 bytes of the lane index's 64-byte edge line of CSR entry e = (u -> v), walk_lanes.hip.h)
    // = { v, |N(u) & N(v)|, position of u in row v (or 0xffffffff), degree(v) }: the
    // number of common neighbours of the two endpoints (a per-edge triangle count) and the place of
    // the reverse edge, built once.  With the count the normaliser `tot` of a step is known BEFORE any
    // membership work, so the membership test can stop as soon as the CDF search has found its element
    // (on average half of the keys are never touched); the reverse position is where `prev` sits in
    // cur's row on the next step (no index probe); v and its degree ride along so that one 16-byte load
    // names the next vertex and tells which of the two rows will supply the keys.  nullptr: not built
    // (graphs with self loops).
    const uint4 *__restrict__ tri;
    const uint8_t *__restrict__ clist;   // ... and the lists too long for their edge line
    // the CSR entry (prev -> cur) the step being sampled arrived by, or NOT_FOUND (first step, mirrored overflow
    // read, resumed walk): set in the per-step copy of the arguments; with it the membership mask of the step is
    // scattered from the entry's list instead of being searched (build_mask_list)
    uint32_t step_edge;
    uint32_t n_nodes;
    uint32_t nnz;
};

// Arithmetic flavour of the two reference paths:
//   float  (SparseOTF): float32 storage, biases applied through float64 (sparse_rw.py:84-87,
//                       Numba dd->d loop), node2vec+ ratio t in float32 (sparse_rw.py:262-264)
//   double (DenseOTF) : everything float64 (dense_rw.py:34-118)
template <typename T> struct Arith;
template <> struct Arith<float> {
    static __device__ __forceinline__ float bias_div(float w, double d) { return (float)((double)w / d); }
    static __device__ __forceinline__ float bias_mul(float w, double a) { return (float)((double)w * a); }
    static __device__ __forceinline__ double t_ratio(float u, float thr) { return (double)(u / thr); }
    static __device__ __forceinline__ bool in_edge(float u, float thr) { return u >= thr; }
    static __device__ __forceinline__ bool noisy(float w, float thr_cur) { return w < thr_cur; }
};
template <> struct Arith<double> {
    static __device__ __forceinline__ double bias_div(double w, double d) { return w / d; }
    static __device__ __forceinline__ double bias_mul(double w, double a) { return w * a; }
    static __device__ __forceinline__ double t_ratio(double u, float thr) { return u / (double)thr; }
    static __device__ __forceinline__ bool in_edge(double u, float thr) { return !(u < (double)thr); }
    static __device__ __forceinline__ bool noisy(double w, float thr_cur) { return w < (double)thr_cur; }
};

struct WalkArgs {
    CsrDev g;
    double p, q;
    uint32_t L;
    uint64_t n_jobs;
    const uint32_t *__restrict__ starts;
    const uint64_t *__restrict__ stream_off;  // per job: absolute index of its first double
    const uint32_t *__restrict__ job_list;    // optional: run only these jobs (repair passes)
    uint64_t n_list;
    const double *__restrict__ rng;           // rng[n - rng_base]
    uint64_t rng_base;
    uint32_t *out;                            // [n_jobs, L + 2]
    unsigned long long *job_counter;
    unsigned long long *stats;  // [0] steps [1] overflow reads [2] clamped reads [3] dead-end walks
    float w_out, w_prev;        // fl32(1/q), fl32(1/p): the unit-weight biases (host computed)
    uint32_t lazy_ok;           // both are powers of two in a safe exponent range (lazy path precondition)
    // weighted CSR graphs: the normaliser of every transition, precomputed per (p, q, extend) by tot_build_kernel
    // with the very code of the walk step (tot_e[e] = sequential float32 sum of the biased weights of row v for a
    // walker that arrived by CSR entry e = (u -> v); tot_v[v] = the unbiased row sum of a first step), or nullptr
    const float *__restrict__ tot_e;
    const float *__restrict__ tot_v;
    // walks handed over by the lane kernel in mid-walk: the row holds cells 0 .. len - 1 and len (>= 1) in its length
    // cell; the walk goes on from there instead of being walked again from its start
    uint32_t resume;
};
#define PW_KARG(T, field) kernarg<T>(offsetof(WalkArgs, field))

// Fresh copy of the kernel arguments for the rarely taken paths: read from the kernarg segment at the
// point of use, so that none of it stays live (or spilled) across the hot path.
__device__ __forceinline__ WalkArgs reload_walk_args() {
    WalkArgs a;
    a.g.indptr = (const uint32_t *)PW_KARG(uint64_t, g.indptr);
    a.g.indices = (const uint32_t *)PW_KARG(uint64_t, g.indices);
    a.g.data = (const void *)PW_KARG(uint64_t, g.data);
    a.g.thr = (const float *)PW_KARG(uint64_t, g.thr);
    a.g.adjbits = (const uint64_t *)PW_KARG(uint64_t, g.adjbits);
    a.g.words_per_row = PW_KARG(uint32_t, g.words_per_row);
    a.g.foff = (const uint32_t *)PW_KARG(uint64_t, g.foff);
    a.g.fbits = (const uint64_t *)PW_KARG(uint64_t, g.fbits);
    a.g.kf = (const uint2 *)PW_KARG(uint64_t, g.kf);
    a.g.tab_off = (const uint64_t *)PW_KARG(uint64_t, g.tab_off);
    a.g.slots = (const uint64_t *)PW_KARG(uint64_t, g.slots);
    a.g.vrec = (const uint4 *)PW_KARG(uint64_t, g.vrec);
    a.g.tri = (const uint4 *)PW_KARG(uint64_t, g.tri);
    a.g.clist = (const uint8_t *)PW_KARG(uint64_t, g.clist);
    a.g.step_edge = NOT_FOUND;
    a.g.n_nodes = PW_KARG(uint32_t, g.n_nodes);
    a.g.nnz = PW_KARG(uint32_t, g.nnz);
    a.p = PW_KARG(double, p);
    a.q = PW_KARG(double, q);
    a.L = PW_KARG(uint32_t, L);
    a.n_jobs = 0;
    a.starts = nullptr;
    a.stream_off = nullptr;
    a.job_list = nullptr;
    a.n_list = 0;
    a.rng = nullptr;
    a.rng_base = 0;
    a.out = nullptr;
    a.job_counter = nullptr;
    a.stats = nullptr;
    a.w_out = PW_KARG(float, w_out);
    a.w_prev = PW_KARG(float, w_prev);
    a.lazy_ok = PW_KARG(uint32_t, lazy_ok);
    a.tot_e = (const float *)PW_KARG(uint64_t, tot_e);
    a.tot_v = (const float *)PW_KARG(uint64_t, tot_v);
    a.resume = 0;
    return a;
}

// Optional per-section cycle accounting (-DPW_PROF builds only; tools/prof_sections.sh).
#ifdef PW_PROF
__device__ unsigned long long g_prof[16];
struct Prof {
    unsigned long long *acc;   // LDS, per wave: [0..7] cycles per section, [8..15] event counts
    unsigned long long last;
};
__device__ __forceinline__ void prof_tick(Prof &p, int i) {
    const unsigned long long now = __builtin_readcyclecounter();
    if (__lane_id() == 0) p.acc[i] += now - p.last;
    p.last = __builtin_readcyclecounter();
}
__device__ __forceinline__ void prof_count(Prof &p, int i, unsigned long long n = 1) {
    if (__lane_id() == 0) p.acc[i] += n;
}
#define PROF_TICK(p, i) prof_tick(p, i)
#define PROF_COUNT(p, i, n) prof_count(p, i, n)
#else
struct Prof {};
#define PROF_TICK(p, i)
#define PROF_COUNT(p, i, n)
#endif

__device__ __forceinline__ uint32_t filter_hash(uint32_t v) {
    uint32_t h = v * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    return h;
}
// fw of a neighbour id v with degree-CDF fraction frac (see CsrDev::kf)
__device__ __forceinline__ uint32_t filter_fw(uint32_t frac, uint32_t v) {
    return (frac & 0xFFFFFC00u) | (filter_hash(v) & 0x3FFu);
}
// word index inside the row's filter: nw = nw_mask + 1 words (power of two <= 2^22), the top log2(nw)
// bits of the CDF fraction
__device__ __forceinline__ uint32_t filter_word(uint32_t fw, uint32_t nw_mask) {
    return nw_mask ? (fw >> (32 - __popc(nw_mask))) : 0u;
}
// the two filter bits of a key: bit (fw & 31) of the low half, bit ((fw >> 5) & 31) of the high half
__device__ __forceinline__ uint64_t filter_bits(uint32_t fw) {
    return (1ull << (fw & 31u)) | (1ull << (32u + ((fw >> 5) & 31u)));
}
__device__ __forceinline__ bool filter_pass(uint64_t word, uint32_t fw) {
    return (__builtin_amdgcn_ubfe((uint32_t)word, fw, 1u) & __builtin_amdgcn_ubfe((uint32_t)(word >> 32), fw >> 5, 1u)) != 0u;
}
__host__ __device__ inline uint32_t filter_words_for_degree(uint32_t d) {
    if (d == 0) return 0;
    uint32_t p = 1;
    while (p < d) p <<= 1;      // next power of two >= d
    p = p >= 8 ? p / 8 : 1;     // 8 filter bits per (rounded) neighbour, at least one word
    return p > (1u << 22) ? (1u << 22) : p;   // the word selector has 22 bits
}

// sizes of a row's filter / adjacency index from its degree d >= 1 (the build code's rules, scalar ALU)
__device__ __forceinline__ uint32_t max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t next_pow2_u32(uint32_t x) {   // smallest power of two >= x, x >= 1
    return x <= 1u ? 1u : 1u << (32 - __builtin_clz(x - 1u));
}
__device__ __forceinline__ uint32_t filter_mask_for_degree(uint32_t d) {
    uint32_t p = next_pow2_u32(d);
    p = p >= 8u ? p >> 3 : 1u;
    return (p > (1u << 22) ? (1u << 22) : p) - 1u;
}
__device__ __forceinline__ uint32_t index_mask_for_degree(uint32_t d) { return next_pow2_u32(2u * d) - 1u; }

constexpr uint64_t SLOT_EMPTY = ~0ull;
__device__ __forceinline__ uint32_t adj_hash(uint32_t v, uint32_t size_mask) {
    uint32_t h = v * 0x85EBCA6Bu;  // independent of filter_hash
    h ^= h >> 16;
    h *= 0xC2B2AE35u;
    h ^= h >> 15;
    return h & size_mask;
}
// position of v in the row that owns `tab`, or 0xffffffff
__device__ __forceinline__ uint32_t adj_lookup(const uint64_t *__restrict__ tab, uint32_t size_mask, uint32_t v,
                                               bool active) {
    uint32_t idx = adj_hash(v, size_mask);
    uint32_t res = 0xffffffffu;
    while (active) {
        const uint64_t e = tab[idx];
        if (e == SLOT_EMPTY) active = false;
        else if ((uint32_t)e == v) { res = (uint32_t)(e >> 32); active = false; }
        else idx = (idx + 1) & size_mask;
    }
    return res;
}

constexpr int WAVES_PER_BLOCK = 4;
#ifndef PW_MASK_WORDS
#define PW_MASK_WORDS 512
#endif
constexpr int MASK_WORDS = PW_MASK_WORDS;        // per wave: 32*MASK_WORDS neighbours per segment
constexpr uint32_t SEG = MASK_WORDS * 32;
#ifndef PW_CHAIN_CKPT
#define PW_CHAIN_CKPT 256   // (round 6: 1024 -> 256: a parked step scans at most 256 elements + its window instead of 1024; C5 311 -> 282 ms
#endif                      //  per pass, 1 404 -> 1 553 M steps/s; 512: 292 ms; the records take 360 instead of 90 MB at weighted RMAT-20)
constexpr uint32_t CHAIN_CKPT = PW_CHAIN_CKPT;   // weighted lane form: spacing of the recorded chain values (divides SEG, multiple of the scan's 256-element trips)
constexpr int EPL = 4;                           // elements per lane per generic scan pass
__device__ __forceinline__ uint32_t uni(uint32_t v) { return readfirst_u32(v); }
__device__ __forceinline__ double uni(double v) {
    return __longlong_as_double((long long)readfirst_u64((uint64_t)__double_as_longlong(v)));
}
__device__ __forceinline__ float uni(float v) { return __uint_as_float(readfirst_u32(__float_as_uint(v))); }
__device__ __forceinline__ uint64_t uni(uint64_t v) { return readfirst_u64(v); }
__device__ __forceinline__ bool uni(bool v) { return readfirst_u32(v ? 1u : 0u) != 0u; }

// ---- lower_bound over a sorted global row; every lane searches its own key --------------------
// Branch-free form with a wave-uniform trip count (n only depends on the row length).
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ base, uint32_t n,
                                                    uint32_t key) {
    if (n == 0) return 0;
    uint32_t lo = 0, len = n;
    while (len > 1) {
        uint32_t half = len >> 1;
        lo = (base[lo + half - 1] < key) ? lo + half : lo;
        len -= half;
    }
    return lo + ((base[lo] < key) ? 1u : 0u);
}

// ---- step 1: membership bitmask of one segment [a, a+len) of cur's row ---------------------------
// mask bit (k - a) = 1  <=>  indices[s0 + k] is a neighbour of prev.   Returns (wave-uniform) the
// position of `prev` itself inside the segment, or NOT_FOUND.
// node2vec+ (in_mask != nullptr): a second bitmask marks the common neighbours x whose edge
// prev->x is not "noisy", w(prev, x) >= thr[x] (in-edges, isnotin_extended, sparse_rw.py:233-295).
//
// The entries of the shorter row are the keys, the longer row is searched.  Keys first pass the
// searched row's Bloom filter (one cache-line access each); the survivors (true members + a few
// false positives, typically < 20 % of the keys) are compacted into an LDS queue and only they
// are looked up exactly, in the searched row's adjacency hash index (one probe, rarely two).
constexpr uint32_t QCAP = 128;  // survivor queue entries per wave (key, tag)
constexpr uint32_t TAG_PREV = 0xffffffffu;

template <typename T>
__device__ __forceinline__ uint32_t build_mask(const CsrDev &g, uint32_t *mask, uint32_t *queue, uint32_t cur,
                                               uint32_t prev, uint32_t s0, uint32_t a, uint32_t len,
                                               uint32_t t0, uint32_t dp, uint32_t *in_mask,
                                               const T *__restrict__ data) {
    const int lane = lane_id();
    const float *__restrict__ thr = g.thr;
    uint32_t prev_pos = NOT_FOUND;
    const uint32_t nwords = (len + 31) >> 5;
    for (uint32_t w = lane; w < nwords; w += WAVE) {
        mask[w] = 0;
        if (in_mask) in_mask[w] = 0;
    }
    const bool scatter = dp <= len;                       // keys = row(prev), searched = cur's segment
    const uint2 *__restrict__ krow = g.kf + (scatter ? t0 : s0 + a);   // keys with their filter words
    const uint32_t kn = scatter ? dp : len;
    const uint32_t sv = scatter ? cur : prev;
    const uint32_t f0 = uni(g.foff[sv]);
    const uint32_t nw_mask = uni(g.foff[sv + 1]) - f0 - 1u;
    const uint64_t *__restrict__ fb = g.fbits + f0;
    uint32_t *qkey = queue, *qtag = queue + QCAP;
    uint32_t qn = 0;
    wave_lds_fence();

    // exact lookup of the queued survivors in the searched row's adjacency index
    const uint64_t tb0 = readfirst_u64(g.tab_off[sv]);
    const uint32_t tmask = (uint32_t)(readfirst_u64(g.tab_off[sv + 1]) - tb0) - 1u;
    const uint64_t *__restrict__ tab = g.slots + tb0;
    const uint32_t seg_lo = scatter ? a : 0u;                 // positions returned are row-global
    auto flush = [&](uint32_t count) {
        wave_lds_fence();
        for (uint32_t base = 0; base < count; base += WAVE) {
            const uint32_t e = base + lane;
            const bool valid = e < count;
            const uint32_t key = valid ? qkey[e] : 0u;
            const uint32_t tag = valid ? qtag[e] : 0u;
            const uint32_t gpos = adj_lookup(tab, tmask, key, valid);
            if (scatter) {
                // tag = index of the key in row(prev); gpos = position in cur's row
                const bool found = gpos != 0xffffffffu && gpos >= seg_lo && gpos < seg_lo + len;
                const uint32_t rel = gpos - seg_lo;
                if (found && tag != TAG_PREV) {
                    atomicOr(&mask[rel >> 5], 1u << (rel & 31));
                    if (in_mask && Arith<T>::in_edge(data[t0 + tag], thr[key])) atomicOr(&in_mask[rel >> 5], 1u << (rel & 31));
                }
                const uint64_t pb = ballot(found && tag == TAG_PREV);
                if (pb) prev_pos = readlane_u32(gpos, __builtin_ctzll(pb));
            } else if (gpos != 0xffffffffu) {
                // tag = position of the key in cur's segment; gpos = index in row(prev)
                atomicOr(&mask[tag >> 5], 1u << (tag & 31));
                if (in_mask && Arith<T>::in_edge(data[t0 + gpos], thr[key])) atomicOr(&in_mask[tag >> 5], 1u << (tag & 31));
            }
        }
        wave_lds_fence();
    };

    if (scatter) {  // prev itself is looked up in cur's row (no filter: it is almost always there)
        if (lane == 0) { qkey[0] = prev; qtag[0] = TAG_PREV; }
        qn = 1;
    }
    for (uint32_t base = 0; base < kn; base += WAVE) {
        const uint32_t i = base + lane;
        const bool valid = i < kn;
        const uint2 kfw = valid ? krow[i] : make_uint2(0u, 0u);
        const uint32_t key = kfw.x;
        const uint64_t word = valid ? fb[filter_word(kfw.y, nw_mask)] : 0ull;
        const bool pass = valid && filter_pass(word, kfw.y);
        if (!scatter) {
            const uint64_t pb = ballot(valid && key == prev);
            if (pb) prev_pos = a + base + (uint32_t)__builtin_ctzll(pb);
        }
        const uint64_t sb = ballot(pass);
        if (pass) {
            const uint32_t slot = qn + (uint32_t)__popcll(sb & ((1ull << lane) - 1ull));
            qkey[slot] = key;
            qtag[slot] = i;
        }
        qn += (uint32_t)__popcll(sb);
        if (qn > QCAP - WAVE) {
            flush(qn);
            qn = 0;
        }
    }
    if (qn) flush(qn);
    wave_lds_fence();
    return prev_pos;
}

// DenseOTF membership: the bit-packed row of prev answers "x in N(prev)" directly, so every
// neighbour of cur in the segment tests its own bit (coalesced read of cur's row + one word of
// prev's 12.5 KB/100k-column row per lane).
template <typename T>
__device__ __forceinline__ uint32_t build_mask_bits(const CsrDev &g, uint32_t *mask, uint32_t s0, uint32_t a,
                                                    uint32_t len, uint32_t t0, uint32_t dp, uint32_t prev,
                                                    uint32_t *in_mask, const T *__restrict__ data) {
    const int lane = lane_id();
    const uint32_t *__restrict__ crow = g.indices + s0 + a;
    const uint64_t *__restrict__ pbits = g.adjbits + (uint64_t)prev * g.words_per_row;
    uint32_t prev_pos = NOT_FOUND;
    for (uint32_t kb = 0; kb < len; kb += WAVE) {
        const uint32_t k = kb + lane;
        const bool valid = k < len;
        const uint32_t x = valid ? crow[k] : 0u;
        const bool found = valid && ((pbits[x >> 6] >> (x & 63)) & 1ull);
        const uint64_t fb = ballot(found);
        if (lane == 0) mask[kb >> 5] = (uint32_t)fb;
        if (lane == 32) mask[(kb >> 5) + 1] = (uint32_t)(fb >> 32);
        if (in_mask) {
            // w(prev, x): position of x in prev's row by search (only for common neighbours)
            bool is_in = false;
            if (fb) {
                const uint32_t jpos = lower_bound_u32(g.indices + t0, dp, x);
                is_in = found && Arith<T>::in_edge(data[t0 + jpos], g.thr[x]);
            }
            const uint64_t ib = ballot(is_in);
            if (lane == 0) in_mask[kb >> 5] = (uint32_t)ib;
            if (lane == 32) in_mask[(kb >> 5) + 1] = (uint32_t)(ib >> 32);
        }
        const uint64_t pb = ballot(valid && x == prev);
        if (pb) prev_pos = a + kb + __builtin_ctzll(pb);
    }
    wave_lds_fence();
    return prev_pos;
}

// ---- per-neighbour biased weight / probability ------------------------------------------------------
// Values of one segment of cur's row as the reference computes them:
//   w_k = data[k]; out edges: fl32(f64(w)/q); return edge: fl32(f64(w)/p)  (sparse_rw.py:84-87)
//   normalised: fl32(w_k / tot)                                             (sparse_rw.py:89)
template <typename T, bool UNIT> struct RowVals {
    const T *__restrict__ drow;      // data + s0 (unused when UNIT)
    const uint32_t *mask;            // LDS bitmask of the current segment
    uint32_t seg_a;                  // first neighbour index covered by mask
    uint32_t prev_pos;               // NOT_FOUND when prev is not a neighbour of cur
    uint32_t kend;                   // neighbours >= kend contribute 0
    bool has_prev;
    bool normalize;
    double p, q;
    T tot;
    T u_in, u_out, u_prev;      // UNIT: the three possible values (already normalised if asked)
    // node2vec+ (extend): get_extended_normalized_probs, sparse_rw.py:93-130 / dense_rw.py:74-118
    bool extend = false;
    const uint32_t *in_mask = nullptr;           // LDS: common neighbour that is an in-edge
    const uint32_t *__restrict__ crow = nullptr; // indices + s0
    const uint32_t *__restrict__ prow = nullptr; // indices + t0
    const T *__restrict__ pdata = nullptr;       // data + t0
    const float *__restrict__ thr = nullptr;
    uint32_t dp = 0;
    float thr_cur = 0.0f;
    const uint64_t *__restrict__ ptab = nullptr; // adjacency index of prev's row (nullptr: search prow)
    uint32_t ptmask = 0;
    // p / q that are powers of two: fl32(f64(w) / q) == w * (1/q) exactly (scaling by a power of two
    // is exact, one rounding either way), which avoids a ~30-instruction float64 division per element
    bool q_pow2 = false, p_pow2 = false;
    T inv_q = (T)0, inv_p = (T)0;

    __device__ __forceinline__ void setup_bias() {
        const uint64_t qb = (uint64_t)__double_as_longlong(q), pb = (uint64_t)__double_as_longlong(p);
        q_pow2 = (qb & 0xfffffffffffffull) == 0 && q > 0x1p-100 && q < 0x1p100;
        p_pow2 = (pb & 0xfffffffffffffull) == 0 && p > 0x1p-100 && p < 0x1p100;
        inv_q = (T)(1.0 / q);
        inv_p = (T)(1.0 / p);
    }
    __device__ __forceinline__ T div_q(T w) const { return q_pow2 ? w * inv_q : Arith<T>::bias_div(w, q); }
    __device__ __forceinline__ T div_p(T w) const { return p_pow2 ? w * inv_p : Arith<T>::bias_div(w, p); }

    __device__ __forceinline__ T value(uint32_t k, uint32_t bit) const {
        if (UNIT) {
            const T vi = u_in, vo = u_out, vp = u_prev;
            T v = bit ? vi : vo;
            return (has_prev && k == prev_pos) ? vp : v;
        } else {
            T w = drow[k];
            if (has_prev && !extend) {
                if (k == prev_pos) w = div_p(w);
                else if (!bit) w = div_q(w);
            }
            return normalize ? w / tot : w;
        }
    }

    // node2vec+ value of element k (valid: k < kend).  Every lane runs the (uniform trip count)
    // row search; only common neighbours that are out-edges need its result (t = w(prev,x)/thr[x]).
    __device__ __forceinline__ T value_ext(uint32_t k, bool valid) const {
        T w = valid ? drow[k] : (T)0;
        if (!has_prev) return (valid && normalize) ? w / tot : w;
        uint32_t r = valid ? k - seg_a : 0u;
        const bool common = valid && ((mask[r >> 5] >> (r & 31)) & 1u);
        const bool is_in = valid && ((in_mask[r >> 5] >> (r & 31)) & 1u);
        const bool need_t = common && !is_in && k != prev_pos;
        double t = 0.0;
        if (ballot(need_t)) {
            const uint32_t x = need_t ? crow[k] : 0u;
            // position of x in prev's row: one probe of the adjacency index (sparse graphs) or a
            // search of the row (dense handles have no index)
            const uint32_t jpos = ptab ? adj_lookup(ptab, ptmask, x, need_t) : lower_bound_u32(prow, dp, x);
            if (need_t) t = Arith<T>::t_ratio(pdata[jpos], thr[x]);
        }
        if (valid) {
            if (k == prev_pos) w = div_p(w);
            else if (!(common && is_in)) {
                const double inv_q = 1.0 / q;
                double alpha = inv_q + (1.0 - inv_q) * t;
                if (Arith<T>::noisy(w, thr_cur)) alpha = inv_q < 1.0 ? inv_q : 1.0;
                w = Arith<T>::bias_mul(w, alpha);
            }
            if (normalize) w = w / tot;
        }
        return w;
    }

    // one element (k < kend required)
    __device__ __forceinline__ T one(uint32_t k) const {
        if (!UNIT && extend) return value_ext(k, k < kend);
        uint32_t bit = 0;
        if (has_prev) { uint32_t r = k - seg_a; bit = (mask[r >> 5] >> (r & 31)) & 1u; }
        return value(k, bit);
    }

    // EPL consecutive elements starting at kb (kb multiple of EPL); 0 beyond kend
    __device__ __forceinline__ void vec(uint32_t kb, T (&xs)[EPL]) const {
        if (!UNIT && extend) {
#pragma unroll
            for (int e = 0; e < EPL; e++) xs[e] = value_ext(kb + e, kb + e < kend);
            return;
        }
        uint32_t bits = 0;
        if (has_prev && kb < kend) { uint32_t r = kb - seg_a; bits = mask[r >> 5] >> (r & 31); }
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            uint32_t k = kb + e;
            xs[e] = (k < kend) ? value(k, (bits >> e) & 1u) : (T)0;
        }
    }
};

// ---- steps 2/3: bit-exact sequential running sum, evaluated wave-parallel -----------------------------
// (round 6: the lanes' values travel as DPP operands -- row_shr 1, 2, 4, 8, row_bcast 15 / 31: wave.h -- instead of through the
//  LDS crossbar; a lane without a source receives (0, 0), the identity of the composition)
#ifndef PW_SCAN_DPP
#define PW_SCAN_DPP 1
#endif
template <typename T> __device__ __forceinline__ Inc<T> wave_scan_inc(Inc<T> f) {
#if !PW_SCAN_DPP
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
#endif
#define PW_STEP_(CTRL, RM) { Inc<T> g; g.a0 = dpp_get<CTRL, RM>(f.a0); g.a1 = dpp_get<CTRL, RM>(f.a1); f = Binade<T>::compose(g, f); }
    PW_DPP_SCAN_LEVELS(PW_STEP_)
#undef PW_STEP_
    return f;
}

// One binade of the running sum: processes elements from k on while `c` stays in its binade.
//   returns SCAN_END     : reached kend (k == kend, c updated)
//           SCAN_CROSSED : c left the binade at element k-1 (added with a real floating-point add)
//           SCAN_FOUND   : element `found` is the first with (double)c_k >= r   (HAS_TARGET only)
enum { SCAN_END = 0, SCAN_CROSSED = 1, SCAN_FOUND = 2 };

template <typename T, bool HAS_TARGET, typename Vals>
__device__ __forceinline__ int seq_scan_binade(T &c, uint32_t &k, uint32_t kend, double r,
                                               const Vals &vals, uint32_t &found) {
    using B = Binade<T>;
    using U = typename B::UInt;
    const int lane = lane_id();
    const int eb = B::eb_of(c);
    const U Tt = HAS_TARGET ? B::threshold(r, eb) : B::TOP;
    while (k < kend) {
        const U C = B::sig_of(c);
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
        bool lane_tie = false;
#pragma unroll
        for (int e = 0; e < EPL; e++) lane_tie |= f[e].a0 != f[e].a1;
        U Cincl;
        if (!ballot(lane_tie)) {
            // no exact tie in this pass (the common case): increments are parity independent, the
            // scan is a plain saturating integer prefix sum
            U sum = f[0].a0;
#pragma unroll
            for (int e = 1; e < EPL; e++) { sum += f[e].a0; sum = sum > B::SAT ? B::SAT : sum; }
#if PW_SCAN_DPP
#define PW_STEP_(CTRL, RM) { const U t = dpp_get<CTRL, RM>(sum); const U nsum = sum + t; sum = nsum > B::SAT ? B::SAT : nsum; }
            PW_DPP_SCAN_LEVELS(PW_STEP_)
#undef PW_STEP_
#else
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                U t = shfl_up_uint<U>(sum, off);
                U nsum = sum + t;
                nsum = nsum > B::SAT ? B::SAT : nsum;
                if (lane >= off) sum = nsum;
            }
#endif
            Cincl = C + sum;
        } else {
            Inc<T> agg = f[0];
#pragma unroll
            for (int e = 1; e < EPL; e++) agg = B::compose(agg, f[e]);
            Inc<T> incl = wave_scan_inc<T>(agg);
            Cincl = B::apply(C, incl);
        }
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
        if (Cn < B::TOP) { c = B::make(Cn, eb); found = kf; return SCAN_FOUND; }  // target reached
        // the sum leaves the binade at element kf: one real floating-point add
        c = B::make(Cprev, eb) + xf;
        k = kf + 1;
        if (HAS_TARGET && (double)c >= r) { found = kf; return SCAN_FOUND; }
        return SCAN_CROSSED;
    }
    if (k > kend) k = kend;
    return SCAN_END;
}

// Leading elements added one by one (cheap while the sum doubles every few elements and would
// otherwise leave its binade on almost every pass).  Returns true when the target was reached.
// The adds run without per-element target checks first (the sum is monotone: if the last partial
// sum is below r no element reached it); only a head that contains the target is replayed.
template <typename T, bool HAS_TARGET, typename Vals>
__device__ __forceinline__ bool seq_head(T &c, uint32_t &k, uint32_t kend, double r, const Vals &vals,
                                         uint32_t head, uint32_t &found) {
    const int lane = lane_id();
    uint32_t n = kend - k < head ? kend - k : head;
    T v = (lane < (int)n) ? (T)vals.one(k + lane) : (T)0;
    T cc = c;
    if (n == WAVE) {  // full head: straight-line code (no loop control on the scalar unit)
#pragma unroll
        for (int j = 0; j < WAVE; j++) cc = cc + readlane_fp<T>(v, j);
    } else {
        for (uint32_t j = 0; j < n; j++) cc = cc + readlane_fp<T>(v, (int)j);
    }
    cc = uni(cc);
    if (!HAS_TARGET || (double)cc < r) {
        c = cc;
        k += n;
        return false;
    }
    for (uint32_t j = 0; j < n; j++) {
        c = uni(c + readlane_fp<T>(v, (int)j));
        if ((double)c >= r) { found = k + j; return true; }
    }
    k += n;  // unreachable: cc >= r guarantees a hit above
    return false;
}

// Continues the running sum `c` over elements [kbeg, kend) of `vals`.
//   HAS_TARGET: returns the first k with (double)c_k >= r, or NOT_FOUND; c is updated either way.
template <typename T, bool HAS_TARGET, typename Vals>
__device__ __forceinline__ uint32_t seq_scan(T &c, uint32_t kbeg, uint32_t kend, double r,
                                             const Vals &vals, uint32_t head) {
    uint32_t k = kbeg, found = NOT_FOUND;
    if (head && seq_head<T, HAS_TARGET>(c, k, kend, r, vals, head, found)) return found;
    while (k < kend) {
        int rc = seq_scan_binade<T, HAS_TARGET>(c, k, kend, r, vals, found);
        if (rc == SCAN_FOUND) return found;
    }
    return NOT_FOUND;
}

// ---- unit-weight fast path: run-length closed form of the same chain -------------------------------
// On an unweighted graph every neighbour of `cur` carries one of three values (common neighbour of
// prev: 1, other: 1/q, prev itself: 1/p; after normalisation x_in, x_out, x_prev).  Inside one binade
// adding a fixed value is adding a fixed integer number of ulps (seqscan.h), so the partial sum
// after element k is   C0 + n_in(k)*inc_in + n_out(k)*inc_out + n_prev(k)*inc_prev   with the
// class counts read from the membership bitmask (rank = prefix popcount).  The first element
// reaching the target / the binade top is found with a 64-ary search over k (one probe per lane),
// i.e. O(log64 d) per binade instead of touching all d elements.  Exact ties (round-half-even,
// parity dependent) fall back to the generic element scan for that binade.
struct UnitRow {
    const uint32_t *mask;   // LDS: bit (k - seg_a) set <=> common neighbour (prev's own bit cleared)
    const uint16_t *rank;   // LDS: rank[w] = popcount(mask[0..w))
    uint32_t seg_a, seg_len;
    uint32_t prev_pos;      // global position of prev in cur's row, or NOT_FOUND
    bool has_prev;
    // number of common neighbours among segment elements [seg_a, k), seg_a <= k <= seg_a + seg_len
    __device__ __forceinline__ uint32_t rank_at(uint32_t k) const {
        if (!has_prev) return 0;
        uint32_t r = k - seg_a, w = r >> 5, b = r & 31;
        uint32_t base = rank[w];
        return b ? base + (uint32_t)__popc(mask[w] & ((1u << b) - 1u)) : base;
    }
    __device__ __forceinline__ uint32_t bit_at(uint32_t k) const {
        if (!has_prev) return 0;
        uint32_t r = k - seg_a;
        return (mask[r >> 5] >> (r & 31)) & 1u;
    }
};

__device__ __forceinline__ void build_rank(const uint32_t *mask, uint16_t *rank, uint32_t nwords) {
    const int lane = lane_id();
    const uint32_t per = (nwords + WAVE - 1) / WAVE;
    const uint32_t w0 = (uint32_t)lane * per;
    uint32_t local = 0;
    for (uint32_t i = 0; i < per; i++) {
        uint32_t w = w0 + i;
        if (w < nwords) local += (uint32_t)__popc(mask[w]);
    }
    uint32_t incl = local;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        uint32_t t = shfl_up_uint<uint32_t>(incl, off);
        if (lane >= off) incl += t;
    }
    uint32_t run = incl - local;
    for (uint32_t i = 0; i < per; i++) {
        uint32_t w = w0 + i;
        if (w < nwords) {
            rank[w] = (uint16_t)run;
            run += (uint32_t)__popc(mask[w]);
        }
    }
    if (lane == WAVE - 1) rank[nwords] = (uint16_t)incl;
    wave_lds_fence();
}

// cnt * inc without wrap-around (float64 increments reach 2^54; anything >= 2^61 only ever needs to
// compare as "beyond the binade top").
template <typename U> __device__ __forceinline__ uint64_t chain_term(uint32_t cnt, U inc);
template <> __device__ __forceinline__ uint64_t chain_term<uint32_t>(uint32_t cnt, uint32_t inc) {
    return (uint64_t)cnt * inc;
}
template <> __device__ __forceinline__ uint64_t chain_term<uint64_t>(uint32_t cnt, uint64_t inc) {
    const uint64_t lo = (uint64_t)cnt * inc;
    const uint64_t hi = __umul64hi((uint64_t)cnt, inc);
    return (hi || (lo >> 61)) ? (1ull << 61) : lo;
}

template <typename T, bool HAS_TARGET>
__device__ __forceinline__ int unit_chain(T &c, uint32_t &k, uint32_t kend, double r, const UnitRow &ur,
                                          const RowVals<T, true> &rv, T x_in, T x_out, T x_prev,
                                          uint32_t &found) {
    using B = Binade<T>;
    using U = typename B::UInt;
    const int lane = lane_id();
    x_in = uni(x_in);
    x_out = uni(x_out);
    x_prev = uni(x_prev);
    c = uni(c);
    while (k < kend) {
        const int eb = B::eb_of(c);
        const U C = B::sig_of(c);
        const U Tt = HAS_TARGET ? uni(B::threshold(r, eb)) : B::TOP;
        const Inc<T> qi = B::quantize(x_in, eb), qo = B::quantize(x_out, eb), qp = B::quantize(x_prev, eb);
        const bool prev_in = ur.has_prev && ur.prev_pos != NOT_FOUND && ur.prev_pos >= k && ur.prev_pos < kend;
        // ties are parity dependent: hand this binade to the generic element scan (rare)
        if (qi.a0 != qi.a1 || qo.a0 != qo.a1 || (prev_in && qp.a0 != qp.a1)) {
            int rc = seq_scan_binade<T, HAS_TARGET>(c, k, kend, r, rv, found);
            if (rc == SCAN_FOUND) return SCAN_FOUND;
            continue;
        }
        const U ii = qi.a0, io = qo.a0, ipv = qp.a0;
        const uint32_t rk0 = uni(ur.rank_at(k));
        uint32_t lo = k, hi = kend - 1, kf = 0;
        uint64_t Cf = 0;
        bool crossed = true;
        for (;;) {
            const uint32_t n = hi - lo + 1;
            const uint32_t step = (n + WAVE - 1) / WAVE;
            uint64_t kp64 = (uint64_t)lo + (uint64_t)(lane + 1) * step - 1;
            const uint32_t kp = kp64 > hi ? hi : (uint32_t)kp64;
            const uint32_t cin = ur.rank_at(kp + 1) - rk0;
            const uint32_t cpv = (prev_in && ur.prev_pos <= kp) ? 1u : 0u;
            const uint32_t cout = (kp + 1 - k) - cin - cpv;
            const uint64_t G = (uint64_t)C + chain_term<U>(cin, ii) + chain_term<U>(cout, io) + chain_term<U>(cpv, ipv);
            const uint64_t hitm = ballot(G >= (uint64_t)Tt);
            if (!hitm) {  // only possible in the first round: the whole range stays below Tt
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
            c = B::make((U)Cf, eb);
            k = kend;
            break;
        }
        const bool f_prev = prev_in && kf == ur.prev_pos;
        const bool f_in = !f_prev && uni(ur.bit_at(kf)) != 0u;
        const T xf = f_prev ? x_prev : (f_in ? x_in : x_out);
        // value just before element kf (exact: every element before kf kept the sum below Tt)
        const uint32_t cin0 = uni(ur.rank_at(kf)) - rk0;
        const uint32_t cpv0 = (prev_in && ur.prev_pos < kf) ? 1u : 0u;
        const uint32_t cout0 = (kf - k) - cin0 - cpv0;
        const uint64_t Cprev = (uint64_t)C + chain_term<U>(cin0, ii) + chain_term<U>(cout0, io) + chain_term<U>(cpv0, ipv);
        if (Cf < (uint64_t)B::TOP) { c = B::make((U)Cf, eb); found = kf; return SCAN_FOUND; }
        c = uni(B::make((U)Cprev, eb) + xf);
        k = kf + 1;
        if (HAS_TARGET && (double)c >= r) { found = kf; return SCAN_FOUND; }
    }
    return SCAN_END;
}

template <typename T> __device__ __forceinline__ bool is_pow2_fp(T x) {
    using U = typename FloatTraits<T>::UInt;
    return (FloatTraits<T>::bits(x) & (((U)1 << FloatTraits<T>::MANT) - 1)) == 0;
}

// Membership mask of one segment from the lane index (walk_lanes.hip.h): the walker arrived by CSR entry e = (prev ->
// cur), whose list holds the positions in cur's row of the common neighbours of prev and cur -- the mask is those
// positions, scattered; no key stream, no filter, no index probe.  node2vec+ (in_mask): the i-th entry of the REVERSE
// entry's list (cur -> prev: same common neighbours, same order, positions in prev's row) locates w(prev, x).
// r0 = the entry's record { cur, n_in, position of prev in cur's row, degree(cur) }.
template <typename T>
__device__ __forceinline__ uint32_t build_mask_list(const CsrDev &g, uint32_t *mask, uint32_t e, const uint4 r0, uint32_t s0,
                                                    uint32_t a, uint32_t len, uint32_t t0, uint32_t dp, uint32_t *in_mask,
                                                    const T *__restrict__ data) {
    const int lane = lane_id();
    const uint32_t nwords = (len + 31) >> 5;
    for (uint32_t w = lane; w < nwords; w += WAVE) {
        mask[w] = 0;
        if (in_mask) in_mask[w] = 0;
    }
    wave_lds_fence();
    const ELine *lines = (const ELine *)g.tri;
    const uint32_t n_in = r0.y, rev = r0.z, d = r0.w;
    const ListView P = edge_list(lines, g.clist, e, d, n_in, lines[e].coff);
    // list entries whose position lies in [a, a + len) (the list is ascending; one segment: all of them)
    uint32_t lo_i = 0, hi_i = n_in;
    if (a != 0 || len < d) {   // (the list's pivots, in the entry's own line, resolve the upper levels of both searches)
        lo_i = a ? list_lower_bound_pos(P, n_in, a) : 0u;
        hi_i = a + len < d ? list_lower_bound_pos(P, n_in, a + len) : n_in;
    }
    ListView Q = P;
    if (in_mask) {
        const uint32_t e2 = s0 + rev;   // (the caller checked that the reverse entry exists)
        Q = edge_list(lines, g.clist, e2, dp, lines[e2].n_in, lines[e2].coff);
    }
    for (uint32_t i = lo_i + (uint32_t)lane; i < hi_i; i += WAVE) {
        const uint32_t pos = P.at(i), rel = pos - a;
        atomicOr(&mask[rel >> 5], 1u << (rel & 31));
        if (in_mask) {
            const uint32_t x = g.indices[s0 + pos];
            if (Arith<T>::in_edge(data[t0 + Q.at(i)], g.thr[x])) atomicOr(&in_mask[rel >> 5], 1u << (rel & 31));
        }
    }
    wave_lds_fence();
    return (rev != NOT_FOUND && rev >= a && rev < a + len) ? rev : NOT_FOUND;
}

// Membership mask of one segment: from the arriving entry's list (sparse, lane index present), search-based (sparse)
// or bit-test based (dense adjacency bits).
template <typename T, bool DENSE>
__device__ __forceinline__ uint32_t segment_mask(const CsrDev &g, uint32_t *mask, uint32_t *in_mask,
                                                 uint32_t *queue, uint32_t cur, uint32_t s0, uint32_t sa,
                                                 uint32_t len, uint32_t t0, uint32_t dp, uint32_t prev) {
    const T *__restrict__ data = (const T *)g.data;
    if (DENSE) return build_mask_bits<T>(g, mask, s0, sa, len, t0, dp, prev, in_mask, data);
    if (g.tri && g.step_edge != NOT_FOUND) {
        const uint32_t e = uni(g.step_edge);
        const uint4 r0 = g.tri[4ull * e];
        const uint4 u0 = make_uint4(uni(r0.x), uni(r0.y), uni(r0.z), uni(r0.w));
        // (node2vec+ on a directed graph without the reverse entry: the weight w(prev, x) has to be searched)
        // (and the list has to be there: the index may have left the longest ones out -- EL_NO_LIST)
        const bool stored = edge_list_stored(u0.w, u0.y, uni(((const ELine *)g.tri)[e].coff)) &&
                            (!in_mask || u0.z == NOT_FOUND ||
                             edge_list_stored(dp, uni(((const ELine *)g.tri)[s0 + u0.z].n_in), uni(((const ELine *)g.tri)[s0 + u0.z].coff)));
        if (stored && u0.x == cur && (!in_mask || u0.z != NOT_FOUND)) return build_mask_list<T>(g, mask, e, u0, s0, sa, len, t0, dp, in_mask, data);
    }
    return build_mask<T>(g, mask, queue, cur, prev, s0, sa, len, t0, dp, in_mask, data);
}

// Membership structures (mask + rank) of one segment; returns prev's position (global) or
// NOT_FOUND.  known_prev_pos: position found by the caller for segmented rows (else NOT_FOUND).
template <typename T, bool DENSE>
__device__ __forceinline__ uint32_t prepare_unit_segment(const CsrDev &g, uint32_t *mask, uint16_t *rank,
                                                         uint32_t *queue, uint32_t cur, uint32_t s0, uint32_t sa, uint32_t len, uint32_t t0,
                                                         uint32_t dp, uint32_t prev, bool multi,
                                                         uint32_t known_prev_pos) {
    uint32_t pp = segment_mask<T, DENSE>(g, mask, nullptr, queue, cur, s0, sa, len, t0, dp, prev);
    if (multi) pp = known_prev_pos;
    if (pp != NOT_FOUND && pp >= sa && pp < sa + len) {
        uint32_t rr = pp - sa;  // keep the three classes disjoint
        if (lane_id() == 0) mask[rr >> 5] &= ~(1u << (rr & 31));
        wave_lds_fence();
    }
    build_rank(mask, rank, (len + 31) >> 5);
    return pp;
}

// Value view of one segment for the head / tie fallback.  Built fresh (never mutated) so that it
// stays in registers: a struct that lives in scratch turns every value() into a scratch load.
template <typename T>
__device__ __forceinline__ RowVals<T, true> make_unit_vals(const uint32_t *mask, uint32_t sa, uint32_t kend,
                                                            uint32_t prev_pos, bool has_prev, T v_in, T v_out,
                                                            T v_prev) {
    RowVals<T, true> rv;
    rv.drow = nullptr;
    rv.mask = mask;
    rv.seg_a = sa;
    rv.prev_pos = prev_pos;
    rv.kend = kend;
    rv.has_prev = has_prev;
    rv.normalize = false;
    rv.p = 1.0;
    rv.q = 1.0;
    rv.tot = (T)1;
    rv.u_in = v_in;
    rv.u_out = has_prev ? v_out : v_in;
    rv.u_prev = v_prev;
    return rv;
}

// Unit-weight transition: same result as the generic path, closed-form chain.
// (t0, dp) = CSR row of prev, carried over from the previous step by the caller.
template <typename T, bool DENSE>
__device__ __forceinline__ uint32_t sample_step_unit(const WalkArgs &a, uint32_t *mask, uint16_t *rank,
                                                     uint32_t *queue, uint32_t cur, bool has_prev, uint32_t prev, uint32_t t0,
                                                     uint32_t dp, double r, uint32_t s0, uint32_t d) {
    const uint32_t *__restrict__ indices = a.g.indices;
    const T w_in = (T)1, w_out = has_prev ? uni(Arith<T>::bias_div((T)1, a.q)) : (T)1,
            w_prev = uni(Arith<T>::bias_div((T)1, a.p));

    const bool multi = has_prev && d > SEG;
    uint32_t prev_pos = NOT_FOUND;
    if (multi) {
        uint32_t pos = uni(lower_bound_u32(indices + s0, d, prev));
        if (pos < d && uni(indices[s0 + pos]) == prev) prev_pos = pos;
    }

    // ---- tot = sequential sum of the biased weights ------------------------------------------------
    T tot = (T)0;
    bool have_tot = false;
    if (!multi) {
        if (has_prev) prev_pos = prepare_unit_segment<T, DENSE>(a.g, mask, rank, queue, cur, s0, 0, d, t0, dp, prev, false, NOT_FOUND);
        const UnitRow ur{mask, rank, 0u, d, prev_pos, has_prev};
        // all partial sums are exact when the weights are dyadic and the total fits the mantissa of
        // the smallest weight: then the left-to-right sum equals the exact sum.
        const uint32_t n_in = has_prev ? uni(ur.rank_at(d)) : 0u;
        const uint32_t n_pv = (has_prev && prev_pos != NOT_FOUND) ? 1u : 0u;
        const uint32_t n_out = d - n_in - n_pv;
        if ((n_out == 0 || is_pow2_fp<T>(w_out)) && (n_pv == 0 || is_pow2_fp<T>(w_prev))) {
            T u = (T)1;
            if (n_out && w_out < u) u = w_out;
            if (n_pv && w_prev < u) u = w_prev;
            double td = (double)n_in + (double)n_out * (double)w_out + (double)n_pv * (double)w_prev;
            if (uni(td / (double)u <= (double)Binade<T>::TOP)) { tot = uni((T)td); have_tot = true; }
        }
    }
    if (!have_tot) {
        for (uint32_t sa = 0; sa < d; sa += SEG) {
            const uint32_t len = d - sa < SEG ? d - sa : SEG;
            if (multi) (void)prepare_unit_segment<T, DENSE>(a.g, mask, rank, queue, cur, s0, sa, len, t0, dp, prev, true, prev_pos);
            const UnitRow ur{mask, rank, sa, len, prev_pos, has_prev};
            const RowVals<T, true> rv = make_unit_vals<T>(mask, sa, sa + len, prev_pos, has_prev, w_in, w_out, w_prev);
            uint32_t k = sa, found = NOT_FOUND;
            if (sa == 0) (void)seq_head<T, false>(tot, k, sa + len, 0.0, rv, WAVE, found);
            (void)unit_chain<T, false>(tot, k, sa + len, 0.0, ur, rv, w_in, w_out, w_prev, found);
        }
    }

    // ---- cdf search ----------------------------------------------------------------------------------
    tot = uni(tot);
    const T x_in = uni(w_in / tot), x_out = uni(w_out / tot), x_prev = uni(w_prev / tot);
    T c = (T)0;
    for (uint32_t sa = 0; sa < d; sa += SEG) {
        const uint32_t len = d - sa < SEG ? d - sa : SEG;
        // single segment: mask and rank from the tot phase are still valid
        if (multi) (void)prepare_unit_segment<T, DENSE>(a.g, mask, rank, queue, cur, s0, sa, len, t0, dp, prev, true, prev_pos);
        const UnitRow ur{mask, rank, sa, len, prev_pos, has_prev};
        const RowVals<T, true> rv = make_unit_vals<T>(mask, sa, sa + len, prev_pos, has_prev, x_in, x_out, x_prev);
        uint32_t k = sa, found = NOT_FOUND;
        if (sa == 0 && seq_head<T, true>(c, k, sa + len, r, rv, WAVE, found)) return found;
        if (unit_chain<T, true>(c, k, sa + len, r, ur, rv, x_in, x_out, x_prev, found) == SCAN_FOUND) return found;
    }
    return d;
}

// ---- lazy unit-weight transition ------------------------------------------------------------------------
// Same result as sample_step_unit, but membership is established progressively, in row order, and
// only as far as the CDF search needs it:
//   tot     = from the per-edge common-neighbour count tri[e(prev->cur)] (closed form, exact)
//   keys    = chunks of 64 entries of the shorter row, ascending; after a chunk every position of
//             cur's row below `known_end` has its final class (gather: the chunk itself; scatter:
//             everything up to the last common neighbour found so far)
//   search  = the closed-form chain runs over [k, known_end) whenever the exact-arithmetic mass of
//             the known prefix reaches r (minus a bound on the float drift), and resumes later if
//             it ended short.
// Returns LAZY_FALLBACK when a precondition fails (caller uses the eager path).
constexpr uint32_t LAZY_FALLBACK = 0xfffffffeu;

__device__ __forceinline__ uint32_t adj_lookup_g(gptr<uint64_t> tab, uint32_t size_mask, uint32_t v, bool active) {
    uint32_t idx = adj_hash(v, size_mask);
    uint32_t res = 0xffffffffu;
    while (active) {
        const uint64_t e = tab[idx];
        if (e == SLOT_EMPTY) active = false;
        else if ((uint32_t)e == v) { res = (uint32_t)(e >> 32); active = false; }
        else idx = (idx + 1) & size_mask;
    }
    return res;
}
// the same probe sequence with wave-uniform operands: runs on the scalar unit
__device__ __forceinline__ uint32_t adj_lookup_s(sptr<uint64_t> tab, uint32_t size_mask, uint32_t v) {
    uint32_t idx = adj_hash(v, size_mask);
    for (;;) {
        const uint64_t e = tab[idx];
        if (e == SLOT_EMPTY) return 0xffffffffu;
        if ((uint32_t)e == v) return (uint32_t)(e >> 32);
        idx = (idx + 1) & size_mask;
    }
}

// Exact-arithmetic companion of unit_chain.  E(k) = mass of elements 0..k in units of the smallest
// weight (integers, see the lazy step).  Returns the first k in [0, kend) with E(k) >= th and E(k)
// through e_at, or kend when E(kend - 1) < th.  One 64-ary search, no binades.
__device__ __forceinline__ uint32_t unit_search_units(const UnitRow &ur, uint32_t kend, uint32_t th, uint32_t sh_in,
                                                      uint32_t sh_out, uint32_t sh_prev, uint32_t &e_at) {
    const int lane = lane_id();
    uint32_t lo = 0, hi = kend - 1;
    for (;;) {
        const uint32_t n = hi - lo + 1;
        const uint32_t step = (n + WAVE - 1) / WAVE;
        const uint64_t kp64 = (uint64_t)lo + (uint64_t)(lane + 1) * step - 1;
        const uint32_t kp = kp64 > hi ? hi : (uint32_t)kp64;
        const uint32_t cin = ur.rank_at(kp + 1);
        const uint32_t cpv = ur.prev_pos <= kp ? 1u : 0u;   // NOT_FOUND compares greater than any position
        const uint32_t cout = kp + 1 - cin - cpv;
        const uint32_t G = (cin << sh_in) + (cout << sh_out) + (cpv << sh_prev);
        const uint64_t hitm = ballot(G >= th);
        if (!hitm) return kend;   // first round only
        const int first = __builtin_ctzll(hitm);
        if (step == 1) {
            e_at = readlane_u32(G, first);
            return lo + (uint32_t)first;
        }
        const uint64_t nhi = (uint64_t)lo + (uint64_t)(first + 1) * step - 1;
        lo += (uint32_t)first * step;
        if (nhi < hi) hi = (uint32_t)nhi;
    }
}

// Vertex context carried by the walk loop: row start/degree plus the offsets of the row's filter and
// adjacency index (one vrec load when the vertex is entered).  (solve_out_run: seqscan.h)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct VertexCtx {
    uint32_t s0, d, f0, tb;   // tb = tab_off / 2
};

__device__ __forceinline__ uint32_t sample_step_unit_lazy(uint32_t *mask, uint16_t *rank, uint32_t cur,
                                                          uint32_t prev, uint32_t n_in, uint32_t prev_pos,
                                                          bool keys_preloaded, uint64_t kfw_pre,
                                                          const VertexCtx &vc, const VertexCtx &vp, double r,
                                                          Prof &pf) {
    const int lane = lane_id();
    const uint32_t s0 = vc.s0, d = vc.d, t0 = vp.s0, dp = vp.d;
    // The step is a chain of memory latencies.  The vertex records of cur and prev, the common-neighbour
    // count of the edge and prev's position in cur's row (prev_pos, 0xffffffff == not a neighbour) all
    // arrived with the previous step; the first thing requested here are the keys.
    const float w_out = PW_KARG(float, w_out), w_prev = PW_KARG(float, w_prev);
    const uint64_t p_slots = PW_KARG(uint64_t, g.slots);
    const uint64_t p_kf = PW_KARG(uint64_t, g.kf), p_fbits = PW_KARG(uint64_t, g.fbits);
    const bool scatter = dp <= d;
    const uint32_t k0 = scatter ? t0 : s0;
    const uint32_t kn = scatter ? dp : d;
    const gptr<uint64_t> krow = as_global<uint64_t>(p_kf) + k0;   // fw << 32 | key
    // first 64 keys: already requested by the previous step when they come from prev's row
    uint64_t kfw_first = kfw_pre;
    if (n_in && !keys_preloaded && (uint32_t)lane < kn) kfw_first = krow[lane];
    const uint32_t ctmask = index_mask_for_degree(d);
    const uint32_t n_pv = prev_pos != NOT_FOUND ? 1u : 0u;
    if (n_in + n_pv > d) return LAZY_FALLBACK;
    const uint32_t n_out = d - n_in - n_pv;
    float u = 1.0f;
    if (n_out && w_out < u) u = w_out;
    if (n_pv && w_prev < u) u = w_prev;
    const double td = (double)n_in + (double)n_out * (double)w_out + (double)n_pv * (double)w_prev;
    if (!(td <= 16777216.0 * (double)u)) return LAZY_FALLBACK;   // every partial sum exact: tot = exact sum
    const float tot = uni((float)td);
    // w_out, w_prev are powers of two: w / tot == w * (1 / tot) exactly (one correctly rounded division)
    const float x_in = uni(1.0f / tot), x_out = uni(x_in * w_out), x_prev = uni(x_in * w_prev);
    // Search trigger (a heuristic, exactness not needed): mass of a prefix in units of the smallest
    // weight u -- integers below 2^24 by the precondition above, so plain scalar arithmetic.
    const uint32_t sh_u = (__float_as_uint(u) >> 23) & 0xffu;
    const uint32_t sh_in = (127u - sh_u) & 31u,   // classes that do not occur in the row get shift 0 (their count is 0)
                   sh_out = n_out ? (((__float_as_uint(w_out) >> 23) & 0xffu) - sh_u) & 31u : 0u,
                   sh_prev = n_pv ? (((__float_as_uint(w_prev) >> 23) & 0xffu) - sh_u) & 31u : 0u;
    const double units = ldexp(td, (int)(127u - sh_u));           // td / u (exact: u is a power of two)
    const uint32_t units_i = uni((uint32_t)units);
    const uint32_t r_units = uni((uint32_t)(r * units));

    const VertexCtx &vs = scatter ? vc : vp;                    // the searched row: cur (scatter) or prev
    const uint32_t nw_mask = filter_mask_for_degree(vs.d);
    const gptr<uint64_t> fb = as_global<uint64_t>(p_fbits) + vs.f0;
    const uint32_t tmask = scatter ? ctmask : index_mask_for_degree(dp);
    const gptr<uint64_t> tab = as_global<uint64_t>(p_slots) + 2ull * vs.tb;

    if (n_in == 0 || kn <= WAVE) {
        // Closed-form decision without the LDS mask.  Either there is no common neighbour at all (a fifth
        // of the edges of an R-MAT graph: every position is "out" except prev's), or all keys fit one
        // 64-key chunk (two fifths of the steps): the common neighbours are then at most 64 known
        // positions P_0 < P_1 < ..., one per lane, and between them the row consists of "out" runs, so the
        // first position whose exact mass reaches the target follows from a ballot and scalar arithmetic.
        const ExactThresholds th = exact_thresholds_f32(r * units, d, 1u << max_u32(sh_in, max_u32(sh_out, sh_prev)));
        const uint32_t lo_th = uni(th.lo), hi_th = uni(th.hi);   // drift bound: seqscan.h
        const uint32_t wp = 1u << sh_prev;
        const uint32_t pp = n_pv ? prev_pos : NOT_FOUND;
        uint32_t k1, e1;
        if (n_in == 0) {
            k1 = solve_out_run(0u, 0u, lo_th, pp, sh_out, wp, e1);
        } else {
            // classify the (at most 64) keys: position of every common neighbour in cur's row
            const bool valid = (uint32_t)lane < kn;
            const uint32_t key = (uint32_t)kfw_first, fw = (uint32_t)(kfw_first >> 32);
            const uint64_t word = valid ? fb[filter_word(fw, nw_mask)] : 0ull;
            const bool pass = valid && filter_pass(word, fw);
            const uint32_t gpos = adj_lookup_g(tab, tmask, key, pass);
            const bool hit = pass && gpos != 0xffffffffu;
            const uint32_t P = scatter ? gpos : (uint32_t)lane;     // keys ascend, so do the positions
            const uint64_t hb = ballot(hit);
            const uint32_t i = (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));   // common neighbours before this one
            const uint32_t pv = (pp != NOT_FOUND && pp < P) ? 1u : 0u;
            const uint32_t e_before = ((P - i - pv) << sh_out) + (i << sh_in) + (pv << sh_prev);   // E(P - 1)
            const uint32_t e_at = e_before + (1u << sh_in);                                         // E(P)
            const uint64_t reach = ballot(hit && e_at >= lo_th);
            // the run of "out" positions that holds the target: after the last common neighbour below it
            uint64_t below = hb;                                   // common neighbours before the run
            bool in_run = true;
            int f = -1;
            if (reach) {
                f = __builtin_ctzll(reach);
                below = hb & ((1ull << f) - 1ull);
                // reached before P_f already?  (only possible if there is a position before it: P_f == 0 with
                // a target of zero is the common neighbour itself)
                in_run = readlane_u32(e_before, f) >= lo_th && readlane_u32(P, f) != 0u;
            }
            if (!in_run) {
                k1 = readlane_u32(P, f);
                e1 = readlane_u32(e_at, f);
            } else {
                uint32_t s_run = 0, base = 0;
                if (below) {
                    const int pl = 63 - __builtin_clzll(below);
                    s_run = readlane_u32(P, pl) + 1u;
                    base = readlane_u32(e_at, pl);
                }
                k1 = solve_out_run(s_run, base, lo_th, (pp != NOT_FOUND && pp >= s_run) ? pp : NOT_FOUND, sh_out, wp, e1);
            }
        }
        if (k1 < d && e1 >= hi_th) return k1;
    }
    const uint32_t nwords_all = ((d < SEG ? d : SEG) + 31) >> 5;
    for (uint32_t w = lane; w < nwords_all; w += WAVE) mask[w] = 0;
    wave_lds_fence();

    PROF_TICK(pf, 1);
    // Rows longer than the LDS mask (SEG positions) are served through a sliding window [wb, wb + SEG):
    // the chain consumes a window completely before the mask is reused for the next one.
    float c = 0.0f;
    uint32_t k = 0, known_end = 0, cnt_in = 0, found = NOT_FOUND;
    uint32_t wb = 0, base = 0;
    bool exact_ok = true;
    while (k < d) {
        const uint32_t wend = d - wb < SEG ? d : wb + SEG;   // end of the current window
        if (base < kn && cnt_in < n_in) {   // keys left and common neighbours still missing
            const uint32_t i = base + lane;
            const bool valid = i < kn;
            uint64_t kfw = kfw_first;
            if (base != 0) kfw = valid ? krow[i] : 0ull;
            const uint32_t key = (uint32_t)kfw, fw = (uint32_t)(kfw >> 32);
            const uint64_t word = valid ? fb[filter_word(fw, nw_mask)] : 0ull;
            const bool pass = valid && filter_pass(word, fw);
            const uint32_t gpos = adj_lookup_g(tab, tmask, key, pass);
            const bool hit = pass && gpos != 0xffffffffu;
            if (scatter) {
                // keys ascend, so do the found positions
                const bool inw = hit && gpos >= wb && gpos < wend;
                const uint64_t hb = ballot(inw);
                const bool beyond = ballot(hit && gpos >= wend) != 0;
                if (inw) atomicOr(&mask[(gpos - wb) >> 5], 1u << ((gpos - wb) & 31));
                if (hb) known_end = readlane_u32(gpos, 63 - __builtin_clzll(hb)) + 1;
                cnt_in += (uint32_t)__popcll(hb);
                if (beyond) known_end = wend;             // the rest of this chunk belongs to later windows
                else {
                    base += WAVE;
                    if (base >= kn) known_end = wend;     // every neighbour of prev has been looked up
                }
            } else {
                const uint64_t hb = ballot(hit);
                if (lane < 2) mask[((base - wb) >> 5) + lane] = lane ? (uint32_t)(hb >> 32) : (uint32_t)hb;
                cnt_in += (uint32_t)__popcll(hb);
                base += WAVE;
                known_end = base < d ? base : d;
            }
        } else {
            known_end = wend;   // every remaining position is "out" (or prev itself)
        }
        PROF_TICK(pf, 2);
        PROF_COUNT(pf, 8, 1);
        if (known_end <= k) continue;
        // exact mass of the known prefix, in units; the float32 chain is within (k + 4) * 2^-24 of the exact
        // partial sums (one rounding per addition, 2^-24 relative on the three values): z units
        const uint32_t pv_k = (n_pv && prev_pos < known_end) ? 1u : 0u;
        const uint32_t est = (cnt_in << sh_in) + ((known_end - cnt_in - pv_k) << sh_out) + (pv_k << sh_prev);
        const uint32_t z = (uint32_t)(((uint64_t)(known_end + 4u) * units_i) >> 24) + 1u;
        if (known_end < wend && est + z + 2u < r_units) continue;   // target not inside the known prefix yet
        wave_lds_fence();
        build_rank(mask, rank, (known_end - wb + 31) >> 5);
        PROF_TICK(pf, 3);
        PROF_COUNT(pf, 9, 1);
        const UnitRow ur{mask, rank, wb, known_end - wb, prev_pos, true};
        if (exact_ok && k == 0 && wb == 0) {
            // Decide in exact arithmetic when no partial sum lies within the float32 drift bound of the
            // target (argument and bound: seqscan.h, exact_thresholds_f32).
            const ExactThresholds th = exact_thresholds_f32(r * units, known_end, 1u << max_u32(sh_in, max_u32(sh_out, sh_prev)));
            const uint32_t lo_th = uni(th.lo), hi_th = uni(th.hi);
            if (est >= hi_th) {
                uint32_t e_at = 0;
                const uint32_t k1 = unit_search_units(ur, known_end, lo_th, sh_in, sh_out, sh_prev, e_at);
                PROF_TICK(pf, 7);
                if (e_at >= hi_th) return k1;
                PROF_COUNT(pf, 11, 1);
            } else if (known_end < wend) {
                continue;   // the prefix may still end below the target: keep loading
            }
            exact_ok = false;   // a partial sum sits next to the target (or r is near the total): float chain
        }
        const RowVals<float, true> rv = make_unit_vals<float>(mask, wb, known_end, prev_pos, true, x_in, x_out, x_prev);
        if (k == 0) {
            const bool hit0 = seq_head<float, true>(c, k, known_end, r, rv, WAVE, found);
            PROF_TICK(pf, 4);
            if (hit0) return found;
        }
        if (k < known_end) {
            const int rc = unit_chain<float, true>(c, k, known_end, r, ur, rv, x_in, x_out, x_prev, found);
            PROF_TICK(pf, 5);
            if (rc == SCAN_FOUND) return found;
        }
        if (known_end == wend && wend < d) {   // window consumed: slide
            wb = wend;
            const uint32_t nw = ((d - wb < SEG ? d - wb : SEG) + 31) >> 5;
            for (uint32_t w = lane; w < nw; w += WAVE) mask[w] = 0;
            wave_lds_fence();
        }
    }
    return d;  // the float CDF never reached r (mirrored overflow read)
}

// ---- general (weighted) transition -------------------------------------------------------------------
// Returns the sampled neighbour *position* k in [0, d] (d == "CDF never reached r").
template <typename T, bool DENSE>
// known_tot (may be nullptr): the row's normaliser from the per-edge table -- pass 1 is skipped.  tot_out (may be
// nullptr): only the normaliser is wanted (table build) -- pass 2 is skipped and the return value is meaningless.
__device__ __forceinline__ uint32_t sample_step_weighted(const WalkArgs &a, uint32_t *mask, uint32_t *in_mask,
                                                         uint32_t *queue, uint32_t cur, bool has_prev, uint32_t prev,
                                                         uint32_t t0, uint32_t dp, double r, uint32_t s0,
                                                         uint32_t d, const T *known_tot = nullptr, T *tot_out = nullptr,
                                                         uint32_t k_start = 0, const T *c_start = nullptr, T *ckpt_out = nullptr,
                                                         uint32_t window = 0, uint32_t known_prev_pos = NOT_FOUND, bool prev_pos_known = false) {
    // k_start / c_start (round 4, weighted lane form): the CDF search starts at element k_start (a multiple of CHAIN_CKPT)
    // with the chain's exact value after element k_start - 1 -- every earlier partial sum is known to stay below r.
    // ckpt_out: no search; the normalised chain is run over the whole row and its value after every CHAIN_CKPT elements is
    // recorded (ckpt_out[m] = c after element (m + 1) * CHAIN_CKPT - 1, for (m + 1) * CHAIN_CKPT < d).
    // window (with a known normaliser): the search works on windows of that many elements instead of whole mask segments --
    // the membership mask is scattered from the arriving entry's list for the window only (a hub-to-hub entry's list has
    // thousands of entries; the scan of a parked step ends within a window or two of where it starts).
    const uint32_t *__restrict__ indices = a.g.indices;
    const T *__restrict__ data = (const T *)a.g.data;
    const bool extend = in_mask != nullptr;
    RowVals<T, false> rv;
    rv.drow = data + s0;
    rv.mask = mask;
    rv.has_prev = has_prev;
    rv.p = a.p;
    rv.q = a.q;
    rv.tot = (T)1;
    rv.prev_pos = NOT_FOUND;
    rv.extend = extend;
    rv.setup_bias();
    if (extend) {
        rv.in_mask = in_mask;
        rv.crow = indices + s0;
        rv.prow = indices + t0;
        rv.pdata = data + t0;
        rv.thr = a.g.thr;
        rv.dp = dp;
        if (!DENSE) {
            const uint64_t tb0 = readfirst_u64(a.g.tab_off[prev]);
            rv.ptmask = (uint32_t)(readfirst_u64(a.g.tab_off[prev + 1]) - tb0) - 1u;
            rv.ptab = a.g.slots + tb0;
        }
        rv.thr_cur = __uint_as_float(uni(__float_as_uint(a.g.thr[cur])));
    }

    const uint32_t seg = (window && known_tot && !ckpt_out) ? window : SEG;
    const bool multi = has_prev && (d > SEG || (seg < SEG && d > seg));
    // prev's position is needed by every segment: find it once when the row is segmented
    if (multi) {
        if (prev_pos_known) rv.prev_pos = known_prev_pos;   // (the caller has it from the arriving entry's record)
        else {
            uint32_t pos = uni(lower_bound_u32(indices + s0, d, prev));
            if (pos < d && uni(indices[s0 + pos]) == prev) rv.prev_pos = pos;
        }
    }

    // pass 1: tot
    T tot = (T)0;
    rv.normalize = false;
    if (known_tot) {
        tot = *known_tot;
        if (has_prev && !multi) rv.prev_pos = segment_mask<T, DENSE>(a.g, mask, in_mask, queue, cur, s0, 0, d, t0, dp, prev);
    } else {
        for (uint32_t sa = 0; sa < d; sa += SEG) {
            uint32_t len = d - sa < SEG ? d - sa : SEG;
            if (has_prev) {
                uint32_t pp = segment_mask<T, DENSE>(a.g, mask, in_mask, queue, cur, s0, sa, len, t0, dp, prev);
                if (!multi) rv.prev_pos = pp;
            }
            rv.seg_a = sa;
            rv.kend = sa + len;
            seq_scan<T, false>(tot, sa, sa + len, 0.0, rv, sa == 0 ? WAVE : 0);
        }
    }
    if (tot_out) { *tot_out = tot; return 0; }

    // pass 2: cdf search
    rv.normalize = true;
    rv.tot = tot;
    if (ckpt_out) {   // the whole normalised chain, recorded every CHAIN_CKPT elements (chunks never straddle a mask segment)
        T c = (T)0;
        for (uint32_t sa = 0; sa < d; sa += SEG) {
            const uint32_t len = d - sa < SEG ? d - sa : SEG;
            if (multi) (void)segment_mask<T, DENSE>(a.g, mask, in_mask, queue, cur, s0, sa, len, t0, dp, prev);
            rv.seg_a = sa;
            for (uint32_t kb = sa; kb < sa + len; kb += CHAIN_CKPT) {
                const uint32_t ke = kb + CHAIN_CKPT < sa + len ? kb + CHAIN_CKPT : sa + len;
                // (the scan works in 256-element trips that may reach past its end after a binade crossing: the VALUES beyond
                //  the chunk have to read as zero, which is what rv.kend does -- not the scan's own end)
                rv.kend = ke;
                (void)seq_scan<T, false>(c, kb, ke, 0.0, rv, kb == 0 ? WAVE : 0);
                if (ke < d && (ke % CHAIN_CKPT) == 0u && lane_id() == 0) ckpt_out[ke / CHAIN_CKPT - 1u] = c;
            }
        }
        return 0;
    }
    T c = c_start ? *c_start : (T)0;
    uint32_t choice = NOT_FOUND;
    for (uint32_t sa = (k_start / seg) * seg; sa < d && choice == NOT_FOUND; sa += seg) {
        uint32_t len = d - sa < seg ? d - sa : seg;
        if (multi) (void)segment_mask<T, DENSE>(a.g, mask, in_mask, queue, cur, s0, sa, len, t0, dp, prev);  // single segment: still valid
        rv.seg_a = sa;
        rv.kend = sa + len;
        const uint32_t kbeg = k_start > sa ? k_start : sa;
        choice = seq_scan<T, true>(c, kbeg, sa + len, r, rv, kbeg == 0 ? WAVE : 0);
    }
    return choice == NOT_FOUND ? d : choice;
}

#ifndef PW_MIN_WAVES
#define PW_MIN_WAVES 8
#endif

// T = float : SparseOTF (reference float32 path).  T = double, DENSE: DenseOTF (float64 path, membership by
// adjacency bits, "choice == degree" clamped to the last neighbour -- App. D quirk 1, dense row).
// UNIT: every edge weight is 1.0 (closed-form chain); EXTEND: node2vec+ (weighted graphs only --
// on unit weights node2vec+ degenerates to node2vec bit for bit, so the host routes it to UNIT).
template <typename T, bool DENSE, bool UNIT, bool EXTEND>
__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE, (EXTEND && !DENSE) ? PW_MIN_WAVES - 1 : PW_MIN_WAVES)
walk_kernel(WalkArgs a) {
    __shared__ uint32_t s_mask[WAVES_PER_BLOCK][MASK_WORDS];
    __shared__ uint16_t s_rank[UNIT ? WAVES_PER_BLOCK : 1][UNIT ? MASK_WORDS + 2 : 2];
    __shared__ uint32_t s_in[EXTEND ? WAVES_PER_BLOCK : 1][EXTEND ? MASK_WORDS : 1];
    __shared__ uint32_t s_queue[DENSE ? 1 : WAVES_PER_BLOCK][DENSE ? 2 : 2 * QCAP];
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    uint32_t *mask = s_mask[wave];
    uint32_t *queue = s_queue[DENSE ? 0 : wave];
    uint16_t *rank = s_rank[UNIT ? wave : 0];
    const uint32_t L = PW_KARG(uint32_t, L);
    const uint64_t W = (uint64_t)L + 2;
    // per-wave statistics live in LDS: [0] steps [1] overflow reads [2] clamped reads [3] dead-end walks
    __shared__ unsigned long long s_stat[WAVES_PER_BLOCK][4];
    unsigned long long *stat = s_stat[wave];
    if (lane < 4) stat[lane] = 0;
    Prof pf;
#ifdef PW_PROF
    __shared__ unsigned long long s_prof[WAVES_PER_BLOCK][16];
    pf.acc = s_prof[wave];
    if (lane < 16) pf.acc[lane] = 0;
    pf.last = __builtin_readcyclecounter();
#endif
    wave_lds_fence();

    // Wave-uniform reads of the read-only arrays go through the scalar unit (wave.h); the pointers are
    // re-read from the kernarg segment where they are used instead of being kept live.
    for (;;) {
        unsigned long long widx = 0;
        if (lane == 0) widx = atomicAdd((unsigned long long *)PW_KARG(uint64_t, job_counter), 1ull);
        widx = readfirst_u64(widx);
        const uint64_t p_list = PW_KARG(uint64_t, job_list);
        if (widx >= (p_list ? PW_KARG(uint64_t, n_list) : PW_KARG(uint64_t, n_jobs))) break;
        const uint64_t job = p_list ? (uint64_t)as_scalar<uint32_t>(p_list)[widx] : (uint64_t)widx;
        const uint32_t start = as_scalar<uint32_t>(PW_KARG(uint64_t, starts))[job];
        const uint64_t soff = as_scalar<uint64_t>(PW_KARG(uint64_t, stream_off))[job] - PW_KARG(uint64_t, rng_base);

        uint32_t cur = start, prev = 0;
        // vertex contexts of cur and prev (= cur one step earlier): sparse graphs read the 16-byte vertex
        // record, dense (compressed-row) graphs only have indptr
        VertexCtx vc{0, 0, 0, 0}, vp{0, 0, 0, 0};
        auto enter = [&](uint32_t v) {
            if (!DENSE) {
                const u32x4 rec = as_scalar<u32x4>(PW_KARG(uint64_t, g.vrec))[v];   // one s_load_dwordx4
                vc = VertexCtx{rec.x, rec.y, rec.z, rec.w};
            } else {
                const sptr<uint32_t> indptr = as_scalar<uint32_t>(PW_KARG(uint64_t, g.indptr));
                const uint32_t b = indptr[v];
                vc = VertexCtx{b, indptr[v + 1] - b, 0, 0};
            }
        };
        enter(cur);
        bool lazy_next = false;    // the edge prev -> cur is a real CSR entry with a common-neighbour count
        uint32_t n_in = 0;         // tri[e(prev -> cur)], requested together with the sampled neighbour
        uint32_t rev_pos = NOT_FOUND;   // position of prev in cur's row (same record)
        bool keys_pre = false;     // the first 64 keys of the next step were requested with that record
        uint64_t kfw_pre = 0;
        uint32_t len_out = L + 1;
        uint32_t prev_edge = NOT_FOUND;   // CSR entry the walker arrived by (NOT_FOUND: first step / mirrored overflow read)
        uint32_t j = 1;
        if (!DENSE && PW_KARG(uint32_t, resume) != 0) {
            const sptr<uint32_t> row_in = as_scalar<uint32_t>(PW_KARG(uint64_t, out));
            const uint32_t j0 = row_in[job * W + L + 1];
            if (j0 >= 2 && j0 <= L) {   // (the edge prev -> cur is not looked up: the first step here is an eager one)
                prev = row_in[job * W + j0 - 2];
                cur = row_in[job * W + j0 - 1];
                enter(prev);
                vp = vc;
                enter(cur);
                j = j0;
            }
        }
        const uint32_t j_first = j;
        for (; j <= L; j++) {
            const uint32_t s0 = vc.s0, d = vc.d, t0 = vp.s0, dp = vp.d;
            if (d == 0) {
                len_out = j;
                if (j > 1 && lane == 0) stat[3]++;
                break;
            }
            const double r = as_scalar<double>(PW_KARG(uint64_t, rng))[soff + (j - 1)];
            uint32_t choice;
            if (UNIT) {
                choice = LAZY_FALLBACK;
#ifndef PW_NO_LAZY
                PROF_TICK(pf, 0);
                if (!DENSE && lazy_next) choice = sample_step_unit_lazy(mask, rank, cur, prev, n_in, rev_pos, keys_pre, kfw_pre, vc, vp, r, pf);
#endif
                if (choice == LAZY_FALLBACK) {
                    PROF_TICK(pf, 1);
                    WalkArgs la = reload_walk_args();
                    la.g.step_edge = j >= 2 ? prev_edge : NOT_FOUND;
                    choice = sample_step_unit<T, DENSE>(la, mask, rank, queue, cur, j >= 2, prev, t0, dp, r, s0, d);
                    PROF_TICK(pf, 6);
                    PROF_COUNT(pf, 10, 1);
                }
            }
            else {
                WalkArgs la = reload_walk_args();   // arguments are not kept live across the loop
                la.g.step_edge = j >= 2 ? prev_edge : NOT_FOUND;
                // normaliser of this transition from the per-edge table (float32 CSR graphs), when the walker
                // arrived by a real CSR entry
                T ktot = (T)0;
                bool have_tot = false;
                if (!DENSE && la.tot_e) {
                    if (j == 1) { ktot = (T)as_scalar<float>((uint64_t)la.tot_v)[cur]; have_tot = true; }
                    else if (prev_edge != NOT_FOUND) { ktot = (T)as_scalar<float>((uint64_t)la.tot_e)[prev_edge]; have_tot = true; }
                }
                choice = sample_step_weighted<T, DENSE>(la, mask, EXTEND ? s_in[EXTEND ? wave : 0] : nullptr, queue, cur,
                                                        j >= 2, prev, t0, dp, r, s0, d, have_tot ? &ktot : nullptr);
            }
            choice = uni(choice);
            bool clamped = false;
            const bool real_edge = choice < d;
            if (!real_edge) {
                if (lane == 0) stat[1]++;
                if (DENSE) { choice = d - 1; clamped = true; }  // reference reads past a temporary: clamp
            }
            uint64_t pos = (uint64_t)s0 + choice;
            const uint32_t nnz = PW_KARG(uint32_t, g.nnz);
            if (pos >= nnz) { pos = nnz - 1; clamped = true; }
            if (clamped && lane == 0) stat[2]++;
            prev_edge = (real_edge && !clamped) ? (uint32_t)pos : NOT_FOUND;
            // The edge just taken: one 16-byte record names the next vertex, its degree, the number of
            // common neighbours and where cur sits in the next vertex's row.  When cur's row is the
            // shorter one it supplies the keys of the next step: request the first 64 right away, together
            // with the next vertex's record.
            uint32_t nxt;
            lazy_next = false;
            keys_pre = false;
            kfw_pre = 0;
            const uint64_t p_tri = (UNIT && !DENSE) ? PW_KARG(uint64_t, g.tri) : 0ull;
            if (p_tri != 0) {
                const u32x4 er = as_scalar<u32x4>(p_tri)[pos * 4u];   // first 16 bytes of the entry's 64-byte edge line
                nxt = er.x;
                n_in = er.y;
                rev_pos = er.z;
                lazy_next = real_edge && PW_KARG(uint32_t, lazy_ok) != 0;
                if (lazy_next && n_in && d <= er.w) {
                    keys_pre = true;
                    if ((uint32_t)lane < d) kfw_pre = (as_global<uint64_t>(PW_KARG(uint64_t, g.kf)) + s0)[lane];
                }
            } else {
                nxt = as_scalar<uint32_t>(PW_KARG(uint64_t, g.indices))[pos];
            }
            if (lane == 0) ((gptr_mut<uint32_t>)PW_KARG(uint64_t, out))[job * W + j] = nxt;
            prev = cur;
            vp = vc;
            cur = nxt;
            enter(cur);
        }
        // header, tail zeros and length cell (cells j..L stay 0 after an early stop)
        gptr_mut<uint32_t> row = (gptr_mut<uint32_t>)PW_KARG(uint64_t, out) + job * W;
        if (lane == 0) {
            row[0] = start;
            row[L + 1] = len_out;
            stat[0] += j - j_first;   // transitions sampled here
        }
        for (uint32_t z = j + lane; z <= L; z += WAVE) row[z] = 0;
    }
    wave_lds_fence();
#ifdef PW_PROF
    PROF_TICK(pf, 0);
    if (lane < 16) atomicAdd(&g_prof[lane], pf.acc[lane]);
#endif
    if (lane < 4 && stat[lane]) atomicAdd((unsigned long long *)PW_KARG(uint64_t, stats) + lane, stat[lane]);
}

// Per-edge normalisers of a weighted CSR graph for one (p, q, extend): work item e < nnz = CSR entry (u -> v):
// tot_e[e] = pass 1 of sample_step_weighted for cur = v, prev = u; item nnz + v: tot_v[v] = the unbiased row sum of v.
// One wavefront per item, one item per wavefront (grid = items / WAVES_PER_BLOCK: no loop around the step code, so
// nothing but the item index is live across it -- a persistent-loop form of this kernel hung whenever its register
// allocation changed); bit-identical to what the walk step would compute, because it IS the walk step's code.
// Cost: sum over edges of the row length of the head = sum of squared degrees -- one pass of ~E[d_visit] elements per
// CSR entry, against E[d_visit] elements TWICE per sampled step without the table.
template <bool EXTEND>
__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE, EXTEND ? PW_MIN_WAVES - 1 : PW_MIN_WAVES)
tot_build_kernel(WalkArgs a_unused, const uint32_t *__restrict__ edge_row_unused, float *tot_e_unused, float *tot_v_unused) {
    // (like walk_kernel, every argument is re-read from the kernarg segment at its point of use)
    __shared__ uint32_t s_mask[WAVES_PER_BLOCK][MASK_WORDS];
    __shared__ uint32_t s_in[EXTEND ? WAVES_PER_BLOCK : 1][EXTEND ? MASK_WORDS : 1];
    __shared__ uint32_t s_queue[WAVES_PER_BLOCK][2 * QCAP];
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    constexpr size_t XARG = (sizeof(WalkArgs) + 7) & ~(size_t)7;   // edge_row, tot_e, tot_v follow the struct
    const uint64_t item = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (uint64_t)wave;
    const uint32_t nnz = PW_KARG(uint32_t, g.nnz);
    if (item >= (uint64_t)nnz + PW_KARG(uint32_t, g.n_nodes)) return;
    const bool is_edge = item < nnz;
    const sptr<uint32_t> indptr = as_scalar<uint32_t>(PW_KARG(uint64_t, g.indptr));
    uint32_t cur = (uint32_t)(item - nnz), prev = 0;
    if (is_edge) {
        cur = as_scalar<uint32_t>(PW_KARG(uint64_t, g.indices))[item];
        prev = as_scalar<uint32_t>(kernarg<uint64_t>(XARG))[item];
    }
    const uint32_t s0 = indptr[cur], d = indptr[cur + 1] - s0;
    const uint32_t t0 = indptr[prev], dp = indptr[prev + 1] - t0;
    float tot = 0.0f;
    if (d) {
        WalkArgs la = reload_walk_args();
        la.g.step_edge = is_edge ? (uint32_t)item : NOT_FOUND;
        (void)sample_step_weighted<float, false>(la, s_mask[wave], EXTEND ? s_in[EXTEND ? wave : 0] : nullptr, s_queue[wave], cur,
                                                 is_edge, prev, t0, dp, 0.0, s0, d, nullptr, &tot);
    }
    if (lane == 0) {
        if (is_edge) ((gptr_mut<float>)kernarg<uint64_t>(XARG + 8))[item] = tot;
        else ((gptr_mut<float>)kernarg<uint64_t>(XARG + 16))[item - nnz] = tot;
    }
}

// ---- single-step probe: the reference's move_forward / get_normalized_probs for ONE (cur, prev) ----------------------
// Backs pw_step / pw_probs (include/pecanpy_amd.h): Base.get_move_forward() of the drop-in API and the bitwise
// comparison of the HIP path's probability vectors with the reference-generated fixtures.  One wavefront; the
// sampling is the walk kernels' own eager step, the probabilities are the values that step sums: w_k / tot with the
// same membership mask, the same biased weights and the same sequential float sum.
struct ProbeArgs {
    uint32_t cur, has_prev, prev, want_probs;
    double r;
    void *probs;        // T[degree(cur)] (float32 CSR graphs, float64 dense graphs), or nullptr
    uint32_t *out;      // [0] sampled position (== degree: the CDF never reached r) [1] next vertex [2] degree(cur)
};

template <typename T, bool DENSE, bool UNIT, bool EXTEND>
__global__ void __launch_bounds__(WAVE)
step_probe_kernel(WalkArgs a_unused, const ProbeArgs *pa_unused) {
    __shared__ uint32_t s_mask[MASK_WORDS];
    __shared__ uint16_t s_rank[MASK_WORDS + 2];
    __shared__ uint32_t s_in[EXTEND ? MASK_WORDS : 1];
    __shared__ uint32_t s_queue[2 * QCAP];
    const int lane = lane_id();
    constexpr size_t XARG = (sizeof(WalkArgs) + 7) & ~(size_t)7;
    const ProbeArgs pa = *(const ProbeArgs *)kernarg<uint64_t>(XARG);
    WalkArgs la = reload_walk_args();
    const uint32_t cur = uni(pa.cur), prev = uni(pa.prev);
    const bool has_prev = uni(pa.has_prev) != 0u;
    const uint32_t s0 = uni(la.g.indptr[cur]), d = uni(la.g.indptr[cur + 1]) - s0;
    const uint32_t t0 = has_prev ? uni(la.g.indptr[prev]) : 0u, dp = has_prev ? uni(la.g.indptr[prev + 1]) - t0 : 0u;
    if (!DENSE && has_prev && dp) {   // the CSR entry (prev -> cur), when there is one: its list gives the membership mask
        const uint32_t pe = uni(lower_bound_u32(la.g.indices + t0, dp, cur));
        if (pe < dp && uni(la.g.indices[t0 + pe]) == cur) la.g.step_edge = t0 + pe;
    }
    if (lane == 0) pa.out[2] = d;
    if (d == 0) return;
    uint32_t choice;
    if (UNIT) choice = sample_step_unit<T, DENSE>(la, s_mask, s_rank, s_queue, cur, has_prev, prev, t0, dp, pa.r, s0, d);
    else choice = sample_step_weighted<T, DENSE>(la, s_mask, EXTEND ? s_in : nullptr, s_queue, cur, has_prev, prev, t0, dp, pa.r, s0, d);
    choice = uni(choice);
    if (lane == 0) {
        uint32_t c = choice;
        if (DENSE && c >= d) c = d - 1;   // dense rows: the reference reads past a temporary, clamped (walk kernel)
        uint64_t pos = (uint64_t)s0 + c;
        if (pos >= la.g.nnz) pos = la.g.nnz - 1;
        pa.out[0] = choice;
        pa.out[1] = la.g.indices[pos];
    }
    if (!pa.want_probs) return;
    T *probs = (T *)pa.probs;
    const T *__restrict__ data = (const T *)la.g.data;
    // the step's value view (identical set-up to sample_step_unit / sample_step_weighted)
    RowVals<T, UNIT> rv;
    rv.drow = UNIT ? nullptr : data + s0;
    rv.mask = s_mask;
    rv.has_prev = has_prev;
    rv.p = la.p;
    rv.q = la.q;
    rv.tot = (T)1;
    rv.prev_pos = NOT_FOUND;
    rv.normalize = false;
    rv.extend = EXTEND;
    rv.setup_bias();
    if (UNIT) {
        rv.u_in = (T)1;
        rv.u_out = has_prev ? uni(Arith<T>::bias_div((T)1, la.q)) : (T)1;
        rv.u_prev = uni(Arith<T>::bias_div((T)1, la.p));
    }
    if (EXTEND) {
        rv.in_mask = s_in;
        rv.crow = la.g.indices + s0;
        rv.prow = la.g.indices + t0;
        rv.pdata = data + t0;
        rv.thr = la.g.thr;
        rv.dp = dp;
        if (!DENSE && has_prev) {
            const uint64_t tb0 = readfirst_u64(la.g.tab_off[prev]);
            rv.ptmask = (uint32_t)(readfirst_u64(la.g.tab_off[prev + 1]) - tb0) - 1u;
            rv.ptab = la.g.slots + tb0;
        }
        rv.thr_cur = __uint_as_float(uni(__float_as_uint(la.g.thr[cur])));
    }
    const bool multi = has_prev && d > SEG;
    if (multi) {
        const uint32_t pos = uni(lower_bound_u32(la.g.indices + s0, d, prev));
        if (pos < d && uni(la.g.indices[s0 + pos]) == prev) rv.prev_pos = pos;
    }
    T tot = (T)0;
    for (int pass = 0; pass < 2; pass++) {   // pass 0: tot, pass 1: normalised values
        for (uint32_t sa = 0; sa < d; sa += SEG) {
            const uint32_t len = d - sa < SEG ? d - sa : SEG;
            if (has_prev && (multi || pass == 0)) {
                const uint32_t pp = segment_mask<T, DENSE>(la.g, s_mask, EXTEND ? s_in : nullptr, s_queue, cur, s0, sa, len, t0, dp, prev);
                if (!multi) rv.prev_pos = pp;
                if (UNIT && rv.prev_pos != NOT_FOUND && rv.prev_pos >= sa && rv.prev_pos < sa + len) {
                    const uint32_t rr = rv.prev_pos - sa;   // unit rows keep prev's own bit clear (three disjoint classes)
                    if (lane == 0) s_mask[rr >> 5] &= ~(1u << (rr & 31));
                    wave_lds_fence();
                }
            }
            rv.seg_a = sa;
            rv.kend = sa + len;
            if (pass == 0) seq_scan<T, false>(tot, sa, sa + len, 0.0, rv, sa == 0 ? WAVE : 0);
            else
                for (uint32_t kb = sa; kb < sa + len; kb += WAVE) {   // (value_ext votes across the wave: uniform trip count)
                    const uint32_t k = kb + (uint32_t)lane;
                    const bool valid = k < sa + len;
                    T v = (T)0;
                    if (!UNIT && EXTEND) v = rv.value_ext(k, valid);
                    else if (valid) v = rv.one(k);
                    if (valid) probs[k] = v;
                }
        }
        if (pass == 0) {
            tot = uni(tot);
            if (UNIT) { rv.u_in = rv.u_in / tot; rv.u_out = rv.u_out / tot; rv.u_prev = rv.u_prev / tot; }
            else { rv.normalize = true; rv.tot = tot; }
        }
    }
}

}  // namespace pw
