// walk_lanes.hip.h -- SparseOTF walk kernel, ONE LANE PER WALK (gfx950), for the headline regime: unit weights,
// 1/p and 1/q powers of two, no self loops (the regime of walk_sparse.hip.h's lazy step).
//
// Why a second kernel.  walk_kernel (walk_sparse.hip.h) gives every walk a whole wavefront and spends ~1.4 k
// wave instructions per step establishing which neighbours of `cur` are neighbours of `prev` (keys -> Bloom word
// -> index slot) before the exact-arithmetic CDF search can run; it is instruction-issue bound (round-1 PMC:
// scalar unit 72 %, VALU 62 %, HBM 33 %).  With 288 GB of HBM the membership question can be answered ONCE per
// graph instead of once per step:
//
//   clist : for every CSR entry e = (u -> v) the ascending positions, in row v, of the common neighbours of u and
//           v (sum over edges of the per-edge triangle count: 2.7 G entries = 11 GB at RMAT-22).
//   erec  : 32-byte record per CSR entry e = (u -> v): { v, |N(u) & N(v)|, position of u in row v, degree(v),
//           indptr[v], offset of e's list in clist } -- everything a step needs about the edge it arrives by.
//
// A step of the reference (SparseOTF.move_forward, src/pecanpy/pecanpy.py:543-559; get_normalized_probs,
// src/pecanpy/rw/sparse_rw.py:51-91) then is: the exact-arithmetic decision of seqscan.h (E(k) = exact mass of
// elements 0..k in units of the smallest weight; first k with E(k) >= ceil(R - z) and E(k) >= ceil(R + z))
// evaluated by ONE LANE: a binary search over the edge's common-neighbour positions (each P_i splits the row into
// runs of "out" neighbours whose mass is a closed form), no membership work at all.  64 walks advance per
// wavefront instruction; a step costs one 32-byte record, ~log2(n_common) list probes, one draw and one store.
//
// Steps the exact decision cannot settle (a partial sum of the exact CDF lies within the float32 drift bound of
// the target, ~8 % of the steps) are resolved by the whole wave for one lane at a time with the bit-exact float32
// chain of walk_sparse.hip.h (seq_head + unit_chain) over an LDS mask scattered from the same list.  The rare
// rest -- the mirrored overflow read (choice == degree, App. D quirk 1), rows whose total is not exact in
// float32 -- is not handled here: the job is appended to a redo list and walked again by walk_kernel.
#pragma once
#include "walk_sparse.hip.h"

namespace pw {

struct ERec {
    uint32_t nxt;       // v
    uint32_t n_in;      // |N(u) & N(v)|
    uint32_t rev_pos;   // position of u in row v, NOT_FOUND when (v -> u) is not an edge
    uint32_t deg;       // degree(v)
    uint32_t s0;        // indptr[v]
    uint32_t coff_lo;   // clist offset of this edge's list (64 bit)
    uint32_t coff_hi;
    uint32_t pad;
};
static_assert(sizeof(ERec) == 32, "edge record is two 16-byte loads");

struct LanesArgs {
    const ERec *__restrict__ erec;
    const uint32_t *__restrict__ clist;
    const uint4 *__restrict__ vrec;           // { indptr[v], degree(v), .. } (walk_sparse.hip.h)
    uint32_t nnz;
    uint32_t L;
    uint64_t n_jobs;
    const uint32_t *__restrict__ starts;
    const uint64_t *__restrict__ stream_off;
    const uint32_t *__restrict__ job_list;    // optional: run only these jobs (repair passes)
    uint64_t n_list;
    const double *__restrict__ rng;
    uint64_t rng_base;
    uint32_t *out;                            // [n_jobs, L + 2], zero-filled by the caller before the first launch
    unsigned long long *job_counter;
    unsigned long long *stats;                // [0] steps [1] overflow reads [2] clamped reads [3] dead-end walks
                                              // [6] list entries read [7] ambiguous steps (float chain)
    uint32_t *redo_list;                      // jobs handed to walk_kernel
    unsigned long long *redo_count;
    float w_out, w_prev;                      // fl32(1/q), fl32(1/p): powers of two (host checked)
};

// (per-lane exact decision: lane_decide / LaneStep in seqscan.h, shared with the host self test)

// ---- wave-cooperative float32 chain for one lane's step ---------------------------------------------------------
// Same arithmetic as sample_step_unit_lazy's fallback (walk_sparse.hip.h): mask of the common neighbours among
// the first kmax positions (scattered from the edge's list), rank array, 64-element head, closed-form binade
// chain.  All arguments wave-uniform.  Returns the position, or d when the float CDF never reaches r.
__device__ __forceinline__ uint32_t wave_chain_step(uint32_t *mask, uint16_t *rank, uint32_t d, uint32_t kmax, uint32_t n_in,
                                                    uint32_t pp, const uint32_t *__restrict__ cl, double r, float tot,
                                                    float w_out, float w_prev) {
    const int lane = lane_id();
    const float x_in = uni(1.0f / tot), x_out = uni(x_in * w_out), x_prev = uni(x_in * w_prev);
    float c = 0.0f;
    uint32_t k = 0, found = NOT_FOUND;
    for (uint32_t wb = 0; wb < kmax; wb += SEG) {
        const uint32_t wend = kmax - wb < SEG ? kmax : wb + SEG;
        const uint32_t nw = (wend - wb + 31) >> 5;
        for (uint32_t w = lane; w < nw; w += WAVE) mask[w] = 0;
        wave_lds_fence();
        // the list ascends: stop at the first chunk that starts beyond the window
        for (uint32_t i0 = 0; i0 < n_in; i0 += WAVE) {
            const uint32_t i = i0 + lane;
            const uint32_t P = i < n_in ? cl[i] : NOT_FOUND;
            if (P >= wb && P < wend) atomicOr(&mask[(P - wb) >> 5], 1u << ((P - wb) & 31));
            if (readlane_u32(P, 0) >= wend) break;
        }
        wave_lds_fence();
        build_rank(mask, rank, nw);
        const UnitRow ur{mask, rank, wb, wend - wb, pp, true};
        const RowVals<float, true> rv = make_unit_vals<float>(mask, wb, wend, pp, true, x_in, x_out, x_prev);
        if (k == 0 && seq_head<float, true>(c, k, wend, r, rv, WAVE, found)) return found;
        if (k < wend && unit_chain<float, true>(c, k, wend, r, ur, rv, x_in, x_out, x_prev, found) == SCAN_FOUND) return found;
    }
    // kmax < d: by construction of kmax the chain has reached r before; reaching this point with kmax < d would
    // contradict the drift bound -- report "never reached" and let the caller hand the walk to walk_kernel
    return d;
}

#ifndef PW_LANES_MIN_WAVES
#define PW_LANES_MIN_WAVES 8
#endif

__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE, PW_LANES_MIN_WAVES)
walk_lanes_kernel(LanesArgs a) {
    __shared__ uint32_t s_mask[WAVES_PER_BLOCK][MASK_WORDS];
    __shared__ uint16_t s_rank[WAVES_PER_BLOCK][MASK_WORDS + 2];
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    uint32_t *mask = s_mask[wave];
    uint16_t *rank = s_rank[wave];
    const uint32_t L = a.L;
    const uint64_t W = (uint64_t)L + 2;
    const uint64_t n_work = a.job_list ? a.n_list : a.n_jobs;
    const float w_out = a.w_out, w_prev = a.w_prev;
    const uint64_t lane_lt = (1ull << lane) - 1ull;

    // per-lane walk state
    bool active = false, exhausted = false;
    uint32_t job = 0, j = 1;            // j = index of the step being sampled (1..L)
    uint64_t soff = 0;
    uint32_t s0 = 0, d = 0, n_in = 0, pp = NOT_FOUND;
    uint64_t coff = 0;
    unsigned long long n_steps = 0, n_dead = 0, n_probes = 0, n_amb = 0;

    for (;;) {
        // ---- refill idle lanes from the job counter -------------------------------------------------------
        for (;;) {
            const uint64_t need = ballot(!active && !exhausted);
            if (!need) break;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(a.job_counter, (unsigned long long)__popcll(need));
            base = readfirst_u64(base);
            if (!active && !exhausted) {
                const uint64_t widx = base + (uint64_t)__popcll(need & lane_lt);
                if (widx >= n_work) exhausted = true;
                else {
                    job = a.job_list ? a.job_list[widx] : (uint32_t)widx;
                    const uint32_t start = a.starts[job];
                    const uint4 vr = a.vrec[start];
                    uint32_t *row = a.out + (uint64_t)job * W;
                    row[0] = start;
                    if (vr.y == 0) {
                        row[L + 1] = 1;          // start without neighbours (pecanpy.py:190-193); cells 1..L stay 0
                        for (uint32_t z = 1; z <= L; z++) row[z] = 0;   // (a repaired row may hold an older walk)
                    } else {
                        soff = a.stream_off[job] - a.rng_base;
                        s0 = vr.x; d = vr.y; n_in = 0; pp = NOT_FOUND; coff = 0; j = 1;
                        active = true;
                    }
                }
            }
        }
        if (!ballot(active)) break;

        // ---- one step for every active lane ---------------------------------------------------------------------
        uint32_t choice = 0;
        LaneStep ls{1.0f, 0u, 0u};
        double r = 0.0;
        const float wo = j >= 2 ? w_out : 1.0f;   // first step of a walk: no bias (sparse_rw.py:66)
        if (active) {
            r = a.rng[soff + (j - 1)];
            choice = lane_decide(d, n_in, pp, r, wo, w_prev, a.clist + coff, ls);
            n_probes += ls.probes;
            if (choice == LANE_AMBIGUOUS) { n_amb++; n_probes += n_in < ls.kmax ? n_in : ls.kmax; }
        }
        // ambiguous steps: the whole wave runs the float32 chain for one lane at a time
        uint64_t amb = ballot(active && choice == LANE_AMBIGUOUS);
        while (amb) {
            const int l = __builtin_ctzll(amb);
            amb &= amb - 1ull;
            const uint32_t c_d = readlane_u32(d, l), c_kmax = readlane_u32(ls.kmax, l), c_nin = readlane_u32(n_in, l),
                           c_pp = readlane_u32(pp, l);
            const uint64_t c_coff = readlane_u64(coff, l);
            const double c_r = readlane_f64(r, l);
            const float c_tot = readlane_f32(ls.tot, l), c_wo = readlane_f32(wo, l);
            const uint32_t res = wave_chain_step(mask, rank, c_d, c_kmax, c_nin, c_pp, a.clist + c_coff, c_r, c_tot, c_wo, w_prev);
            if (lane == l) choice = res;
        }
        if (active) {
            if (choice >= d) {
                // overflow read / failed precondition: walk_kernel redoes this job from its start
                const unsigned long long slot = atomicAdd(a.redo_count, 1ull);
                a.redo_list[slot] = job;
                n_steps -= (j - 1);
                active = false;
            } else {
                const uint64_t pos = (uint64_t)s0 + choice;
                const uint4 *rp = (const uint4 *)(a.erec + pos);
                const uint4 r0 = rp[0], r1 = rp[1];
                a.out[(uint64_t)job * W + j] = r0.x;
                n_in = r0.y; pp = r0.z; d = r0.w;
                s0 = r1.x; coff = ((uint64_t)r1.z << 32) | r1.y;
                n_steps++;
                j++;
                if (j > L || d == 0) {
                    uint32_t *row = a.out + (uint64_t)job * W;
                    row[L + 1] = j;              // effective length (pecanpy.py:196-206)
                    if (j <= L) {                // dead end: the remaining cells are 0
                        n_dead++;
                        for (uint32_t z = j; z <= L; z++) row[z] = 0;
                    }
                    active = false;
                }
            }
        }
    }
    // wave totals
    for (int off = 32; off > 0; off >>= 1) {
        n_steps += (unsigned long long)__shfl_down((long long)n_steps, (unsigned)off, WAVE);
        n_dead += (unsigned long long)__shfl_down((long long)n_dead, (unsigned)off, WAVE);
        n_probes += (unsigned long long)__shfl_down((long long)n_probes, (unsigned)off, WAVE);
        n_amb += (unsigned long long)__shfl_down((long long)n_amb, (unsigned)off, WAVE);
    }
    if (lane == 0) {
        if (n_steps) atomicAdd(a.stats + 0, n_steps);
        if (n_dead) atomicAdd(a.stats + 3, n_dead);
        if (n_probes) atomicAdd(a.stats + 6, n_probes);
        if (n_amb) atomicAdd(a.stats + 7, n_amb);
    }
}

// ---- index build ------------------------------------------------------------------------------------------------
// clist / erec from the per-edge records of tri_build_kernel (tri[e] = {v, count, reverse position, degree(v)}).
constexpr int CL_BLOCK = 256;
constexpr int CL_ITEMS = 16;
constexpr int CL_TILE = CL_BLOCK * CL_ITEMS;

__global__ void __launch_bounds__(CL_BLOCK)
clist_tile_sums_kernel(const uint4 *__restrict__ tri, uint32_t nnz, uint64_t *tile_sums) {
    __shared__ uint64_t sh[CL_BLOCK];
    const uint64_t base = (uint64_t)blockIdx.x * CL_TILE + (uint64_t)threadIdx.x * CL_ITEMS;
    uint64_t s = 0;
    for (int k = 0; k < CL_ITEMS; k++)
        if (base + k < nnz) s += tri[base + k].y;
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = CL_BLOCK / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = sh[0];
}

// coff[e] = exclusive prefix sum of the per-edge counts (tile_sums already scanned: scan_tile_sums_kernel)
__global__ void __launch_bounds__(CL_BLOCK)
clist_offsets_kernel(const uint4 *__restrict__ tri, uint32_t nnz, const uint64_t *__restrict__ tile_sums, uint64_t *coff) {
    __shared__ uint64_t sh[CL_BLOCK];
    const int t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * CL_TILE + (uint64_t)t * CL_ITEMS;
    uint32_t loc[CL_ITEMS];
    uint64_t s = 0;
    for (int k = 0; k < CL_ITEMS; k++) {
        loc[k] = base + k < nnz ? tri[base + k].y : 0u;
        s += loc[k];
    }
    sh[t] = s;
    __syncthreads();
    for (int off = 1; off < CL_BLOCK; off <<= 1) {
        const uint64_t add = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    uint64_t run = tile_sums[blockIdx.x] + sh[t] - s;
    for (int k = 0; k < CL_ITEMS; k++) {
        if (base + k < nnz) coff[base + k] = run;
        run += loc[k];
    }
}

// One lane per CSR entry e = (u -> v): walks the shorter of the two rows through the longer row's filter + index
// (as tri_build_kernel did for the count) and writes the position IN ROW v of every common neighbour, ascending;
// then the edge record.
__global__ void __launch_bounds__(256)
clist_fill_kernel(CsrDev g, const uint32_t *__restrict__ edge_row, const uint64_t *__restrict__ coff, uint32_t *clist,
                  ERec *erec) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.nnz) return;
    const uint4 t = g.tri[e];
    const uint32_t u = edge_row[e], v = t.x;
    const uint32_t su = g.indptr[u], du = g.indptr[u + 1] - su;
    const uint32_t sv = g.indptr[v], dv = t.w;
    const uint64_t c0 = coff[e];
    if (t.y) {
        const bool u_short = du <= dv;
        const uint32_t ks = u_short ? su : sv, kn = u_short ? du : dv;   // keys: the shorter row
        const uint32_t w = u_short ? v : u;                               // searched vertex
        const uint32_t f0 = g.foff[w], nw_mask = g.foff[w + 1] - f0 - 1u;
        const uint64_t tb0 = g.tab_off[w];
        const uint32_t tmask = (uint32_t)(g.tab_off[w + 1] - tb0) - 1u;
        uint32_t cnt = 0;
        for (uint32_t i = 0; i < kn && cnt < t.y; i++) {
            const uint2 kfw = g.kf[ks + i];
            const uint64_t word = g.fbits[f0 + filter_word(kfw.y, nw_mask)];
            if (!filter_pass(word, kfw.y)) continue;
            const uint32_t gpos = adj_lookup(g.slots + tb0, tmask, kfw.x, true);
            if (gpos == 0xffffffffu) continue;
            clist[c0 + cnt] = u_short ? gpos : i;   // keys from row u: position found in row v; keys from row v: i
            cnt++;
        }
    }
    ERec r;
    r.nxt = v;
    r.n_in = t.y;
    r.rev_pos = t.z;
    r.deg = dv;
    r.s0 = sv;
    r.coff_lo = (uint32_t)c0;
    r.coff_hi = (uint32_t)(c0 >> 32);
    r.pad = 0;
    erec[e] = r;
}

}  // namespace pw
