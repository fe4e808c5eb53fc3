// walk_bsp.hip.h -- step-synchronous form of the lane kernel (gfx950): one launch per walk step over ALL walks.
//
// walk_lanes_kernel (walk_lanes.hip.h) keeps a walk in a lane for its whole life.  Measured it sits at
// two thirds of the GPU's random-sector rate with 8.5 scattered 64-byte sectors per step -- the draw, the output cell
// and the walk state of a lane all live in per-walk rows, so every access of a step is a sector of its own -- and
// its waves park on dependent loads; the 12 % of the steps that need the refined decision or the float32 chain hold
// up the other lanes of their wave.  Here the same per-thread routines (seqscan.h: lane_decide, lane_refine,
// lane_chain) run one step at a time:
//
//   layout   everything indexed by walk is STEP-MAJOR: draws rngT[L][n_act] (transposed once from the stream), output
//            outT[L][n_act] (transposed once into the reference's row-major matrix at the end), walk state = ONE
//            32-bit word e_cur[n_act] (the CSR entry the walk arrived by).  A step reads / writes them coalesced; the
//            only scattered accesses left are the 32-byte edge record and the list probes.
//   step j   bsp_step_kernel: one thread per walk: edge record -> previous vertex out, draw, lane_decide; decided
//            walks store their next edge, the rest append a 64-byte record to the ambiguous queue;
//            bsp_refine_kernel: one thread per queue entry (every lane busy): lane_refine; what it leaves open goes
//            to the chain queue; bsp_chain_kernel: lane_chain.  Queue lengths stay on the device (grid-stride loops),
//            so the L x 3 launches of a pass need no host synchronisation.
//   ends     walks that dead-end record their length and drop out; walks none of the routines can finish (mirrored
//            overflow read, rows outside the exact range) are marked and redone by walk_kernel, as before.
#pragma once
#include "walk_lanes.hip.h"

namespace pw {

constexpr uint32_t BSP_DEAD = 0xffffffffu;      // e_cur: the walk has ended (dead end) or was handed to the redo list
constexpr uint32_t BSP_LEN_REDO = 0xffffffffu;  // len: the row is rewritten by the redo pass

struct AmbRec {   // one ambiguous step (64 bytes)
    uint32_t w, d, n_in, pp;
    uint64_t coff;
    uint32_t s0, kmax;
    uint32_t k1, f, shifts, j;
    float tot, wo;
    double r;
};
static_assert(sizeof(AmbRec) == 64, "queue record");

struct BspArgs {
    const ERec *__restrict__ erec;
    const uint32_t *__restrict__ clist;
    const uint4 *__restrict__ vrec;
    const uint32_t *__restrict__ starts;      // by job
    const uint64_t *__restrict__ stream_off;  // by job
    const double *__restrict__ rng;
    uint64_t rng_base;
    uint32_t *out;                            // [n_jobs, L + 2], zero filled
    uint64_t n_jobs;
    uint32_t L;
    float w_out, w_prev;
    uint32_t *act_job;                        // [n_act] job of walk slot w
    unsigned long long *n_act;                // device counter (compaction)
    uint32_t *e_cur;                          // [n_act]
    uint32_t *len;                            // [n_act] 0: full length, BSP_LEN_REDO, else effective length (nodes)
    double *rngT;                             // [L][n_act]
    uint32_t *outT;                           // [L][n_act]
    AmbRec *amb;                              // ambiguous steps of the current step
    unsigned long long *amb_count;
    uint32_t *chainq;                         // indices into amb[]
    unsigned long long *chain_count;
    uint32_t *redo_list;
    unsigned long long *redo_count;
    unsigned long long *stats;                // [0] steps [3] dead-end walks [6] list entries read [7] ambiguous [9] chains
};

// wave-aggregated append: returns this lane's slot when `want`
__device__ __forceinline__ unsigned long long wave_append(unsigned long long *counter, bool want) {
    const uint64_t m = ballot(want);
    if (!m) return 0;
    unsigned long long base = 0;
    const int leader = __builtin_ctzll(m);
    if (lane_id() == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
    base = readlane_u64(base, leader);
    return base + (unsigned long long)__popcll(m & ((1ull << lane_id()) - 1ull));
}
// wave-reduced statistics
__device__ __forceinline__ void wave_stat_add(unsigned long long *p, unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += (unsigned long long)__shfl_down((long long)v, (unsigned)off, WAVE);
    if (lane_id() == 0 && v) atomicAdd(p, v);
}

// jobs whose start has neighbours get a walk slot; the others are complete rows already: [start, 0 .., 1]
__global__ void __launch_bounds__(256)
bsp_compact_kernel(BspArgs a) {
    const uint64_t job = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = job < a.n_jobs;
    uint32_t start = 0;
    bool walks = false;
    if (valid) {
        start = a.starts[job];
        walks = a.vrec[start].y != 0;
        uint32_t *row = a.out + job * ((uint64_t)a.L + 2);
        row[0] = start;
        if (!walks) row[a.L + 1] = 1;
    }
    const unsigned long long slot = wave_append(a.n_act, walks);
    if (walks) {
        a.act_job[slot] = (uint32_t)job;
        a.e_cur[slot] = 0;
        a.len[slot] = 0;
    }
}

// rngT[j][w] = draw j of walk w
__global__ void __launch_bounds__(256)
bsp_rng_transpose_kernel(BspArgs a, uint32_t n_act) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_act) return;
    const double *src = a.rng + (a.stream_off[a.act_job[w]] - a.rng_base);
    for (uint32_t j = 0; j < a.L; j++) a.rngT[(size_t)j * n_act + w] = src[j];
}

__global__ void __launch_bounds__(256)
bsp_step_kernel(BspArgs a, uint32_t n_act, uint32_t j) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_steps = 0, n_dead = 0, n_probes = 0, n_amb = 0;
    bool ambiguous = false, redo = false;
    AmbRec rec;
    rec.w = w;
    if (w < n_act) {
        const uint32_t e = a.e_cur[w];
        if (e != BSP_DEAD) {
            uint32_t s0, d, n_in = 0, pp = NOT_FOUND;
            uint64_t coff = 0;
            float wo = 1.0f;   // first step of a walk: no bias (sparse_rw.py:66)
            if (j == 1) {
                const uint4 vr = a.vrec[a.starts[a.act_job[w]]];
                s0 = vr.x; d = vr.y;
            } else {
                const uint4 *rp = (const uint4 *)(a.erec + e);
                const uint4 r0 = rp[0], r1 = rp[1];
                a.outT[(size_t)(j - 2) * n_act + w] = r0.x;   // the vertex reached by step j - 1
                n_in = r0.y; pp = r0.z; d = r0.w;
                s0 = r1.x; coff = ((uint64_t)r1.z << 32) | r1.y;
                wo = a.w_out;
            }
            if (d == 0) {   // dead end before step j: the walk has j nodes (pecanpy.py:196-206)
                a.len[w] = j;
                a.e_cur[w] = BSP_DEAD;
                n_dead = 1;
            } else {
                const double r = a.rngT[(size_t)(j - 1) * n_act + w];
                LaneStep ls{1.0f, 0u, 0u, 0u, 0u, 0u};
                const uint32_t choice = lane_decide(d, n_in, pp, r, wo, a.w_prev, a.clist + coff, ls);
                n_probes = ls.probes;
                if (choice < d) { a.e_cur[w] = s0 + choice; n_steps = 1; }
                else if (choice == LANE_AMBIGUOUS) {
                    ambiguous = true;
                    n_amb = 1;
                    rec.d = d; rec.n_in = n_in; rec.pp = pp; rec.coff = coff; rec.s0 = s0; rec.kmax = ls.kmax;
                    rec.k1 = ls.k1; rec.f = ls.f; rec.shifts = ls.shifts; rec.j = j; rec.tot = ls.tot; rec.wo = wo; rec.r = r;
                } else redo = true;
            }
        }
    }
    const unsigned long long slot = wave_append(a.amb_count, ambiguous);
    if (ambiguous) a.amb[slot] = rec;
    const unsigned long long rslot = wave_append(a.redo_count, redo);
    if (redo) {
        a.redo_list[rslot] = a.act_job[w];
        a.e_cur[w] = BSP_DEAD;
        a.len[w] = BSP_LEN_REDO;
        n_steps -= (j - 1);   // its steps are counted again by the redo pass
    }
    wave_stat_add(a.stats + 0, n_steps);
    wave_stat_add(a.stats + 3, n_dead);
    wave_stat_add(a.stats + 6, n_probes);
    wave_stat_add(a.stats + 7, n_amb);
}

// refined decision of the queued steps (seqscan.h: lane_refine), every lane busy
__global__ void __launch_bounds__(256)
bsp_refine_kernel(BspArgs a) {
    const unsigned long long n = *a.amb_count;
    unsigned long long n_steps = 0, n_probes = 0;
    for (unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x; i0 < n; i0 += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long i = i0 + threadIdx.x;
        bool open = false;
        if (i < n) {
            const AmbRec q = a.amb[i];
            LaneStep ls{q.tot, q.kmax, 0u, q.k1, q.f, q.shifts};
            uint32_t reads = 0;
            const uint32_t res = lane_refine(q.d, q.n_in, q.pp, q.r, q.wo, a.w_prev, a.clist + q.coff, ls, reads);
            n_probes += reads;
            if (res < q.d) { a.e_cur[q.w] = q.s0 + res; n_steps++; }
            else open = true;
        }
        const unsigned long long slot = wave_append(a.chain_count, open);
        if (open) a.chainq[slot] = (uint32_t)i;
    }
    wave_stat_add(a.stats + 0, n_steps);
    wave_stat_add(a.stats + 6, n_probes);
}

// the float32 chain of what the refinement left open (seqscan.h: lane_chain)
__global__ void __launch_bounds__(256)
bsp_chain_kernel(BspArgs a) {
    const unsigned long long n = *a.chain_count;
    unsigned long long n_steps = 0, n_probes = 0, n_chain = 0;
    for (unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x; i0 < n; i0 += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long i = i0 + threadIdx.x;
        bool redo = false;
        uint32_t w = 0, j = 0;
        if (i < n) {
            const AmbRec q = a.amb[a.chainq[i]];
            w = q.w; j = q.j;
            const float x_in = 1.0f / q.tot;
            uint32_t reads = 0;
            const uint32_t res = lane_chain(q.kmax, q.n_in, q.pp, q.r, x_in, x_in * q.wo, x_in * a.w_prev, a.clist + q.coff, reads);
            n_probes += reads;
            n_chain++;
            if (res < q.d) { a.e_cur[w] = q.s0 + res; n_steps++; }
            else redo = true;   // never reached (mirrored overflow read) / tie budget: walk_kernel redoes the job
        }
        const unsigned long long rslot = wave_append(a.redo_count, redo);
        if (redo) {
            a.redo_list[rslot] = a.act_job[w];
            a.e_cur[w] = BSP_DEAD;
            a.len[w] = BSP_LEN_REDO;
            n_steps -= (j - 1);
        }
    }
    wave_stat_add(a.stats + 0, n_steps);
    wave_stat_add(a.stats + 6, n_probes);
    wave_stat_add(a.stats + 9, n_chain);
}

// the vertex reached by the last step
__global__ void __launch_bounds__(256)
bsp_final_kernel(BspArgs a, uint32_t n_act) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_act) return;
    const uint32_t e = a.e_cur[w];
    if (e != BSP_DEAD) a.outT[(size_t)(a.L - 1) * n_act + w] = a.erec[e].nxt;
}

// outT -> rows of the reference's matrix: row = [start, n_1 .. n_L, len]; 64 walks per block through LDS
__global__ void __launch_bounds__(256)
bsp_out_transpose_kernel(BspArgs a, uint32_t n_act) {
    extern __shared__ uint32_t tile[];   // [64][L + 1]
    const uint32_t L = a.L, W = L + 2, w0 = blockIdx.x * 64u;
    const uint32_t nw = n_act - w0 < 64u ? n_act - w0 : 64u;
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;   // 4 steps at a time, 64 walks wide (coalesced)
    for (uint32_t j = ty; j < L; j += 4)
        if (tx < nw) tile[tx * (L + 1) + j] = a.outT[(size_t)j * n_act + w0 + tx];
    __syncthreads();
    for (uint32_t t = ty; t < nw; t += 4) {   // one wave per row: 64 consecutive cells per store
        const uint32_t w = w0 + t;
        const uint32_t ln = a.len[w];
        if (ln == BSP_LEN_REDO) continue;
        const uint32_t steps = ln == 0 ? L : ln - 1u;   // cells 1 .. steps hold vertices, the rest stay 0
        uint32_t *row = a.out + (uint64_t)a.act_job[w] * W;
        for (uint32_t c = tx; c < steps; c += 64) row[1 + c] = tile[t * (L + 1) + c];
        if (tx == 0) row[L + 1] = ln == 0 ? L + 1 : ln;
    }
}

}  // namespace pw
