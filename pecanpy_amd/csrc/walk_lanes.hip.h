// walk_lanes.hip.h -- SparseOTF walk kernel, ONE LANE PER WALK (gfx950), for the headline regime: unit weights,
// 1/p and 1/q powers of two, no self loops (the regime of walk_sparse.hip.h's lazy step); a FLOATS form covers unit
// weights with arbitrary p, q.
//
// Why a second kernel.  walk_kernel (walk_sparse.hip.h) gives every walk a whole wavefront and spends ~1.4 k
// wave instructions per step establishing which neighbours of `cur` are neighbours of `prev` (keys -> Bloom word
// -> index slot) before the exact-arithmetic CDF search can run; it is instruction-issue bound (round-1 PMC:
// scalar unit 72 %, VALU 62 %, HBM 33 %).  With 288 GB of HBM the membership question can be answered ONCE per
// graph instead of once per step -- the LANE INDEX, built by the kernels at the end of this file:
//
//   lines : one 64-byte EDGE LINE per CSR entry e = (u -> v) (walk_sparse.hip.h: ELine): the record { v, |N(u) & N(v)|,
//           position of u in row v, degree(v), indptr[v], list offset } -- everything a step needs about the edge it
//           arrives by, 24 bytes -- and a 40-byte inline area that holds THE LIST ITSELF when it has at most 20
//           entries (uint16 positions, in row v, of the common neighbours of u and v; rows beyond 65 536 entries use
//           uint32 positions and never live in a line).  RMAT-22: 4.2 GB.  One more line per VERTEX behind them
//           (lines[nnz + v]) serves the reference's mirrored "choice == degree" read (App. D quirk 1).
//   clist : the lists that do not fit their line, 16-byte aligned (RMAT-22: 2.7 G entries, 5.2 GB).  The line of such
//           a list keeps 20 (10) PIVOTS of it in its inline area -- every (n / 21)-th entry (seqscan.h: ListView) -- so
//           that the upper levels of a search are probes of the line the step has fetched anyway.
//           A byte budget (PECANPY_AMD_INDEX_BUDGET, default half of the free memory) may leave the LONGEST lists out
//           (EL_NO_LIST): a step that arrives by such an entry is decided by lanes_eager_kernel, the walk stays here.
//
// A step of the reference (SparseOTF.move_forward, src/pecanpy/pecanpy.py:543-559; get_normalized_probs,
// src/pecanpy/rw/sparse_rw.py:51-91) then is: the exact-arithmetic decision of seqscan.h (E(k) = exact mass of
// elements 0..k in units of the smallest weight; first k with E(k) >= ceil(R - z) and E(k) >= ceil(R + z))
// evaluated by ONE LANE: a search over the edge's common-neighbour positions (each P_i splits the row into
// runs of "out" neighbours whose mass is a closed form), no membership work at all.  64 walks advance per
// wavefront instruction.
//
// Round 5, QUAD form: what a step reads of the graph arrives in WHOLE 64-byte sectors fetched by quads of lanes straight into
// LDS -- the edge line the walk enters (record + inline list or pivots: one request per step) and the sectors of an overflow
// list its search looks into (chosen by interpolation between the masses at the ends of the range) -- because the vector memory
// path charges per (instruction, line) request, not per byte (tools/lane_mem_bench.hip, profiles/r05_lane_mem_bench.txt).
//
// Steps the a-priori bound cannot settle (a partial sum of the exact CDF lies within the float32 drift bound of
// the target, 12 % of the steps at RMAT-22) go through two more routines of seqscan.h:
//   lane_tight : the chain's SYSTEMATIC drift bounded from the class counts the decision already has -- arithmetic
//                only, ~850 instructions, settles nine in ten of them.  DEFERRED (round 4): the walk's context waits in a
//                per-wavefront POOL in LDS, the lane takes another walk, and one PASS decides 32+ waiting steps at once.
//   lane_chain : the float32 chain itself (1.2 % of the steps).  Queueing form: the walk is PARKED (SuspRec into a global
//                queue), lanes_chain_kernel settles a whole queue at full width and the next ROUND of this kernel resumes the
//                walks (host loop: pecanpy_amd.hip, launch_lane_walks); the last, small round runs them in place.  CHAINS form
//                (round 5, small job arrays; round 6: also the LATE rounds of a large one, PECANPY_AMD_LATE_CHAINS): the open steps
//                stay in the pool and the wavefront runs their chains itself, 20 at a time -- one launch, no further rounds.
// Other forms of the same kernel: FLOATS (unit weights, 1/p or 1/q not a power of two: a float64-bounded decision from
// closed-form prefix sums and per-line row totals, chains for what it leaves open) and WEIGHTED (the same bound over per-(p, q)
// tables; open steps decided by lanes_eager_weighted_kernel).
// The rare rest -- rows whose total is not exact in float32, tie binades beyond the budget, overflow reads without a
// line -- is not handled here: the job is appended to a redo list and walk_kernel takes the walk over at that step.
// Registers decide everything here: any spill in the step loop costs more than a wave brings (queueing form: 106 VGPRs,
// 4 workgroups per CU with its 36 KB of LDS; in place: 128; CHAINS: 142, 3 per CU).
#pragma once
#include "walk_sparse.hip.h"

namespace pw {

// A walk parked at a step that only the float32 chain can settle: everything the step and the rest of the walk
// need.  The lane kernel appends these to a queue instead of running the chain with a handful of its 64 lanes enabled;
// lanes_chain_kernel settles a whole queue at full width (choice), and the next lane launch resumes the walks.
struct SuspRec {
    uint32_t job, j, s0, d;
    uint32_t n_in, pp, e, coff;               // (e: CSR entry the walk arrived by -- its edge line may hold the list)
    uint32_t kmax, choice, soff_lo, soff_hi;   // (soff: position of the walk's draws in this call's stream block)
    float tot, wo;
    double r;
};
static_assert(sizeof(SuspRec) == 64, "queue record is four 16-byte accesses");

// A step the interval decision (lane_tight) settled, kept for verification: what lane_chain needs to decide it again.
struct VerRec {
    uint32_t kmax, n_in, pp, choice;
    uint32_t e, coff, d, job;                // (job: the walk this step belongs to -- a mismatch sends it to walk_kernel)
    float tot, wo;
    double r;
};
static_assert(sizeof(VerRec) == 48, "verification record is three 16-byte stores");

constexpr uint32_t LANE_NEEDS_WAVE = 0xfffffff9u;   // a step only walk_kernel can decide (tie budget, row outside the exact range)
constexpr uint32_t LANE_EAGER_MARK = 0xffffffffu;   // SuspRec::kmax of a step parked because its entry's list is not in the (partial)
                                                    // index: lanes_eager_kernel decides it, lanes_chain_kernel leaves it alone

struct LanesArgs {
    const ELine *__restrict__ lines;          // [nnz] edge lines, then [n_nodes] OVERFLOW lines (when vlines != 0)
    const uint8_t *__restrict__ clist;
    uint32_t vlines;                          // lines[nnz + v] = the line of the mirrored overflow read at vertex v
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
                                              // [6] list entries read [7] steps the a-priori bound left open
                                              // [9] of those, steps that needed the float32 chain
    uint32_t *redo_list;                      // jobs handed to walk_kernel
    unsigned long long *redo_count;
    float w_out, w_prev;                      // fl32(1/q), fl32(1/p): powers of two (host checked)
    SuspRec *susp;                            // queue for walks that need the float chain (nullptr: chains run in-kernel)
    unsigned long long *susp_count;
    uint32_t susp_chunk;                      // queue slots a wavefront reserves at a time (1: exact, no void slots)
    uint32_t job_chunk;                       // jobs a wavefront reserves at a time while plenty are left
    const SuspRec *resume;                    // walks to take up again (their `choice` settled) INSTEAD of fresh jobs
    uint64_t n_resume;
    // verification mode (PECANPY_AMD_VERIFY_TIGHT=1, kernels instantiated with VERIFY): every step lane_tight settles
    // is ALSO recorded here and re-decided by the float32 chain itself (lanes_verify_kernel)
    VerRec *ver;
    unsigned long long *ver_count;            // [0] records appended (may exceed ver_cap: the excess is dropped and counted)
    uint64_t ver_cap;
    uint32_t ver_poison;                      // PECANPY_AMD_VERIFY_TIGHT=poison: every 1024th RECORD (not the walk) gets a wrong
                                              // position, which the check must report -- proves the check is live
    uint32_t ver_mask;                        // a settled step of (job, j) is recorded when ((job + 7919 j) & ver_mask) == 0: 0 = every
                                              // step (test mode), 1023 = the production sample (round 6: a safety net under the argued
                                              // bounds of lane_tight, ~1e-3 of the interval decisions re-decided by the float chain)
    // WEIGHTED form (weighted CSR graphs; per (p, q, extend, thresholds) tables built by wbase / wprefix / wlist kernels)
    const PrefixPair *wpq;                    // [nnz] per-row inclusive float64 prefix sums of the base values (+ their running sums)
    const double *wdl;                        // per-entry lists of delta prefix sums (entry e: wdl + wl_off[e], n_in values)
    const unsigned long long *wl_off;         // [nnz] offset of entry e's deltas; ~0: no table for this entry (eager step)
    const double *wl_dprev;                   // [nnz] (step value - base value) of prev's element when arriving by entry e
    const float *tot_e;                       // [nnz] the step's normaliser (sequential float32 row total) by arriving entry
    const PrefixPair *wp1;                    // [nnz] per-row inclusive float64 prefix sums of the RAW weights (first step of a walk)
    uint32_t wl_pos;                          // q >= 1: the common neighbours' (step value - base value) are all >= 0
    const float *tot_v;                       // [n_nodes] ... and its normaliser, by vertex
};

// (per-lane exact decision: lane_decide / LaneStep in seqscan.h, shared with the host self test)

// Optional section timing of the lane kernel (-DPW_PROF_LANES builds; tools/prof_lanes.sh): wave-level cycle sums
//   [0] refill [1] draw + exact decision [2] refined decisions [3] edge record + store [4] float chains   and counts
//   [8] loop iterations [9] refinement passes [10] lanes in them [5] chain passes [6] lanes in them
#ifdef PW_PROF_LANES
__device__ unsigned long long g_lprof[16];
#define LPROF_T(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) lp[i] += now_ - lp_last; lp_last = __builtin_readcyclecounter(); } while (0)
#define LPROF_C(i, n) do { if (lane == 0) lp[i] += (n); } while (0)
#else
#define LPROF_T(i)
#define LPROF_C(i, n)
#endif

// -DPW_LANES_WATCHDOG builds: a wavefront whose main / refill loop runs away records its state and leaves
#ifdef PW_LANES_WATCHDOG
#define PW_WD(which, limit, counter)                                                                              \
    do {                                                                                                          \
        if (++(counter) > (limit)) {                                                                              \
            const uint64_t act_ = ballot((A.flags & F_ACTIVE) != 0), wt_ = ballot((A.flags & F_WAIT2) != 0);      \
            const uint64_t ex_ = ballot(exhausted), amb_ = ballot(choice_wd == LANE_AMBIGUOUS);                   \
            if (lane == 0 && atomicAdd(&g_wd[0], 1ull) == 0ull) {                                                 \
                g_wd[1] = (which); g_wd[2] = act_; g_wd[3] = wt_; g_wd[4] = ex_; g_wd[5] = amb_;                  \
                g_wd[6] = pool_lo; g_wd[7] = pool_hi; g_wd[8] = n_work; g_wd[9] = blockIdx.x;                     \
                g_wd[10] = A.flags; g_wd[11] = A.j; g_wd[12] = A.d; g_wd[13] = A.n_in; g_wd[14] = A.job;          \
            }                                                                                                     \
            return;                                                                                               \
        }                                                                                                         \
    } while (0)
#else
#define PW_WD(which, limit, counter)
#endif

#ifndef PW_LANES_MIN_WAVES
#define PW_LANES_MIN_WAVES 4   // 128 VGPRs: the in-place form needs ~130 (chain code); at 5-6 waves it spills in the hot loop and runs 1.6-2x slower
#endif
#ifndef PW_LANES_MIN_WAVES_Q
#define PW_LANES_MIN_WAVES_Q 6   // ... of the queueing form (no chain code; statistics in scalar registers): 80 VGPRs, six waves without spilling
#endif
#ifndef PW_LANES_DEFER
#define PW_LANES_DEFER 1         // queueing form: ambiguous steps wait in a per-wavefront LDS pool for a full-width interval decision
#endif
#ifndef PW_LANES_MIN_WAVES_D
#define PW_LANES_MIN_WAVES_D 5   // ... its occupancy (96 VGPRs; 5 KB of pool + 2 KB of job window per wavefront: 5 x 4 x 7 KB = 140 KB of LDS per CU)
#endif
#ifndef PW_LANES_DEFER_TH
#define PW_LANES_DEFER_TH 32     // deferred steps that trigger a pass of the interval decision (lanes enabled in it)
#endif
#ifndef PW_LANES_POOL
#define PW_LANES_POOL 64         // slots of the pool (slot s is decided by lane s: at most 64)
#endif
#ifndef PW_LANES_WIN
#define PW_LANES_WIN 64          // jobs prefetched into the job window at a time
#endif
#ifndef PW_LANES_DRAW_LDS
#define PW_LANES_DRAW_LDS 0      // 1: a walk's draws are fetched a 64-byte sector (8 draws) at a time, straight into LDS
#endif
#ifndef PW_LANES_LINE_LDS
#define PW_LANES_LINE_LDS 0      // 1: the rest of the edge line a step enters (inline list / pivots) is copied to LDS with the record
#endif
#ifndef PW_LANES_QUAD
#define PW_LANES_QUAD 1          // round 5: edge lines and list sectors are fetched WHOLE (64 bytes) by quads of lanes, straight into LDS
#endif
#ifndef PW_LANES_QPOOL
#define PW_LANES_QPOOL 48        // ... pool slots / window jobs of that form (LDS: 4 KB of lines per wavefront on top)
#endif
#ifndef PW_LANES_QWIN
#define PW_LANES_QWIN 32
#endif
#ifndef PW_LANES_MIN_WAVES_QD
#define PW_LANES_MIN_WAVES_QD 4  // ... its occupancy: 35 KB of LDS per workgroup = 4 per CU; the memory path, not the wave count, sets its pace
#endif
#ifndef PW_LANES_QPIPE
#define PW_LANES_QPIPE 0         // 1: the line requested when a step is taken is applied at the top of the NEXT iteration, behind the pool's pass
                                 // (measured: 122 vs 118 ms per RMAT-22 pass -- the wait then also covers the stores the pass has just issued)
#endif
#ifndef PW_LANES_QDRAW_EARLY
#define PW_LANES_QDRAW_EARLY 1   // the next step's draw is requested together with the line, not after the record has arrived
#endif
#ifndef PW_LANES_CHAIN_TH
#define PW_LANES_CHAIN_TH 20     // CHAINS form: steps waiting for their float chain that trigger a chain pass of the pool (28: the pool
                                 // fills up and steps are parked after all -- 5.2 M jobs at RMAT-22: 23.8 ms, two rounds; 20: 21.4, one; 12: 22.8)
#endif
#ifndef PW_LANES_CPOOL
#define PW_LANES_CPOOL 64        // ... its pool slots / window jobs
#endif
#ifndef PW_LANES_CWIN
#define PW_LANES_CWIN 64
#endif
#ifndef PW_LANES_MIN_WAVES_C
#define PW_LANES_MIN_WAVES_C 3   // ... its occupancy (the chain code needs ~150 VGPRs)
#endif
#ifndef PW_LANES_FWAIT
#define PW_LANES_FWAIT 16        // FLOATS form: steps left open by the bound AND by the interval decision gather before their float
                                 // chains run together (round 5, without the interval decision: 40)
#endif
#ifndef PW_LANES_FTIGHT
#define PW_LANES_FTIGHT 16       // FLOATS form (round 6): steps the float64 bound leaves open gather before the interval decision
                                 // (lane_tight_values, ~900 instructions, no memory access) runs for all of them
#endif
#ifndef PW_LANES_CHUNK
#define PW_LANES_CHUNK 1024   // most jobs a wavefront reserves per access to the shared job counter (host: a quarter of
                              // its share of the work at most)
#endif
#ifndef PW_LANES_WAIT2
#define PW_LANES_WAIT2 2      // in-place form: ambiguous lanes that gather before their float chains run
#endif

// Takes the sampled edge of walk A: choice >= d hands the job to walk_kernel (overflow read / precondition / tie),
// otherwise the record of the sampled entry's 64-byte line names the next vertex and everything the next step needs.
// HEAD: the entry the walk moves to (A.e; fetch_ = its line is wanted) or the hand-over; TAIL: the record (r0_: first 16
// bytes of the line, r1_: the next 8) applied to the walk.  Between the two the line is loaded -- by the lane itself, or by
// its quad into LDS (QUAD form).
#define PW_LANE_APPLY_HEAD(fetch_)                                                              \
    do {                                                                                        \
        uint32_t oe_ = NOT_FOUND;                                                               \
        if (choice == A.d && a.vlines) {                                                        \
            /* The CDF never reached r: the reference reads indices[indptr[cur] + degree], the first neighbour of the  \
               next non-empty row (App. D quirk 1) -- a vertex that only depends on cur.  Its edge line (next vertex,   \
               common neighbours of cur and that vertex, ...) was built with the index: lines[nnz + cur]. */            \
            const uint32_t vtx_ = A.j == 1u ? a.starts[A.job] : a.lines[A.e].nxt;                \
            if (a.lines[(uint64_t)a.nnz + vtx_].nxt != NOT_FOUND) oe_ = a.nnz + vtx_;           \
        }                                                                                       \
        if (choice >= A.d && oe_ == NOT_FOUND) {   /* walk_kernel takes the walk over AT THIS STEP (WalkArgs::resume) */ \
            const unsigned long long slot_ = atomicAdd(LAP(unsigned long long *, redo_count), 1ull);                     \
            LAP(uint32_t *, redo_list)[slot_] = A.job;                                                         \
            const uint32_t st_ = (A.j - 1u) & 3u;   /* staged cells out, length cell = A.j */     \
            uint32_t *row_ = a.out + (uint64_t)A.job * W;                                       \
            if (st_ >= 1u) row_[A.j - st_] = ob.v[0];                                           \
            if (st_ >= 2u) row_[A.j - st_ + 1u] = ob.v[1];                                      \
            if (st_ >= 3u) row_[A.j - st_ + 2u] = ob.v[2];                                      \
            row_[L + 1] = A.j;                                                                  \
            A.flags = 0;                                                                        \
        } else {                                                                                \
            A.e = oe_ != NOT_FOUND ? oe_ : A.s0 + choice;                                       \
            if (oe_ != NOT_FOUND) {   /* (~1e-5 of the steps: counted where it happens; a sampled transition too) */ \
                atomicAdd(LAP(unsigned long long *, stats) + 1, 1ull); atomicAdd(LAP(unsigned long long *, stats) + 0, 1ull);                     \
            }                                                                                   \
            fetch_ = true;                                                                      \
        }                                                                                       \
    } while (0)
#define PW_LANE_APPLY_TAIL(r0_, r1_, draw_prefetched_)                                          \
    do {                                                                                        \
            {   /* output cells are staged four steps at a time: one 16-byte store instead of four 4-byte ones */ \
                const uint32_t slot_ = (A.j - 1u) & 3u;                                         \
                ob.v[0] = slot_ == 0u ? r0_.x : ob.v[0];                                        \
                ob.v[1] = slot_ == 1u ? r0_.x : ob.v[1];                                        \
                ob.v[2] = slot_ == 2u ? r0_.x : ob.v[2];                                        \
                ob.v[3] = slot_ == 3u ? r0_.x : ob.v[3];                                        \
                const bool last_ = A.j + 1u > L || r0_.w == 0u;                                 \
                uint32_t *cell_ = a.out + (uint64_t)A.job * W + (A.j - slot_);                  \
                if (slot_ == 3u) *(OutCells *)cell_ = ob;                                       \
                else if (last_) {                                                               \
                    cell_[0] = ob.v[0];                                                         \
                    if (slot_ >= 1u) cell_[1] = ob.v[1];                                        \
                    if (slot_ >= 2u) cell_[2] = ob.v[2];                                        \
                }                                                                               \
            }                                                                                   \
            A.n_in = r0_.y; A.pp = r0_.z; A.d = r0_.w;                                          \
            A.s0 = r1_.x; A.coff = r1_.y;                                                       \
            A.j++;                                                                              \
            if (A.j > L || A.d == 0) {                                                          \
                uint32_t *row_ = a.out + (uint64_t)A.job * W;                                   \
                row_[L + 1] = A.j;               /* effective length (pecanpy.py:196-206) */    \
                if (A.j <= L) {                  /* dead end: the remaining cells are 0 */      \
                    n_dead++;                                                                   \
                    for (uint32_t z_ = A.j; z_ <= L; z_++) row_[z_] = 0;                        \
                }                                                                               \
                A.flags = 0;                                                                    \
            } else if (PW_LANES_DRAW_LDS) { PW_DRAW_STAGE(A.soff + (A.j - 1)); }                \
            else if (!(draw_prefetched_)) r = a.rng[A.soff + (A.j - 1)];   /* the next step's draw, asked for now */ \
    } while (0)

// ---- a binade with a ROUNDING TIE, walked by the whole wavefront (round 6) ------------------------------------------------------
// In such a binade a value sits exactly half way between two sums (round half to even depends on the parity of the running sum),
// so the class counts no longer determine the sum and lane_chain walks the binade RUN BY RUN: one iteration per common neighbour,
// ~10^4 on the largest hub rows, ONE lane busy -- 7.7 of the chain kernels' 17.6 ms per RMAT-22 pass, and the tail of every chain
// launch.  Here the 64 lanes take 64 consecutive list entries per trip.  Entry j at position P_j stands for the run of m_j "out"
// elements before it and the common neighbour itself; what that adds to an even / an odd sum is a PARITY FUNCTION (a0, a1) in
// closed form (a run adds qo.a[parity] once and qo.a0 for each further element -- after a tying addition the sum is even, without
// a tie a0 == a1 --, the common neighbour qi.a[parity behind the run]); parity functions compose associatively (seqscan.h:
// Binade::compose), so one wave scan (wave_scan_inc: DPP) yields the sum behind every entry of the trip, a ballot the first entry
// at which it reaches the target or the binade's top, and the element inside that entry's run is the division lane_chain does.
// Same integers as the sequential loop, entry by entry (A/B: the walk matrices of the RMAT-22 passes, tests/).
// All arguments are wave uniform; lp / wide: the list in global memory (uint16 / uint32 positions).
struct TieOut {
    uint32_t leave;     // the sum reached Tt (target or top of the binade) at element kf
    uint32_t at_in;     // ... kf is a common neighbour (else an "out" element)
    uint32_t kf;
    uint32_t k, i0;     // next element and number of common neighbours before it (leave: i0 behind kf's entry when at_in)
    uint32_t Cc, Cprev; // sum in ulps of the binade behind k - 1 (leave: behind kf, and before kf)
};
__device__ __forceinline__ TieOut coop_tie_binade(const void *lp, uint32_t wide, uint32_t n_in, uint32_t k, uint32_t lim, uint32_t i0,
                                                  uint32_t C, uint32_t Tt, Inc<float> qi, Inc<float> qo) {
    using B = Binade<float>;
    const int lane = lane_id();
    TieOut o;
    o.leave = 0u; o.at_in = 0u; o.kf = 0u; o.k = k; o.i0 = i0; o.Cc = C; o.Cprev = C;
    // the run of m "out" elements from sum Cx: does it reach Tt, and where
    auto out_run = [&](uint32_t Cx, uint32_t k0, uint32_t m) -> bool {
        const uint64_t first = (Cx & 1u) ? qo.a1 : qo.a0, each = qo.a0;
        const uint64_t Cend = (uint64_t)Cx + first + (uint64_t)(m - 1u) * each;
        if (Cend < (uint64_t)Tt) { o.Cc = (uint32_t)Cend; return false; }
        uint64_t t = 1;
        if ((uint64_t)Cx + first < (uint64_t)Tt) t = 2ull + div_floor_small((uint64_t)Tt - ((uint64_t)Cx + first) - 1ull, each);
        const uint64_t Cf = (uint64_t)Cx + first + (t - 1ull) * each;
        o.kf = k0 + (uint32_t)t - 1u;
        o.Cprev = t == 1 ? Cx : (uint32_t)(Cf - each);
        o.Cc = (uint32_t)(Cf > 0xffffffffull ? 0xffffffffull : Cf);
        o.leave = 1u; o.at_in = 0u;
        return true;
    };
    for (;;) {
        const uint32_t j = o.i0 + (uint32_t)lane;
        uint32_t P = 0xffffffffu;
        if (j < n_in) P = wide ? ((const uint32_t *)lp)[j] : (uint32_t)((const uint16_t *)lp)[j];
        const bool in_range = P < lim;                                   // (ascending: the in-range entries are a prefix of the lanes)
        uint32_t prevP = (uint32_t)__shfl_up((int)P, 1u, WAVE);
        if (lane == 0) prevP = o.k - 1u;                                 // (k >= 1: the head was added one by one)
        const uint32_t m = in_range ? P - prevP - 1u : 0u;
        Inc<float> F;
        F.a0 = 0u; F.a1 = 0u;
        if (in_range) {
            uint64_t v0 = 0, v1 = 0;
            if (m) {
                const uint64_t rest = (uint64_t)(m - 1u) * (uint64_t)qo.a0;
                v0 = (uint64_t)qo.a0 + rest;
                v1 = (uint64_t)qo.a1 + rest;
            }
            const uint64_t t0 = v0 + (uint64_t)((v0 & 1ull) ? qi.a1 : qi.a0);
            const uint64_t t1 = v1 + (uint64_t)(((v1 + 1ull) & 1ull) ? qi.a1 : qi.a0);
            F.a0 = t0 > (uint64_t)B::SAT ? B::SAT : (uint32_t)t0;
            F.a1 = t1 > (uint64_t)B::SAT ? B::SAT : (uint32_t)t1;
        }
        const Inc<float> G = wave_scan_inc<float>(F);
        uint64_t ci = (uint64_t)o.Cc + (uint64_t)((o.Cc & 1u) ? G.a1 : G.a0);
        const uint32_t Cincl = ci > (uint64_t)B::SAT ? B::SAT : (uint32_t)ci;   // sum behind this entry's common neighbour
        const uint64_t rng = ballot(in_range);
        const uint32_t n_valid = (uint32_t)__popcll(rng);
        const uint64_t hit = ballot(in_range && Cincl >= Tt);
        if (hit) {
            const int l = __builtin_ctzll(hit);
            const uint32_t Cx = l ? readlane_u32(Cincl, l - 1) : o.Cc;
            const uint32_t Pl = readlane_u32(P, l), ml = readlane_u32(m, l);
            o.i0 += (uint32_t)l;
            if (ml && out_run(Cx, Pl - ml, ml)) { o.k = o.kf + 1u; return o; }
            const uint32_t Cb = ml ? o.Cc : Cx;                          // (out_run left the sum behind the run in o.Cc)
            const uint32_t inc = (Cb & 1u) ? qi.a1 : qi.a0;
            o.kf = Pl; o.Cprev = Cb; o.Cc = Cb + inc; o.leave = 1u; o.at_in = 1u;
            o.i0 += 1u; o.k = Pl + 1u;
            return o;
        }
        if (n_valid) {
            o.Cc = readlane_u32(Cincl, (int)n_valid - 1);
            o.k = readlane_u32(P, (int)n_valid - 1) + 1u;
            o.i0 += n_valid;
        }
        if (n_valid < (uint32_t)WAVE) break;                             // the list is exhausted, or its next entry lies beyond lim
    }
    if (o.k < lim) {                                                     // the "out" elements up to lim
        if (out_run(o.Cc, o.k, lim - o.k)) { o.k = o.kf + 1u; return o; }
        o.k = lim;
    }
    return o;
}

#ifndef PW_CHAIN_COOP
#define PW_CHAIN_COOP 1
#endif
#ifndef PW_CHAIN_STEP
#define PW_CHAIN_STEP 1   // the wavefront's chains advance one binade at a time, in step (tying binades are walked between the steps)
#endif
// The float32 chains of the lanes with `active` set, by the wavefront: every chain advances one iteration of lane_chain's binade
// loop at a time (PW_CHAIN_STEP), a lane that stops in front of a binade with a rounding tie has it walked by all 64 lanes
// (coop_tie_binade) and goes on.  CONVERGED code: every lane of the wavefront calls it.  list_p: the list in global memory
// (cl.p before any staging).  Returns what lane_chain returns (position, LANE_CHAIN_END, LANE_TIE).
__device__ __forceinline__ uint32_t lane_chain_wave(bool active, uint32_t kend, uint32_t n_in, uint32_t pp, double r, float x_in,
                                                    float x_out, float x_prev, const ListView &cl, const void *list_p, uint32_t &reads_total) {
    using B = Binade<float>;
    ChainResume rs;
    rs.c = 0.0f; rs.k = 0u; rs.i0 = 0u; rs.started = 0u; rs.yield = PW_CHAIN_STEP ? 1u : 0u;
    uint32_t res = 0u;
    reads_total = 0;
    if (active) {
        uint32_t reads = 0;
        res = lane_chain(kend, n_in, pp, r, x_in, x_out, x_prev, cl, reads, nullptr, LANE_TIE_BUDGET, PW_CHAIN_COOP ? &rs : nullptr);
        reads_total += reads;
    }
#if PW_CHAIN_COOP
    for (;;) {
        uint64_t pend = ballot(active && res == LANE_TIE_PENDING);
        if (!pend && !ballot(active && res == LANE_YIELD)) break;
        while (pend) {
            const int l = __builtin_ctzll(pend);
            pend &= pend - 1ull;
            // the owner's chain, wave uniform
            const float c_l = readlane_f32(rs.c, l), xi_l = readlane_f32(x_in, l), xo_l = readlane_f32(x_out, l);
            const uint32_t k_l = readlane_u32(rs.k, l), i0_l = readlane_u32(rs.i0, l);
            const uint32_t kend_l = readlane_u32(kend, l), nin_l = readlane_u32(n_in, l), pp_l = readlane_u32(pp, l);
            const double r_l = readlane_f64(r, l);
            const void *lp_l = (const void *)readlane_u64((uint64_t)(uintptr_t)list_p, l);
            const uint32_t wide_l = readlane_u32(cl.wide, l);
            const uint32_t lim_l = (pp_l != 0xffffffffu && pp_l > k_l && pp_l < kend_l) ? pp_l : kend_l;
            const int eb = B::eb_of(c_l);
            const uint32_t C = B::sig_of(c_l);
            const uint32_t Tt = (uint32_t)B::threshold(r_l, eb);
            const Inc<float> qi = B::quantize(xi_l, eb), qo = B::quantize(xo_l, eb);
            const TieOut o = coop_tie_binade(lp_l, wide_l, nin_l, k_l, lim_l, i0_l, C, Tt, qi, qo);
            if (lane_id() == l) {                  // what lane_chain does behind the binade's loop
                reads_total += o.i0 - i0_l;
                if (!o.leave) {                    // [k, lim) stays inside the binade and below the target
                    const float cn = B::make(o.Cc, eb);
                    if (lim_l == kend_l) res = LANE_CHAIN_END;
                    else { rs.c = cn; rs.k = lim_l; rs.i0 = o.i0; res = LANE_YIELD; }   // (k == lim == pp: prev is added next)
                } else if (o.Cc < (uint32_t)B::TOP) res = o.kf;             // target reached inside the binade
                else {
                    const float cn = B::make(o.Cprev, eb) + (o.at_in ? xi_l : xo_l);
                    if ((double)cn >= r_l) res = o.kf;
                    else if (o.kf + 1u >= kend_l) res = LANE_CHAIN_END;     // (no element left: the chain ends below the target)
                    else { rs.c = cn; rs.k = o.kf + 1u; rs.i0 = o.i0; res = LANE_YIELD; }
                }
            }
        }
        if (active && res == LANE_YIELD) {   // (the chain goes on: behind its tying binade, or -- PW_CHAIN_STEP -- one iteration at a time)
            uint32_t reads = 0;
            res = lane_chain(kend, n_in, pp, r, x_in, x_out, x_prev, cl, reads, nullptr, LANE_TIE_BUDGET, &rs);
            reads_total += reads;
        }
    }
#endif
    return res;
}


struct __attribute__((packed, aligned(4))) OutCells {   // four staged output cells: one 16-byte store, 4-byte aligned
    uint32_t v[4];
};

// INPLACE: steps the interval decision leaves open run their float chain in the kernel (waiting lanes, below) -- the
// form used when no queue is given (a.susp == nullptr: short job lists, the last round).  !INPLACE: such walks are
// always parked (a.susp != nullptr); without the chain code the kernel needs fewer registers.
// VERIFY: steps settled by lane_tight are recorded for lanes_verify_kernel (test mode, PECANPY_AMD_VERIFY_TIGHT=1).
// FLOATS (with INPLACE): 1/p or 1/q is not a power of two -- the row values are arbitrary float32 numbers, no exact
// integer decision exists, and EVERY step is the reference's two float32 chains, each evaluated by the lane in closed
// form per binade (lane_chain): the row total w.sum() (sparse_rw.py:89), then cumsum / searchsorted over w / tot
// (pecanpy.py:556-557).  ~6x fewer wave instructions per step than walk_kernel's eager step, which gives every walk a
// whole wavefront.
// DEFER (queueing form, !INPLACE, PW_LANES_DEFER): the interval decision (lane_tight, ~850 instructions, no memory access)
// is NOT run where the ambiguity turns up -- 12 % of the lanes, so nearly every loop iteration would pay for it with an
// eighth of its lanes enabled (round 3: ~40 % of the kernel's vector instructions).  The lane writes the walk's context
// (80 bytes: walk state, draw, what lane_decide found, staged output cells) into a POOL of 64 slots in LDS, one pool per
// wavefront, and takes another walk at once; when PW_LANES_DEFER_TH slots wait (or nothing else can run) one PASS decides
// them all, slot s by lane s -- half of the lanes or more enabled.  Settled walks stay in the pool until a lane is free
// (they are picked up BEFORE fresh jobs, with their step's choice known: the resume path, F_PRE); the ones the interval
// decision leaves open go to the global queue of parked walks from there.  Slots are handed out by rank (r-th deferring
// lane <- r-th free slot, through a 64-byte map in LDS); a step that finds no free slot is parked in the global queue
// undecided (lanes_chain_kernel settles it by the float chain: any ambiguous step may go there).
// TAILS: bytes 16..63 of the edge line a step enters (the inline list, or the pivots of a longer one) are copied to LDS with
// the record's load (LDS-DMA, no registers) and the next step's searches read them there.  Chosen by the host for graphs
// whose lines stay cache resident (RMAT-18: the probes it removes were L2 hits, 15.0 -> 13.4 ms per pass); at RMAT-22 the
// dependent HBM probes of the long lists set the pace and it is neutral (139.7 vs 138.8 ms), so the plain form stays.
// The LDS it takes comes out of the pool and the job window (32 slots / 32 jobs instead of 64 / 64: measured equal).
// WEIGHTED (round 4): weighted CSR graphs, node2vec and node2vec+.  A step is decided by lane_decide_weighted (seqscan.h)
// from the per-vertex float64 prefix sums of the base values + the arriving entry's delta prefix sums, with a rigorous
// bound on the float32 chain; what the bound leaves open (RMAT-20 with hashed weights: 10 % of the steps), the first
// step of every walk and arrivals without a table are parked for lanes_eager_kernel<true, ..> -- the wave-per-walk scan
// with the normaliser from the per-entry table -- and resumed by the next round.  No pool, no interval decision, no
// float chain per lane here; without a queue (small job lists, the last round) such a walk goes to walk_kernel for good.
// CHAINS (round 5; queueing form): a step the interval decision leaves open is NOT parked in the global queue for lanes_chain_kernel
// and the next round -- it stays in its pool slot, and when PW_LANES_CHAIN_TH such steps wait the wavefront runs their float
// chains itself, slot s by lane s, like the passes of the interval decision.  The walk is then taken up by a free lane in the
// SAME launch: no rounds, no chain launches, no tails of half-empty rounds -- the form for job arrays of a few walks per
// resident lane (a shard of a multi-GPU run, RMAT-18..20 sized calls), where every round lasts as long as its slowest walks.
// The chain code costs registers (3 waves per SIMD), so whole-array calls at RMAT-22 keep the rounds.  Only a step that finds
// the pool full is parked (the host's round loop handles what is left, usually nothing).
template <bool INPLACE, bool VERIFY, bool FLOATS = false, bool TAILS = false, bool WEIGHTED = false, bool CHAINS = false>
__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE, CHAINS ? PW_LANES_MIN_WAVES_C : INPLACE ? PW_LANES_MIN_WAVES
                                      : ((PW_LANES_QUAD && PW_LANES_DEFER && !FLOATS && !WEIGHTED) ? PW_LANES_MIN_WAVES_QD
                                         : (PW_LANES_DEFER ? PW_LANES_MIN_WAVES_D : PW_LANES_MIN_WAVES_Q)))
walk_lanes_kernel(LanesArgs a) {
    constexpr bool DEFER = PW_LANES_DEFER && !INPLACE && !FLOATS && !WEIGHTED;
    static_assert(!CHAINS || (DEFER && !TAILS), "the CHAINS form is the deferred queueing form plus in-kernel chain passes");
    // QUAD (round 5; the dyadic forms): what a step reads of the graph comes in WHOLE 64-byte sectors, fetched by quads of
    // lanes straight into LDS (global_load_lds, 16 bytes per lane: lane l moves piece l & 3 of the sector wanted by lane
    // 16 k + (l >> 2), k = 0..3 -- the 64 bytes of lane w's sector land contiguously at byte 64 w of the wavefront's buffer):
    // the edge line a step enters (record + inline list or pivots), and every sector of an overflow list a search looks
    // into.  The memory path charges per (instruction, line) REQUEST, not per byte (tools/lane_mem_bench.hip,
    // profiles/r05_lane_mem_bench.txt: whole lines by quads 52.9 G lines/s; the 16 + 8-byte record load of rounds 2-4
    // 29.1 G/s; three dependent 2-byte probes of the same line on top of the record 15.7 G/s), so the record load, the
    // probes of the inline list / the pivots and the last five levels of every overflow-list bisection -- 5.7 requests per
    // step in round 4 -- become one request per line entered plus one per list sector visited.
    constexpr bool QUAD = PW_LANES_QUAD && !FLOATS && !WEIGHTED && !TAILS;
    constexpr bool LINE_LDS = (TAILS || PW_LANES_LINE_LDS) && !QUAD;
    static_assert(!(QUAD && PW_LANES_DRAW_LDS), "the QUAD form fetches its draws with the lines");
    constexpr int POOL_N = CHAINS ? PW_LANES_CPOOL : QUAD ? (PW_LANES_POOL < PW_LANES_QPOOL ? PW_LANES_POOL : PW_LANES_QPOOL)
                                : (TAILS ? (PW_LANES_POOL < 32 ? PW_LANES_POOL : 32) : PW_LANES_POOL);
    constexpr int WIN_N = CHAINS ? PW_LANES_CWIN : QUAD ? (PW_LANES_WIN < PW_LANES_QWIN ? PW_LANES_WIN : PW_LANES_QWIN)
                               : (TAILS ? (PW_LANES_WIN < 32 ? PW_LANES_WIN : 32) : PW_LANES_WIN);
    constexpr uint32_t DEFER_TH = TAILS ? (PW_LANES_DEFER_TH < 16 ? PW_LANES_DEFER_TH : 16) : PW_LANES_DEFER_TH;
    const int lane = lane_id();
    const uint32_t L = a.L;
    const uint64_t W = (uint64_t)L + 2;
    const uint64_t n_work = a.resume ? a.n_resume : (a.job_list ? a.n_list : a.n_jobs);
    // Rarely used kernel arguments -- counters, queues, the job list, the verification buffer -- are RE-READ from the kernarg segment
    // where they are used (wave.h: kernarg<T>, an s_load) instead of staying live across the loop: kept live they do not fit
    // the 102 scalar registers of a wavefront, the allocator parks them in VGPR lanes and every use becomes one v_readlane per
    // dword -- 430 of them in the QUAD instantiation before this (round 6, fourth session), each a vector-ALU issue slot of a
    // kernel that is bound by exactly those.
#define LA64(field) kernarg<uint64_t>(offsetof(LanesArgs, field))
#define LA32(field) kernarg<uint32_t>(offsetof(LanesArgs, field))
#define LAP(T, field) ((T)LA64(field))
    const bool resuming = a.resume != nullptr;
    const float w_out = a.w_out, w_prev = a.w_prev;
    const uint64_t lane_lt = (1ull << lane) - 1ull;
#ifdef PW_PROF_LANES
    unsigned long long lp[16] = {0};
    unsigned long long lp_last = __builtin_readcyclecounter();
#endif

    // per-lane walk state (prefix A: the macro PW_LANE_APPLY works on it)
    struct Walk {
        uint32_t job, j;                 // j = index of the step being sampled (1..L)
        uint64_t soff;
        uint32_t s0, d, n_in, pp;        // row of the current vertex; edge it was entered by
        uint32_t e, coff;                // ... its CSR entry and the offset of its list
        uint32_t flags;                  // bit 0: active, bit 2: waiting for the float chain, bit 3: resumed
    };
    constexpr uint32_t F_ACTIVE = 1u, F_WAIT2 = 4u;   // waiting for the float chain (in-place form; FLOATS: for the interval decision)
    constexpr uint32_t F_PRE = 8u;                    // resumed walk: the pending step's choice is known (pre)
    constexpr uint32_t F_WAIT3 = 16u;                 // FLOATS: the interval decision left the step open -- waiting for the float chain
    // FLOATS: what the bounded decision knows about a step it left open (seqscan.h: BoundedAmb), kept while the lane waits
    uint32_t tk1 = 0, ti1 = 0, tpn = 0;
    float tz = 0.0f;
    uint32_t pre = 0;
    Walk A{0, 1, 0, 0, 0, 0, NOT_FOUND, 0, 0, 0};
    bool exhausted = false;
    uint64_t pool_lo = 0, pool_hi = 0;   // wavefront-uniform: job indices reserved from the shared counter
    // JOB WINDOW: what a new walk needs before its first step -- job, start vertex, its row, its stream position, its
    // first draw: a chain of three dependent scattered loads -- is fetched for the NEXT 64 jobs of the pool at once and
    // parked in LDS; a refill then costs one LDS read instead of that chain (nearly every loop iteration refills a lane
    // or two: ~40 steps per walk, 64 lanes).  Round 5: only jobs whose start has neighbours enter the window (refill loop below).
    struct JobSlot { uint32_t job, start, s0, d; uint32_t soff_lo, soff_hi, r_lo, r_hi; };
    __shared__ JobSlot s_win[WAVES_PER_BLOCK][WIN_N];
    JobSlot *const win = s_win[readfirst_u32(threadIdx.x / WAVE)];
    uint32_t win_pos = 0, win_cnt = 0;   // entries [win_pos, win_cnt) of the window wait for a lane
    uint64_t sp_lo = 0, sp_hi = 0;       // ... queue slots reserved for parked walks
    // POOL of deferred steps (DEFER): slot s = words pool[q][s], q = 0..4:
    //   { job, j, s0, d } { n_in, pp, e, coff } { kmax -> choice once settled, soff lo, soff hi, k1 }
    //   { commons before k1, shifts, p_next, staged cell 0 } { draw lo, draw hi, staged cells 1, 2 }
    // m_def / m_set (wave-uniform): slots waiting for the interval decision / settled, waiting for a free lane
    __shared__ uint4 s_pool[DEFER ? WAVES_PER_BLOCK : 1][5][POOL_N];
    __shared__ uint8_t s_map[DEFER ? WAVES_PER_BLOCK : 1][WAVE];
    uint4 (*const pool)[POOL_N] = s_pool[DEFER ? readfirst_u32(threadIdx.x / WAVE) : 0];
    constexpr uint64_t POOL_MASK = POOL_N >= 64 ? ~0ull : ((1ull << (POOL_N & 63)) - 1ull);
    uint8_t *const pmap = s_map[DEFER ? readfirst_u32(threadIdx.x / WAVE) : 0];
    uint64_t m_def = 0, m_set = 0;
    uint64_t m_chn = 0;       // CHAINS: slots whose step waits for its float chain
    bool force_pass = false, force_chain = false;
    // queue slots for `pm` parked walks of this wavefront (wave-uniform mask; rank of a lane = its order in pm): what is
    // left of the previous reservation is used up first (only a wavefront's LAST reservation leaves void slots: the queue
    // never holds more than parked walks + susp_chunk slots per wavefront -- the host sizes it for that)
    auto queue_slot = [&](uint64_t pm) -> uint64_t {
        const uint64_t np = (uint64_t)__popcll(pm);
        const uint64_t left = sp_hi - sp_lo;
        const uint64_t old_lo = sp_lo;
        uint64_t new_lo = 0;
        if (left < np) {
            const unsigned long long sc_ = LA32(susp_chunk), chunk = sc_ > np - left ? sc_ : np - left;
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(LAP(unsigned long long *, susp_count), chunk);
            new_lo = readfirst_u64(base);
            sp_lo = new_lo + (np - left);
            sp_hi = new_lo + chunk;
        } else sp_lo += np;
        const uint64_t rk = (uint64_t)__popcll(pm & lane_lt);
        return rk < left ? old_lo + rk : new_lo + (rk - left);
    };
    const uint64_t grid_lanes = (uint64_t)gridDim.x * (WAVES_PER_BLOCK * WAVE);
    // statistics: wave-uniform sums of ballots where a count of lanes is all that is needed (scalar registers), 32-bit
    // per-lane counters for the rest
    unsigned long long n_steps = 0, n_amb = 0, n_wave = 0;
    uint32_t n_dead = 0, n_probes = 0;
    // a step in flight, kept while the lane waits for the chains: draw, row total, prefix bound, out weight
    double r = 0.0;
    OutCells ob = {{0u, 0u, 0u, 0u}};   // staged output cells of the current walk
    float tot = 1.0f, wo = 1.0f;
    uint32_t kmax = 0;

    // DRAWS (PW_LANES_DRAW_LDS): the draws of a walk are consecutive doubles of the stream, but between two steps of a lane
    // every other lane of the GPU touches its own sectors and the sector is gone from L2 -- one 64-byte fetch per 8-byte
    // draw.  Instead the whole sector (8 draws) is fetched ONCE, straight into LDS (global_load_lds_dwordx4: lane t's
    // 16-byte piece c lands at s_draw[c][t], no VGPRs), when the walk's stream position enters it.
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    typedef const __attribute__((address_space(1))) void *glb_ptr_t;
    __shared__ uint4 s_draw[PW_LANES_DRAW_LDS ? WAVES_PER_BLOCK : 1][4][WAVE];
    uint4 (*const dslot)[WAVE] = s_draw[PW_LANES_DRAW_LDS ? readfirst_u32(threadIdx.x / WAVE) : 0];
    uint32_t dsec = 0xffffffffu;         // the sector (stream position / 8) this lane's slot holds
#define PW_DRAW_STAGE(di)                                                                                              \
    do {                                                                                                               \
        const uint32_t sec_ = (uint32_t)((uint64_t)(di) >> 3);                                                         \
        if (sec_ != dsec) {                                                                                            \
            const double *src_ = a.rng + (uint64_t)sec_ * 8u;                                                          \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src_ + 0), (lds_ptr_t)&dslot[0][0], 16, 0, 0);               \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src_ + 2), (lds_ptr_t)&dslot[1][0], 16, 0, 0);               \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src_ + 4), (lds_ptr_t)&dslot[2][0], 16, 0, 0);               \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src_ + 6), (lds_ptr_t)&dslot[3][0], 16, 0, 0);               \
            dsec = sec_;                                                                                               \
        }                                                                                                              \
    } while (0)
    // LINE TAILS (PW_LANES_LINE_LDS): bytes 16..63 of the edge line a step enters -- the inline list or the pivots of a longer
    // one -- go to LDS with the record's load; the next step's searches read them there (seqscan.h: ListView::tail)
    __shared__ uint4 s_tail[LINE_LDS ? WAVES_PER_BLOCK : 1][3][WAVE];
    uint4 (*const dtail)[WAVE] = s_tail[LINE_LDS ? readfirst_u32(threadIdx.x / WAVE) : 0];
    const uint32_t tail_addr = LINE_LDS ? (uint32_t)(uintptr_t)(lds_ptr_t)&dtail[0][lane] : 0xffffffffu;
#define PW_LINE_STAGE(rp)                                                                                              \
    do {                                                                                                               \
        __builtin_amdgcn_global_load_lds((glb_ptr_t)((rp) + 1), (lds_ptr_t)&dtail[0][0], 16, 0, 0);                   \
        __builtin_amdgcn_global_load_lds((glb_ptr_t)((rp) + 2), (lds_ptr_t)&dtail[1][0], 16, 0, 0);                   \
        __builtin_amdgcn_global_load_lds((glb_ptr_t)((rp) + 3), (lds_ptr_t)&dtail[2][0], 16, 0, 0);                   \
    } while (0)
    // the list view of the entry this lane's walk arrived by (staged: what lives in the line is read from LDS)
    auto lane_list = [&](uint32_t e_, uint32_t d_, uint32_t n_in_, uint32_t coff_) -> ListView {
        ListView v = edge_list(a.lines, a.clist, e_, d_, n_in_, coff_);
        if (LINE_LDS) { v.tail = tail_addr; v.inl = (d_ <= 65536u && n_in_ <= EL_INLINE) ? 1u : 0u; }
        return v;
    };
    // QUAD: one 64-byte slot per lane -- the edge line the walk entered, later the list sector its search looks into
    __shared__ uint4 s_quad[QUAD ? WAVES_PER_BLOCK : 1][4][WAVE];
    uint4 (*const qbuf)[WAVE] = s_quad[QUAD ? readfirst_u32(threadIdx.x / WAVE) : 0];
    // (round 6, second half: instruction k of a fetch serves lane k of every QUAD -- the sector a quad's lanes load is one of
    //  their own four, handed round by a DPP quad broadcast instead of a ds_bpermute trip through the LDS crossbar -- so lane
    //  4 j + k owns the 64 bytes the quad j writes in instruction k: slot 16 k + j)
    const uint32_t qslot = QUAD ? (uint32_t)(uintptr_t)(lds_ptr_t)&qbuf[0][0] + (((uint32_t)lane & 3u) * 16u + ((uint32_t)lane >> 2)) * 64u : 0u;   // LDS address of this lane's slot
    // sector `sec` (64-byte units from `base`) of every lane with `want` set -> that lane's slot.  Converged code only.
    // (0xffffffff names no sector: the lines array has fewer than 2^32 - 1 entries, the overflow array fewer sectors still)
    const uint32_t q_piece = (uint32_t)(lane & 3) * 16u;
    auto quad_issue = [&](bool want, const uint8_t *base, uint32_t sec) {
        const uint32_t ws = want ? sec : 0xffffffffu;
#define PW_QUAD_K(k, ctrl)                                                                                             \
        {                                                                                                              \
            const uint32_t osec = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ws, ctrl, 0xF, 0xF, false);          \
            if (osec != 0xffffffffu)                                                                                   \
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + (uint64_t)osec * 64u + q_piece), (lds_ptr_t)&qbuf[k][0], 16, 0, 0); \
        }
        PW_QUAD_K(0, 0x00)   // quad_perm:[0,0,0,0]
        PW_QUAD_K(1, 0x55)   // quad_perm:[1,1,1,1]
        PW_QUAD_K(2, 0xAA)   // quad_perm:[2,2,2,2]
        PW_QUAD_K(3, 0xFF)   // quad_perm:[3,3,3,3]
#undef PW_QUAD_K
    };
    // (the compiler does not order an LDS read behind the LDS-DMA that fills it: the wait is explicit)
    auto quad_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto quad_fetch = [&](bool want, const uint8_t *base, uint32_t sec) {
        if (!ballot(want)) return;
        quad_issue(want, base, sec);
        quad_wait();
    };
    bool pending = false;   // QUAD: this lane's walk has taken an edge whose line is on its way (requested at the end of the last iteration)
    auto q_u32 = [&](uint32_t off) -> uint32_t { return *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)(qslot + off); };
    auto q_u16 = [&](uint32_t off) -> uint32_t { return (uint32_t) * (const __attribute__((address_space(3))) uint16_t *)(uintptr_t)(qslot + off); };
    // (the compiler does not order an LDS read behind the LDS-DMA that fills it: the wait is explicit)
#define PW_DRAW_READ(di)                                                                                               \
    (*(const double *)((const char *)&dslot[((uint32_t)(di) & 7u) >> 1][lane] + (((uint32_t)(di) & 1u) << 3)))

#ifdef PW_LANES_WATCHDOG
    unsigned long long wd_main = 0, wd_refill = 0;
    uint32_t choice_wd = 0;
#endif
    for (;;) {
        PW_WD(1, 2000000ull, wd_main);
        if (DEFER) {
            // ---- PASS: the interval decision of every deferred step, slot s by lane s ---------------------------------
            if (m_def && (force_pass || (uint32_t)__popcll(m_def) >= (CHAINS ? 24u : DEFER_TH) || (uint32_t)__popcll(m_def | m_set | m_chn) >= (uint32_t)(POOL_N - POOL_N / 8))) {
                force_pass = false;
                LPROF_C(9, 1);
                LPROF_C(10, __popcll(m_def));
                const bool mine = (m_def >> lane) & 1ull;
                uint32_t ch = LANE_AMBIGUOUS, kmax_s = 0;
                if (mine) {
                    const uint4 p0 = pool[0][lane];
                    const uint2 p1 = *(const uint2 *)&pool[1][lane];
                    const uint4 p2 = pool[2][lane];
                    const uint4 p3 = pool[3][lane];
                    const uint2 p4 = *(const uint2 *)&pool[4][lane];
                    const float wo_s = p0.y >= 2u ? w_out : 1.0f;
                    const double r_s = __longlong_as_double((long long)(((unsigned long long)p4.y << 32) | p4.x));
                    LaneStep ls;
                    ls.tot = (float)lane_row_total(p0.w, p1.x, p1.y, wo_s, w_prev);
                    kmax_s = p2.x;
                    ls.kmax = p2.x; ls.probes = 0; ls.k1 = p2.w; ls.f = p3.x; ls.shifts = p3.y; ls.p_next = p3.z;
                    ch = lane_tight(p0.w, p1.y, r_s, wo_s, w_prev, ls);
                    if (ch != LANE_AMBIGUOUS) *(uint32_t *)&pool[2][lane] = ch;   // settled: the choice takes kmax's place
                }
                if (VERIFY) {   // keep what the float chain needs to decide this step again (lanes_verify_kernel)
                    bool rec = mine && ch != LANE_AMBIGUOUS;
                    if (rec && a.ver_mask) { const uint2 pj = *(const uint2 *)&pool[0][lane]; rec = ((pj.x + pj.y * 7919u) & a.ver_mask) == 0u; }
                    const uint64_t vm = ballot(rec);
                    if (vm) {
                        unsigned long long vb = 0;
                        if (lane == 0) vb = atomicAdd(LAP(unsigned long long *, ver_count), (unsigned long long)__popcll(vm));
                        vb = readfirst_u64(vb);
                        const uint64_t slot = vb + (uint64_t)__popcll(vm & lane_lt);
                        if (rec && slot < LA64(ver_cap)) {
                            const uint4 p0 = pool[0][lane], p1 = pool[1][lane], p4 = pool[4][lane];
                            const float wo_s = p0.y >= 2u ? w_out : 1.0f;
                            uint4 *vp = (uint4 *)(LAP(VerRec *, ver) + slot);
                            vp[0] = make_uint4(kmax_s, p1.x, p1.y, (LA32(ver_poison) && (slot & 1023u) == 0u) ? ch ^ 1u : ch);
                            vp[1] = make_uint4(p1.z, p1.w, p0.w, p0.x);
                            vp[2] = make_uint4(__float_as_uint((float)lane_row_total(p0.w, p1.x, p1.y, wo_s, w_prev)), __float_as_uint(wo_s), p4.x, p4.y);
                        }
                    }
                }
                const uint64_t settled = ballot(mine && ch != LANE_AMBIGUOUS);
                const uint64_t pm = CHAINS ? 0ull : m_def & ~settled;
                if (CHAINS) m_chn |= m_def & ~settled;   // left open: the float chain, by a chain pass of this wavefront (below)
                if (pm) {   // left open: the float chain (lanes_chain_kernel) -- the walk is parked from its slot
                    const uint64_t qs = queue_slot(pm);
                    if (mine && ch == LANE_AMBIGUOUS) {
                        const uint4 p0 = pool[0][lane], p1 = pool[1][lane], p2 = pool[2][lane], p4 = pool[4][lane];
                        const float wo_s = p0.y >= 2u ? w_out : 1.0f;
                        uint4 *qp = (uint4 *)(LAP(SuspRec *, susp) + qs);
                        qp[0] = p0;
                        qp[1] = p1;
                        qp[2] = make_uint4(p2.x, LANE_AMBIGUOUS, p2.y, p2.z);
                        qp[3] = make_uint4(__float_as_uint((float)lane_row_total(p0.w, p1.x, p1.y, wo_s, w_prev)), __float_as_uint(wo_s), p4.x, p4.y);
                        const uint32_t slot = (p0.y - 1u) & 3u;      // staged output cells: written out now
                        uint32_t *cell = a.out + (uint64_t)p0.x * W + (p0.y - slot);
                        if (slot >= 1u) cell[0] = pool[3][lane].w;
                        if (slot >= 2u) cell[1] = p4.z;
                        if (slot >= 3u) cell[2] = p4.w;
                    }
                }
                m_set |= settled;
                m_def = 0;
                wave_lds_fence();
                LPROF_T(2);
            }
            // ---- CHAINS: the float chains of the steps the interval decision left open, slot s by lane s -----------------------
            if (CHAINS && m_chn && (force_chain || (uint32_t)__popcll(m_chn) >= PW_LANES_CHAIN_TH ||
                                    (uint32_t)__popcll(m_def | m_set | m_chn) >= (uint32_t)(POOL_N - POOL_N / 8))) {
                force_chain = false;
                n_wave += (unsigned long long)__popcll(m_chn);
                {   // (round 6: the chains advance in step, tying binades are walked by the whole wavefront: lane_chain_wave)
                    const bool mine = (m_chn >> lane) & 1ull;
                    uint4 p0 = make_uint4(0u, 0u, 0u, 1u), p1 = make_uint4(0u, 0xffffffffu, 0u, 0u);
                    uint32_t kmax_s = 0u;
                    uint2 p4 = make_uint2(0u, 0u);
                    if (mine) {
                        p0 = pool[0][lane]; p1 = pool[1][lane];
                        kmax_s = *(const uint32_t *)&pool[2][lane];
                        p4 = *(const uint2 *)&pool[4][lane];
                    }
                    const float wo_s = p0.y >= 2u ? w_out : 1.0f;
                    const double r_s = __longlong_as_double((long long)(((unsigned long long)p4.y << 32) | p4.x));
                    const float x_in = 1.0f / (float)lane_row_total(p0.w, p1.x, p1.y, wo_s, w_prev);
                    uint32_t reads = 0;
                    const ListView cl_s = edge_list(a.lines, a.clist, p1.z, p0.w, p1.x, p1.w);
                    const uint32_t res = lane_chain_wave(mine, kmax_s, p1.x, p1.y, r_s, x_in, x_in * wo_s, x_in * w_prev, cl_s, cl_s.p, reads);
                    if (mine) {
                        n_probes += reads;
                        uint32_t ch = res;
                        if (res == LANE_CHAIN_END) ch = p0.w;              // never reached: the mirrored overflow read (choice == degree)
                        if (res == LANE_TIE) ch = LANE_NEEDS_WAVE;         // tie budget: the wave kernel takes the walk over at this step
                        *(uint32_t *)&pool[2][lane] = ch;
                    }
                }
                m_set |= m_chn;
                m_chn = 0;
                wave_lds_fence();
            }
            // ---- settled walks first: free lanes take them up with their step's choice known -------------------------
            const uint64_t freel = ballot(!(A.flags & F_ACTIVE));
            if (m_set && freel) {
                const uint32_t ns = (uint32_t)__popcll(m_set), nf = (uint32_t)__popcll(freel);
                const bool give = ((m_set >> lane) & 1ull) && (uint32_t)__popcll(m_set & lane_lt) < nf;
                if (give) pmap[__popcll(m_set & lane_lt)] = (uint8_t)lane;
                wave_lds_fence();
                if (!(A.flags & F_ACTIVE) && (uint32_t)__popcll(freel & lane_lt) < ns) {
                    const uint32_t sl = pmap[__popcll(freel & lane_lt)];
                    const uint4 p0 = pool[0][sl], p1 = pool[1][sl], p2 = pool[2][sl], p4 = pool[4][sl];
                    A.job = p0.x; A.j = p0.y; A.s0 = p0.z; A.d = p0.w;
                    A.n_in = p1.x; A.pp = p1.y; A.e = p1.z; A.coff = p1.w;
                    pre = p2.x;
                    A.soff = ((uint64_t)p2.z << 32) | p2.y;
                    ob.v[0] = pool[3][sl].w; ob.v[1] = p4.z; ob.v[2] = p4.w;
                    A.flags = F_ACTIVE | F_PRE;
                }
                m_set &= ~ballot(give);
                wave_lds_fence();
            }
        }
        // ---- QUAD: the lines requested when the last iteration's steps were taken have had the pool's work to arrive in ----
        if (QUAD && PW_LANES_QPIPE && ballot(pending)) {
            quad_wait();
            if (pending) {
                typedef uint32_t __attribute__((ext_vector_type(4))) lds_u4;
                typedef uint32_t __attribute__((ext_vector_type(2))) lds_u2;
                const lds_u4 t0_ = *(const __attribute__((address_space(3))) lds_u4 *)(uintptr_t)qslot;
                const lds_u2 t1_ = *(const __attribute__((address_space(3))) lds_u2 *)(uintptr_t)(qslot + 16u);
                const uint4 r0_ = make_uint4(t0_.x, t0_.y, t0_.z, t0_.w);
                const uint2 r1_ = make_uint2(t1_.x, t1_.y);
                PW_LANE_APPLY_TAIL(r0_, r1_, PW_LANES_QDRAW_EARLY != 0);
                pending = false;
            }
        }
        // ---- refill idle lanes from the job counter -------------------------------------------------------
        for (;;) {
            PW_WD(2, 2000000ull, wd_refill);
            const uint64_t need = ballot(!(A.flags & F_ACTIVE) && !exhausted);
            if (!need) break;
            // jobs are taken from a wavefront-local pool; the shared counter is touched once per PW_LANES_CHUNK jobs
            // (one atomic per refill made every wavefront queue on ONE address ~40 M times a second -- the counter's
            // L2 channel, not the walks, set the pace).  Near the end of the work the chunks shrink to what is needed.
            if ((resuming || win_pos == win_cnt) && pool_lo == pool_hi) {
                const uint64_t left = n_work > pool_hi ? n_work - pool_hi : 0;   // (as far as this wavefront knows)
                const unsigned long long chunk = left > 4ull * grid_lanes ? (unsigned long long)LA32(job_chunk)
                                                                         : (unsigned long long)__popcll(need) * (resuming ? 1ull : 2ull);
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(LAP(unsigned long long *, job_counter), chunk);
                base = readfirst_u64(base);
                pool_lo = base;
                pool_hi = base + chunk < n_work ? base + chunk : n_work;
                if (pool_lo >= n_work) { pool_lo = pool_hi = n_work; exhausted = true; continue; }
            }
            const uint32_t rank = (uint32_t)__popcll(need & lane_lt);
            if (!resuming) {
                // JOB WINDOW (round 5: compacted).  The next WIN_N jobs of the pool are fetched together -- job, start vertex, its
                // row, stream position, first draw: three dependent scattered loads, once per batch instead of once per
                // refill -- and the jobs whose start has no neighbours (52 % of an R-MAT job array) are FINISHED right
                // there, by the lane that fetched them ([start, 0, .., 0, 1]: pecanpy.py:190-193); only the others enter
                // the window.  A refill is then one LDS read, and one trip of this loop (round 4 handed the isolated starts
                // to the idle lanes one by one: three to four trips per iteration, 14 % of the kernel's time).
                if (win_pos == win_cnt) {
                    const uint64_t avail = pool_hi - pool_lo;
                    const uint32_t nb = avail < (uint64_t)WIN_N ? (uint32_t)avail : (uint32_t)WIN_N;
                    wave_lds_fence();                                      // (earlier reads of the window are over)
                    bool live = false;
                    uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = w0;
                    if ((uint32_t)lane < nb) {
                        const uint64_t widx = pool_lo + (uint64_t)lane;
                        const uint32_t *jl_ = LAP(const uint32_t *, job_list);
                        const uint32_t job = jl_ ? jl_[widx] : (uint32_t)widx;
                        const uint32_t start = LAP(const uint32_t *, starts)[job];
                        const uint4 vr = LAP(const uint4 *, vrec)[start];
                        uint32_t *row = a.out + (uint64_t)job * W;
                        row[0] = start;
                        if (vr.y == 0) {
                            row[L + 1] = 1;          // start without neighbours; cells 1..L stay 0
                            if (jl_)          // a repaired row may hold an older walk
                                for (uint32_t z = 1; z <= L; z++) row[z] = 0;
                        } else {
                            const uint64_t so = LAP(const uint64_t *, stream_off)[job] - LA64(rng_base);
                            double r0 = 0.0;
                            if (!PW_LANES_DRAW_LDS) r0 = a.rng[so];
                            w0 = make_uint4(job, start, vr.x, vr.y);
                            w1 = make_uint4((uint32_t)so, (uint32_t)(so >> 32), (uint32_t)__double_as_longlong(r0),
                                            (uint32_t)((unsigned long long)__double_as_longlong(r0) >> 32));
                            live = true;
                        }
                    }
                    const uint64_t lm = ballot(live);
                    if (live) {
                        uint4 *wp = (uint4 *)(win + __popcll(lm & lane_lt));
                        wp[0] = w0;
                        wp[1] = w1;
                    }
                    win_pos = 0;
                    win_cnt = (uint32_t)__popcll(lm);
                    pool_lo += nb;
                    wave_lds_fence();
                    continue;
                }
                const uint32_t avail_w = win_cnt - win_pos;
                const uint32_t take_w = avail_w < (uint32_t)__popcll(need) ? avail_w : (uint32_t)__popcll(need);
                if (!(A.flags & F_ACTIVE) && !exhausted && rank < take_w) {
                    const uint4 *wp = (const uint4 *)(win + (win_pos + rank));
                    const uint4 j0 = wp[0], j1 = wp[1];
                    A.job = j0.x;
                    A.soff = ((uint64_t)j1.y << 32) | j1.x;
                    A.s0 = j0.z; A.d = j0.w; A.n_in = 0; A.pp = NOT_FOUND; A.e = WEIGHTED ? j0.y : 0u; A.coff = 0; A.j = 1;
                    if (PW_LANES_DRAW_LDS) { PW_DRAW_STAGE(A.soff); }
                    else r = __longlong_as_double((long long)(((unsigned long long)j1.w << 32) | j1.z));
                    A.flags = F_ACTIVE;
                }
                win_pos += take_w;
                continue;
            }
            const uint64_t avail = pool_hi - pool_lo;
            const uint64_t pool_base = pool_lo;
            const uint64_t take = avail < (uint64_t)__popcll(need) ? avail : (uint64_t)__popcll(need);
            pool_lo += take;
            if (!(A.flags & F_ACTIVE) && !exhausted && rank < take) {
                const uint64_t widx = pool_base + rank;
                const uint4 *qp = (const uint4 *)(LAP(const SuspRec *, resume) + widx);
                const uint4 q0 = qp[0], q1 = qp[1], q2 = qp[2];
                if (q0.x != NOT_FOUND) {                        // (void: a reserved slot no walk was parked in)
                    A.job = q0.x; A.j = q0.y; A.s0 = q0.z; A.d = q0.w;
                    A.n_in = q1.x; A.pp = q1.y; A.e = q1.z; A.coff = q1.w;
                    pre = q2.y;
                    A.soff = ((uint64_t)q2.w << 32) | q2.z;
                    const uint32_t slot = (A.j - 1u) & 3u;      // staged output cells of the group in progress
                    const uint32_t *cell = a.out + (uint64_t)A.job * W + (A.j - slot);
                    if (slot >= 1u) ob.v[0] = cell[0];
                    if (slot >= 2u) ob.v[1] = cell[1];
                    if (slot >= 3u) ob.v[2] = cell[2];
                    A.flags = F_ACTIVE | F_PRE;
                }
            }
        }
        if (!ballot(A.flags & F_ACTIVE)) {
            if (DEFER && m_def) { force_pass = true; continue; }   // (nothing else can run: decide what waits)
            if (CHAINS && m_chn) { force_chain = true; continue; } // (... and run the chains that wait)
            break;                                                 // (m_set is empty: every lane was free to take from it)
        }
        LPROF_T(0);
        LPROF_C(8, 1);

        // ---- one step for every runnable lane ------------------------------------------------------------------
        // decision -> interval decision -> (still open:) the walk is parked for lanes_chain_kernel, or -- in-place form --
        // the lane WAITS (keeps its draw and prefix bound) while the others go on stepping, and the waiting lanes run
        // their chains together once a few have gathered or nothing else can run.
        uint32_t choice = LANE_AMBIGUOUS;
        const bool runnable = A.flags == F_ACTIVE;
        if (PW_LANES_DRAW_LDS || LINE_LDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (what was requested when the last steps were applied has landed in LDS)
            if (PW_LANES_DRAW_LDS && runnable) r = PW_DRAW_READ(A.soff + (A.j - 1));
        }
        if (FLOATS) {
            if (runnable && A.n_in != 0u && !edge_list_stored(A.d, A.n_in, A.coff)) choice = LANE_NEEDS_WAVE;   // (partial index)
            else if (runnable && a.tot_e) {
                // Round 5: the row total of every arriving entry (the reference's sequential float32 w.sum(), sparse_rw.py:89) was
                // computed once per (p, q) by the chain itself (unit_tot_kernel), and the step is decided from the REAL prefix
                // sums of the three row values -- closed form: n_in(k) + n_out(k) fl32(1/q) + [pp <= k] fl32(1/p) -- with a
                // rigorous bound on the float32 chain's drift (seqscan.h: lane_decide_unit_bounded): one list search instead of
                // two chains of ~17 binades x a search each.  What the bound leaves open waits (F_WAIT2) until a few lanes have
                // gathered and is decided by the chain over w / tot, as before.
                wo = A.j >= 2 ? w_out : 1.0f;   // first step of a walk: no bias (sparse_rw.py:66), total = degree exactly
                const float t = A.j >= 2 ? a.tot_e[A.e] : (float)A.d;
                if (!(t > 0.0f)) choice = LANE_NEEDS_WAVE;          // (NaN: the total's chain met a tie binade beyond the budget)
                else {
                    uint32_t probes_ = 0, ks_ = 0;
                    BoundedAmb amb_;
                    choice = lane_decide_unit_bounded(A.d, A.n_in, A.pp, r, t, wo, w_prev, lane_list(A.e, A.d, A.n_in, A.coff), probes_, ks_, &amb_);
                    n_probes += probes_;
                    if (choice == LANE_REDO) choice = LANE_NEEDS_WAVE;
                    if (choice == LANE_AMBIGUOUS) {
                        // (k_safe == 0: nothing is known below the draw -- straight to the chain)
                        A.flags = F_ACTIVE | (ks_ ? F_WAIT2 : F_WAIT3);
                        tot = t; tk1 = ks_; ti1 = amb_.f; tpn = amb_.p_next;
                        tz = (float)amb_.z_abs * 1.000001f + 1e-37f;       // (rounded UP: a bound)
                    }
                }
            } else if (runnable) {
                wo = A.j >= 2 ? w_out : 1.0f;   // first step of a walk: no bias (sparse_rw.py:66)
                const ListView cl = lane_list(A.e, A.d, A.n_in, A.coff);
                float xi = 1.0f, xo = wo, xp = w_prev, rowsum = 0.0f;
                double target = __longlong_as_double(0x7ff0000000000000ll);   // +inf: the chain runs to the end of the row
                uint32_t res = LANE_CHAIN_END;
                for (int phase = 0; phase < 2 && res == LANE_CHAIN_END; phase++) {   // (one inlined copy of the chain)
                    uint32_t reads = 0;
                    res = lane_chain<true>(A.d, A.n_in, A.pp, target, xi, xo, xp, cl, reads, &rowsum);
                    n_probes += reads;
                    if (phase == 0 && res == LANE_CHAIN_END) {   // tot known: the normalised values, float32 / float32
                        xi = 1.0f / rowsum; xo = wo / rowsum; xp = w_prev / rowsum;
                        target = r;
                        continue;
                    }
                    break;
                }
                // LANE_CHAIN_END after the search: the CDF never reached r (mirrored overflow read); LANE_TIE: a tie binade
                // beyond the budget -- both: walk_kernel takes the walk over at this step
                choice = res < A.d ? res : (res == LANE_CHAIN_END ? A.d : LANE_NEEDS_WAVE);
            }
            if (a.tot_e) n_amb += (unsigned long long)__popcll(ballot(runnable && (A.flags & (F_WAIT2 | F_WAIT3)) != 0));   // (left open just now)
            {   // Round 6: the steps the bound left open go through the INTERVAL DECISION first (lane_tight_values: the float32 chain's
                // systematic drift bounded from the class counts the bounded decision already has -- arithmetic only), once enough
                // lanes wait or nothing else can run; nine in ten are settled there (RMAT-20 rows, uniform draws: 6.6 % open after
                // the bound, 0.46 % after this), the rest wait on for the chain.
                const uint64_t wt = ballot((A.flags & F_WAIT2) != 0);
                if (wt != 0 && ((uint32_t)__popcll(wt) >= PW_LANES_FTIGHT || ballot(A.flags == F_ACTIVE && choice != LANE_AMBIGUOUS) == 0)) {
                    bool rec = false;
                    if (A.flags & F_WAIT2) {
                        const uint32_t res = lane_tight_values(A.d, A.pp, r, 1.0f / tot, wo / tot, w_prev / tot, tk1, ti1, tpn, (double)tz);
                        if (res != LANE_AMBIGUOUS) { choice = res; A.flags = F_ACTIVE; rec = ((A.job + A.j * 7919u) & a.ver_mask) == 0u; }
                        else A.flags = F_ACTIVE | F_WAIT3;
                    }
                    if (VERIFY) {   // keep what the float chain needs to decide this step again (lanes_verify_kernel, floats = 1)
                        const uint64_t vm = ballot(rec);
                        if (vm) {
                            unsigned long long vb = 0;
                            if (lane == 0) vb = atomicAdd(LAP(unsigned long long *, ver_count), (unsigned long long)__popcll(vm));
                            vb = readfirst_u64(vb);
                            const uint64_t slot = vb + (uint64_t)__popcll(vm & lane_lt);
                            if (rec && slot < LA64(ver_cap)) {
                                uint4 *vp = (uint4 *)(LAP(VerRec *, ver) + slot);
                                vp[0] = make_uint4(A.d, A.n_in, A.pp, (LA32(ver_poison) && (slot & 1023u) == 0u) ? choice ^ 1u : choice);
                                vp[1] = make_uint4(A.e, A.coff, A.d, A.job);
                                vp[2] = make_uint4(__float_as_uint(tot), __float_as_uint(wo), (uint32_t)__double_as_longlong(r),
                                                   (uint32_t)((unsigned long long)__double_as_longlong(r) >> 32));
                            }
                        }
                    }
                }
            }
            {   // what is still open: the float chains together, once enough lanes wait or nothing else can run
                const uint64_t w2 = ballot((A.flags & F_WAIT3) != 0);
                if (w2 != 0 && ((uint32_t)__popcll(w2) >= PW_LANES_FWAIT ||
                                ballot((A.flags == F_ACTIVE && choice != LANE_AMBIGUOUS) || (A.flags & F_WAIT2) != 0) == 0)) {
                    n_wave += (unsigned long long)__popcll(w2);
                    if (A.flags & F_WAIT3) {
                        uint32_t reads = 0;
                        const uint32_t res = lane_chain<true>(A.d, A.n_in, A.pp, r, 1.0f / tot, wo / tot, w_prev / tot,
                                                              lane_list(A.e, A.d, A.n_in, A.coff), reads);
                        n_probes += reads;
                        choice = res < A.d ? res : (res == LANE_CHAIN_END ? A.d : LANE_NEEDS_WAVE);
                        A.flags = F_ACTIVE;
                    }
                }
            }
        } else {
        LaneStep ls{1.0f, 0u, 0u, 0u, 0u, 0u, 0u};
        // partial index: the list of the entry this walk arrived by was left out -- the step is parked for
        // lanes_eager_kernel (queueing form) or handed to walk_kernel with the rest of the walk (no queue)
        const bool nolist = runnable && A.n_in != 0u && !edge_list_stored(A.d, A.n_in, A.coff);
        uint32_t wk_safe = 0;   // WEIGHTED: every partial sum before this element stays below the draw (parked with the step)
        if (WEIGHTED) {
            if (runnable) {
                // (first step of a walk: no prev, plain weights -- the eager kernel has the per-vertex normaliser; arrivals by
                //  an overflow line or an entry without tables likewise)
                unsigned long long wo_ = ~0ull;
                if (A.j >= 2u && A.e < a.nnz && !nolist) wo_ = a.wl_off[A.e];
                if (A.j == 1u && a.wp1) {
                    // first step: no prev, the raw weights over their sequential float32 total (sparse_rw.py:66-67 / 89) -- the
                    // same decision on the prefix sums of the raw row (A.e holds the start vertex until the step is applied)
                    const WeightedRow wr{a.wp1 + A.s0, nullptr, 0.0, true};
                    uint32_t probes_ = 0;
                    choice = lane_decide_weighted(A.d, 0u, NOT_FOUND, r, a.tot_v[A.e], wr, ListView{nullptr, 0u}, probes_, wk_safe);
                    n_probes += probes_;
                    if (choice == LANE_REDO) choice = LANE_AMBIGUOUS;
                }
                if (wo_ != ~0ull) {
                    const WeightedRow wr{a.wpq + A.s0, a.wdl + wo_, a.wl_dprev[A.e], a.wl_pos != 0u};
                    uint32_t probes_ = 0;
                    choice = lane_decide_weighted(A.d, A.n_in, A.pp, r, a.tot_e[A.e], wr, lane_list(A.e, A.d, A.n_in, A.coff), probes_, wk_safe);
                    n_probes += probes_;
                    if (choice == LANE_REDO) choice = LANE_AMBIGUOUS;
                }
                if (choice == LANE_AMBIGUOUS) {
                    wo = w_out;
                    ls.kmax = LANE_EAGER_MARK;                               // parked for lanes_eager_kernel<true, ..>
                    if (INPLACE && !a.susp) choice = LANE_NEEDS_WAVE;       // no queue: walk_kernel takes the walk over here
                }
            }
        } else if (QUAD) {
            // lane_decide with the search run here: the line the walk entered waits in this lane's LDS slot (record, inline list
            // or pivots); an overflow list is searched SECTOR by sector -- every lane that still needs one names it, the quads
            // fetch them all (one request per sector), and the lane looks at the sector's first and last entry of its range:
            // the answer lies inside (a bisection in LDS finishes it) or the range shrinks to one side.  The sector is chosen
            // by INTERPOLATION between the masses at the two ends of the range (the pivots give them; the masses grow almost
            // linearly with the index): lists of thousands of entries are settled in one or two sectors where the bisection
            // of rounds 2-4 took log2(n / 21 / 32) + 1 dependent trips to memory and five more probes inside the last sector.
            DecideCtx dc;
            SearchResult sr;
            sr.f = 0; sr.p_below = 0; sr.v_below = 0; sr.has_below = false; sr.p_at = 0xffffffffu; sr.v_at = 0;
            uint32_t s_lo = 0, s_hi = 0;
            bool need = false, go = false;
            const bool narrow = A.d <= 65536u;
            if (runnable && !nolist) {
                wo = A.j >= 2 ? w_out : 1.0f;   // first step of a walk: no bias (sparse_rw.py:66)
                // (r = this step's draw: loaded when the previous step was applied / the walk was started)
                choice = lane_decide_begin(A.d, A.n_in, A.pp, r, wo, w_prev, ls, dc);
                go = choice != LANE_REDO;
                if (go && A.n_in) {
                    // ONE bisection over what the line holds -- the list itself (n_in <= 20 entries of a narrow row) or the pivots of a
                    // longer one, entry (t + 1) * step at slot t -- for all lanes together (two loops, one per kind, cost the
                    // wavefront twice the instructions); every mass is below 2^24 units: 32-bit arithmetic
                    const MassEval ev64{A.pp, dc.sh_in, dc.sh_out, dc.sh_prev};
                    const uint32_t wide = narrow ? 0u : 1u;
                    const bool inl = narrow && A.n_in <= EL_INLINE;
                    const uint32_t step = inl ? 1u : list_pivot_step(wide, A.n_in);
                    uint32_t klo = 0, khi = inl ? A.n_in : (list_has_pivots(wide, A.n_in) ? list_pivot_count(wide) : 0u);
                    s_hi = A.n_in;
                    while (klo < khi) {
                        const uint32_t km = (klo + khi) >> 1;
                        const uint32_t ik = inl ? km : (km + 1u) * step;
                        const uint32_t P = narrow ? q_u16(24u + (km << 1)) : q_u32(24u + (km << 2));
                        const uint32_t vm = (uint32_t)ev64(ik, P);
                        if (vm >= dc.lo_th) { khi = km; s_hi = ik; sr.p_at = P; sr.v_at = vm; }
                        else { klo = km + 1u; s_lo = ik + 1u; sr.p_below = P; sr.v_below = vm; sr.has_below = true; }
                    }
                    need = !inl && s_lo < s_hi;
                    sr.f = s_lo;
                }
            }
            for (uint32_t trip = 0; ballot(need); trip++) {
                const uint32_t esh = narrow ? 1u : 2u;
                const uint64_t lbase = (uint64_t)A.coff * 16u;               // the list's byte offset in the overflow array
                uint32_t g = s_lo + ((s_hi - s_lo) >> 1);
                if (need && trip < 2u && sr.has_below && sr.p_at != 0xffffffffu && sr.v_at > sr.v_below) {
                    // entry s_lo - 1 weighs v_below < target, entry s_hi weighs v_at >= target: where a straight line crosses
                    // (every mass of a row lane_decide_begin accepts is below 2^24 units: 32-bit arithmetic)
                    const float t = (float)(dc.lo_th - (uint32_t)sr.v_below) / (float)((uint32_t)sr.v_at - (uint32_t)sr.v_below);
                    const float fe = (float)s_lo - 1.0f + t * (float)(s_hi - s_lo + 1u);
                    const uint32_t gi = fe > 0.0f ? (uint32_t)fe : 0u;
                    g = gi < s_lo ? s_lo : (gi >= s_hi ? s_hi - 1u : gi);
                }
                const uint32_t sec = (uint32_t)((lbase + ((uint64_t)g << esh)) >> 6);
                quad_fetch(need, a.clist, sec);
                if (need) {
                    const MassEval ev64{A.pp, dc.sh_in, dc.sh_out, dc.sh_prev};
                    auto ev = [&](uint32_t i, uint32_t P) -> uint32_t { return (uint32_t)ev64(i, P); };   // (below 2^24 units, as above)
                    const uint64_t sb = (uint64_t)sec << 6;
                    const uint32_t w_lo = sb > lbase ? (uint32_t)((sb - lbase) >> esh) : 0u;
                    uint32_t w_hi = (uint32_t)((sb + 64u - lbase) >> esh);
                    if (w_hi > A.n_in) w_hi = A.n_in;
                    const uint32_t c_lo = w_lo > s_lo ? w_lo : s_lo, c_hi = w_hi < s_hi ? w_hi : s_hi;   // (g lies in both: not empty)
                    const uint32_t eoff = (uint32_t)(lbase - sb);                                    // (mod 2^32: may be "negative")
                    auto ent = [&](uint32_t i) -> uint32_t { const uint32_t o = eoff + (i << esh); return narrow ? q_u16(o) : q_u32(o); };
                    const uint32_t P0 = ent(c_lo);
                    const uint32_t v0 = ev(c_lo, P0);
                    if (v0 >= dc.lo_th) { s_hi = c_lo; sr.p_at = P0; sr.v_at = v0; }
                    else {
                        sr.p_below = P0; sr.v_below = v0; sr.has_below = true;
                        s_lo = c_lo + 1u;
                        if (s_lo < c_hi) {
                            const uint32_t P1 = ent(c_hi - 1u);
                            const uint32_t v1 = ev(c_hi - 1u, P1);
                            if (v1 < dc.lo_th) { s_lo = c_hi; sr.p_below = P1; sr.v_below = v1; }
                            else {                                   // inside this sector: a bisection in LDS
                                uint32_t l2 = s_lo, h2 = c_hi - 1u;
                                sr.p_at = P1; sr.v_at = v1;
                                while (l2 < h2) {
                                    const uint32_t mid = (l2 + h2) >> 1;
                                    const uint32_t P = ent(mid);
                                    const uint32_t vm = ev(mid, P);
                                    if (vm >= dc.lo_th) { h2 = mid; sr.p_at = P; sr.v_at = vm; }
                                    else { l2 = mid + 1u; sr.p_below = P; sr.v_below = vm; }
                                }
                                s_lo = s_hi = l2;
                            }
                        }
                    }
                    ls.probes += 32u;                             // (one 64-byte sector = 32 two-byte entries' worth of list bytes)
                    sr.f = s_lo;
                    need = s_lo < s_hi;
                }
            }
            if (go) choice = lane_decide_end(A.d, A.n_in, A.pp, dc, &sr, ls);
            n_probes += ls.probes;
        } else
        if (runnable && !nolist) {
            wo = A.j >= 2 ? w_out : 1.0f;   // first step of a walk: no bias (sparse_rw.py:66)
            // (r = this step's draw: loaded when the previous step was applied / the walk was started)
            choice = lane_decide(A.d, A.n_in, A.pp, r, wo, w_prev, lane_list(A.e, A.d, A.n_in, A.coff), ls);
            n_probes += ls.probes;
        }
        if (nolist && !WEIGHTED) {
            wo = w_out;
            ls.kmax = LANE_EAGER_MARK;
            if (INPLACE && !a.susp) choice = LANE_NEEDS_WAVE;
        }
#ifdef PW_PROF_LANES
        {   // [11] sum over iterations of the DEEPEST list search of the wavefront, [12] lanes whose list lives in the overflow
            // array, [13] runnable lanes, [14] sum of all probes
            uint32_t pm_ = runnable ? ls.probes : 0u, ps_ = pm_;
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o_ = (uint32_t)__shfl_down((int)pm_, (unsigned)off, WAVE);
                pm_ = o_ > pm_ ? o_ : pm_;
                ps_ += (uint32_t)__shfl_down((int)ps_, (unsigned)off, WAVE);
            }
            LPROF_C(11, pm_);
            LPROF_C(14, ps_);
            const int n_over_ = __popcll(ballot(runnable && A.n_in != 0u && !(A.d <= 65536u && A.n_in <= EL_INLINE)));
            const int n_run_ = __popcll(ballot(runnable));
            LPROF_C(12, n_over_);
            LPROF_C(13, n_run_);
        }
#endif
        LPROF_T(1);
        // the chain's drift bounded from the class counts (seqscan.h: lane_tight): arithmetic only, right away
        const bool amb0 = !WEIGHTED && runnable && choice == LANE_AMBIGUOUS && !nolist;
        if (DEFER) {
            // the step waits in the pool for the next pass of the interval decision; this lane takes another walk
            const uint64_t am = ballot(amb0);
            if (am) {
                n_amb += (unsigned long long)__popcll(am);
                const uint64_t freem = ~(m_def | m_set | m_chn) & POOL_MASK;
                const uint32_t nfree = (uint32_t)__popcll(freem), nd = (uint32_t)__popcll(am);
                const bool taken = ((freem >> lane) & 1ull) && (uint32_t)__popcll(freem & lane_lt) < nd;   // (slot `lane`)
                if (taken) pmap[__popcll(freem & lane_lt)] = (uint8_t)lane;
                wave_lds_fence();
                if (amb0 && (uint32_t)__popcll(am & lane_lt) < nfree) {
                    const uint32_t sl = pmap[__popcll(am & lane_lt)];
                    pool[0][sl] = make_uint4(A.job, A.j, A.s0, A.d);
                    pool[1][sl] = make_uint4(A.n_in, A.pp, A.e, A.coff);
                    pool[2][sl] = make_uint4(ls.kmax, (uint32_t)A.soff, (uint32_t)(A.soff >> 32), ls.k1);
                    pool[3][sl] = make_uint4(ls.f, ls.shifts, ls.p_next, ob.v[0]);
                    pool[4][sl] = make_uint4((uint32_t)__double_as_longlong(r), (uint32_t)((unsigned long long)__double_as_longlong(r) >> 32),
                                             ob.v[1], ob.v[2]);
                    A.flags = 0;
                }
                m_def |= ballot(taken);
                wave_lds_fence();
            }
        } else if (ballot(amb0)) {
            n_amb += (unsigned long long)__popcll(ballot(amb0));
            LPROF_C(9, 1);
            LPROF_C(10, __popcll(ballot(amb0)));
            if (amb0) choice = lane_tight(A.d, A.pp, r, wo, w_prev, ls);
            if (VERIFY) {
                // keep what the float chain needs to decide this step again (lanes_verify_kernel)
                const bool rec = amb0 && choice != LANE_AMBIGUOUS && ((A.job + A.j * 7919u) & a.ver_mask) == 0u;
                const uint64_t vm = ballot(rec);
                if (vm) {
                    unsigned long long vb = 0;
                    if (lane == 0) vb = atomicAdd(LAP(unsigned long long *, ver_count), (unsigned long long)__popcll(vm));
                    vb = readfirst_u64(vb);
                    const uint64_t slot = vb + (uint64_t)__popcll(vm & lane_lt);
                    if (rec && slot < LA64(ver_cap)) {
                        uint4 *vp = (uint4 *)(LAP(VerRec *, ver) + slot);
                        vp[0] = make_uint4(ls.kmax, A.n_in, A.pp, (LA32(ver_poison) && (slot & 1023u) == 0u) ? choice ^ 1u : choice);
                        vp[1] = make_uint4(A.e, A.coff, A.d, A.job);
                        vp[2] = make_uint4(__float_as_uint(ls.tot), __float_as_uint(wo), (uint32_t)__double_as_longlong(r),
                                           (uint32_t)((unsigned long long)__double_as_longlong(r) >> 32));
                    }
                }
            }
            LPROF_T(2);
        }
        if (!INPLACE || a.susp) {
            // park the walk: the chain runs later, at full width (lanes_chain_kernel); this lane takes another walk
            // (DEFER: only the steps that found no free pool slot arrive here, undecided -- the float chain settles them)
            const bool park = runnable && choice == LANE_AMBIGUOUS && A.flags == F_ACTIVE;
            const uint64_t pm = ballot(park);
            if (pm) {
                const uint64_t qs = queue_slot(pm);   // (queue slots come from a wavefront-local reservation too)
                if (park) {
                    uint4 *qp = (uint4 *)(LAP(SuspRec *, susp) + qs);
                    qp[0] = make_uint4(A.job, A.j, A.s0, A.d);
                    qp[1] = make_uint4(A.n_in, A.pp, A.e, A.coff);
                    qp[2] = make_uint4(ls.kmax, LANE_AMBIGUOUS, (uint32_t)A.soff, (uint32_t)(A.soff >> 32));
                    qp[3] = make_uint4(WEIGHTED ? wk_safe : __float_as_uint(ls.tot), __float_as_uint(wo), (uint32_t)__double_as_longlong(r),
                                       (uint32_t)((unsigned long long)__double_as_longlong(r) >> 32));
                    const uint32_t slot = (A.j - 1u) & 3u;          // staged output cells: written out now
                    uint32_t *cell = a.out + (uint64_t)A.job * W + (A.j - slot);
                    if (slot >= 1u) cell[0] = ob.v[0];
                    if (slot >= 2u) cell[1] = ob.v[1];
                    if (slot >= 3u) cell[2] = ob.v[2];
                    A.flags = 0;
                }
            }
        } else if (INPLACE && runnable && choice == LANE_AMBIGUOUS) {
            A.flags = F_ACTIVE | F_WAIT2; tot = ls.tot; kmax = ls.kmax;
        }
        if (A.flags == (F_ACTIVE | F_PRE)) { choice = pre; A.flags = F_ACTIVE; }   // resumed walk: its parked step
        if (INPLACE) {
            const uint64_t w2 = ballot((A.flags & F_WAIT2) != 0);
            if (w2 != 0 && ((uint32_t)__popcll(w2) >= PW_LANES_WAIT2 || ballot(A.flags == F_ACTIVE && choice != LANE_AMBIGUOUS) == 0)) {
                n_wave += (unsigned long long)__popcll(w2);                 // (steps decided by the float chain)
                LPROF_C(5, 1);
                LPROF_C(6, __popcll(w2));
                if (A.flags & F_WAIT2) {
                    // the float32 chain over the first kmax positions, by this lane alone (seqscan.h: lane_chain)
                    const float x_in = 1.0f / tot;
                    uint32_t reads = 0;
                    const uint32_t res = lane_chain(kmax, A.n_in, A.pp, r, x_in, x_in * wo, x_in * w_prev,
                                                    lane_list(A.e, A.d, A.n_in, A.coff), reads);
                    n_probes += reads;
                    choice = res;
                    if (res == LANE_CHAIN_END) choice = A.d;                // never reached: the mirrored overflow read
                    if (res == LANE_TIE) choice = LANE_NEEDS_WAVE;          // tie binade too long for one lane -> redo
                    A.flags = F_ACTIVE;
                }
                LPROF_T(4);
            }
        }
        }
        n_steps += (unsigned long long)__popcll(ballot(A.flags == F_ACTIVE && choice < A.d));   // (LANE_* codes are >= any degree)
        if (QUAD) {
            // the lines of all lanes that move on: ONE request each, by the quads, into the lanes' slots -- and the next step's
            // draw with them.  Nothing waits here: the record is applied at the top of the next iteration, behind the pool's
            // pass (PW_LANE_APPLY_TAIL above), so the round trip runs under the interval decisions of the deferred steps.
            bool fetch = false;
            if (A.flags == F_ACTIVE && choice != LANE_AMBIGUOUS) PW_LANE_APPLY_HEAD(fetch);
            if (PW_LANES_QDRAW_EARLY && fetch && A.j < L) r = a.rng[A.soff + A.j];   // (step A.j + 1 samples with double #(soff + A.j); unused if the walk ends)
            quad_issue(fetch, (const uint8_t *)a.lines, A.e);
            pending = fetch;
            if (!PW_LANES_QPIPE && ballot(pending)) {
                quad_wait();
                if (pending) {
                    typedef uint32_t __attribute__((ext_vector_type(4))) lds_u4;
                    typedef uint32_t __attribute__((ext_vector_type(2))) lds_u2;
                    const lds_u4 t0_ = *(const __attribute__((address_space(3))) lds_u4 *)(uintptr_t)qslot;
                    const lds_u2 t1_ = *(const __attribute__((address_space(3))) lds_u2 *)(uintptr_t)(qslot + 16u);
                    const uint4 r0_ = make_uint4(t0_.x, t0_.y, t0_.z, t0_.w);
                    const uint2 r1_ = make_uint2(t1_.x, t1_.y);
                    PW_LANE_APPLY_TAIL(r0_, r1_, PW_LANES_QDRAW_EARLY != 0);
                    pending = false;
                }
            }
        } else if (A.flags == F_ACTIVE && choice != LANE_AMBIGUOUS) {
            bool fetch = false;
            PW_LANE_APPLY_HEAD(fetch);
            if (fetch) {
                const uint4 *rp_ = (const uint4 *)(a.lines + A.e);
                const uint4 r0_ = rp_[0];
                const uint2 r1_ = *(const uint2 *)(rp_ + 1);
                if (LINE_LDS) { PW_LINE_STAGE(rp_); }
                PW_LANE_APPLY_TAIL(r0_, r1_, false);
            }
        }
#ifdef PW_LANES_WATCHDOG
        choice_wd = choice;
#endif
        LPROF_T(3);
    }
#undef PW_LANE_APPLY_HEAD
#undef PW_LANE_APPLY_TAIL
#undef PW_DRAW_STAGE
#undef PW_DRAW_READ
#undef PW_LINE_STAGE
    if (LA64(susp))
        for (uint64_t v = sp_lo + (uint64_t)lane; v < sp_hi; v += WAVE) LAP(SuspRec *, susp)[v].job = NOT_FOUND;   // reserved, unused
#ifdef PW_PROF_LANES
    if (lane == 0) for (int i = 0; i < 16; i++) if (lp[i]) atomicAdd(&g_lprof[i], lp[i]);
#endif
    // wave totals
    unsigned long long dead_w = n_dead, probes_w = n_probes;
    for (int off = 32; off > 0; off >>= 1) {
        dead_w += (unsigned long long)__shfl_down((long long)dead_w, (unsigned)off, WAVE);
        probes_w += (unsigned long long)__shfl_down((long long)probes_w, (unsigned)off, WAVE);
    }
    if (lane == 0) {
        if (n_steps) atomicAdd(LAP(unsigned long long *, stats) + 0, n_steps);
        if (dead_w) atomicAdd(LAP(unsigned long long *, stats) + 3, dead_w);
        if (probes_w) atomicAdd(LAP(unsigned long long *, stats) + 6, probes_w);
        if (n_amb) atomicAdd(LAP(unsigned long long *, stats) + 7, n_amb);
        if (n_wave) atomicAdd(LAP(unsigned long long *, stats) + 9, n_wave);
    }
}
#undef LA64
#undef LA32
#undef LAP

// ---- the float32 chains of a whole queue of parked walks, one lane each, every lane busy ------------------------------
#ifndef PW_CHAIN_TAILS
#define PW_CHAIN_TAILS 1   // the chain's ~17 searches bisect the same pivots (or the same inline list): bytes 16..63 of the entry's
                           // line are staged in LDS once per chain (seqscan.h: ListView::tail).  RMAT-22 pass 137.6 -> 134.3 ms
#endif
#ifndef PW_CHAIN_WAVES
#define PW_CHAIN_WAVES 5   // 88 VGPRs; six waves (80) spill 44 bytes and gain 0.5 %
#endif
#ifndef PW_CHAIN_SORT
#define PW_CHAIN_SORT 1    // the 256 records of a workgroup are dealt to its lanes in the order of their prefix bound kmax (the chain's
                           // length): a wavefront lasts as long as its longest chain, so chains of similar length share one
#endif
__global__ void __launch_bounds__(256, PW_CHAIN_WAVES)
lanes_chain_kernel(SuspRec *q, uint64_t n, const ELine *__restrict__ lines, const uint8_t *__restrict__ clist, float w_prev,
                   unsigned long long *stats) {
    using B = Binade<float>;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (PW_CHAIN_SORT) {
        __shared__ uint32_t s_key[256];
        __shared__ uint16_t s_ord[256];
        uint32_t key = 0u;                                  // (void slots / steps for lanes_eager_kernel: first, they cost nothing)
        if (i < n && q[i].job != NOT_FOUND && q[i].kmax != LANE_EAGER_MARK) key = q[i].kmax + 1u;
        s_key[threadIdx.x] = key;
        __syncthreads();
        uint32_t rank = 0;
        for (uint32_t j = 0; j < 256u; j++) {
            const uint32_t kj = s_key[j];
            rank += (kj < key || (kj == key && j < threadIdx.x)) ? 1u : 0u;
        }
        s_ord[rank] = (uint16_t)threadIdx.x;
        __syncthreads();
        i = (uint64_t)blockIdx.x * 256 + s_ord[threadIdx.x];
    }
    unsigned long long reads_l = 0, done = 0;
    typedef __attribute__((address_space(3))) void *lds_ptr_t;
    typedef const __attribute__((address_space(1))) void *glb_ptr_t;
    __shared__ uint4 s_tail[PW_CHAIN_TAILS ? 4 : 1][3][WAVE];
    uint4 (*const dtail)[WAVE] = s_tail[PW_CHAIN_TAILS ? readfirst_u32(threadIdx.x / WAVE) : 0];
    uint4 q0 = make_uint4(NOT_FOUND, 0u, 0u, 0u), q1 = make_uint4(0u, 0u, 0u, 0u), q2 = make_uint4(0u, 0u, 0u, 0u), q3 = make_uint4(0u, 0u, 0u, 0u);
    if (i < n) {
        const uint4 *qp = (const uint4 *)(q + i);
        q0 = qp[0]; q1 = qp[1]; q2 = qp[2]; q3 = qp[3];
    }
    const bool active = i < n && q0.x != NOT_FOUND && q2.x != LANE_EAGER_MARK;   // (not: void slot / a step for lanes_eager_kernel)
    const float tot = __uint_as_float(q3.x), wo = __uint_as_float(q3.y);
    const double r = __longlong_as_double((long long)(((unsigned long long)q3.w << 32) | q3.z));
    const float x_in = 1.0f / tot, x_out = x_in * wo, x_prev = x_in * w_prev;
    ListView cl = edge_list(lines, clist, q1.z, q0.w, q1.x, q1.w);
    const void *const list_p = cl.p;                       // (global memory: what the cooperative walk of a tying binade reads)
    if (active && PW_CHAIN_TAILS && q1.x != 0u) {   // bytes 16..63 of the entry's line -> LDS (the inline list or the pivots)
        const uint4 *rp = (const uint4 *)(lines + q1.z);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(rp + 1), (lds_ptr_t)&dtail[0][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(rp + 2), (lds_ptr_t)&dtail[1][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(rp + 3), (lds_ptr_t)&dtail[2][0], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cl.tail = (uint32_t)(uintptr_t)(lds_ptr_t)&dtail[0][threadIdx.x & (WAVE - 1)];
        cl.inl = (q0.w <= 65536u && q1.x <= EL_INLINE) ? 1u : 0u;
    }
    uint32_t reads_c = 0;
    const uint32_t res = lane_chain_wave(active, q2.x, q1.x, q1.y, r, x_in, x_out, x_prev, cl, list_p, reads_c);
    reads_l += reads_c;
    if (active) {
        uint32_t choice = res;
        if (res == LANE_CHAIN_END) choice = q0.w;          // never reached: the mirrored overflow read (choice == degree)
        if (res == LANE_TIE) choice = LANE_NEEDS_WAVE;     // tie budget: the wave kernel redoes the walk
        q[i].choice = choice;
        done = 1;
    }
    for (int off = 32; off > 0; off >>= 1) {
        reads_l += (unsigned long long)__shfl_down((long long)reads_l, (unsigned)off, WAVE);
        done += (unsigned long long)__shfl_down((long long)done, (unsigned)off, WAVE);
    }
    if (lane_id() == 0 && done) {
        atomicAdd(stats + 6, reads_l);
        atomicAdd(stats + 9, done);
    }
}

// ---- FLOATS form: the row total of every arriving entry, once per (p, q) --------------------------------------------------
// tot[e] = the reference's sequential float32 sum of the biased row of lines[e].nxt for a walker that arrived by entry e
// (w.sum(), sparse_rw.py:89: common neighbours 1, prev fl32(1/p), the others fl32(1/q)) -- lane_chain<true> with r = +inf,
// the first of the two chains the FLOATS step used to run every time.  NaN: the chain declined (a rounding-tie binade
// beyond the budget, or the entry's list is not in a partial index): steps arriving by that entry go to walk_kernel.
// One lane per line (CSR entries, then the overflow lines of the vertices).
__global__ void __launch_bounds__(256)
unit_tot_kernel(const ELine *__restrict__ lines, const uint8_t *__restrict__ clist, uint64_t n_lines, float w_out, float w_prev, float *tot) {
    const uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_lines) return;
    const uint4 r0 = *(const uint4 *)(lines + e);
    const uint32_t coff = *((const uint32_t *)(lines + e) + 5);
    const uint32_t n_in = r0.y, pp = r0.z, d = r0.w;
    float t = __uint_as_float(0x7fc00000u);
    if (r0.x == NOT_FOUND || d == 0u) t = 0.0f;                       // (an overflow line without a target / a dead end: never stepped from)
    else if (n_in == 0u || edge_list_stored(d, n_in, coff)) {
        uint32_t reads = 0;
        float rowsum = 0.0f;
        const uint32_t res = lane_chain<true>(d, n_in, pp, __longlong_as_double(0x7ff0000000000000ll), 1.0f, w_out, w_prev,
                                              edge_list(lines, clist, (uint32_t)e, d, n_in, coff), reads, &rowsum);
        if (res == LANE_CHAIN_END) t = rowsum;
    }
    tot[e] = t;
}

// ---- steps whose entry's list is not in the (partial) index: walk_kernel's eager step, one wavefront per parked record ----
// The record names the entry e = (prev -> cur) the walk arrived by and the step's draw; membership of cur's row in prev's
// is established the way walk_kernel does without lists (key stream -> filter -> adjacency index: sample_step_unit with
// step_edge = none), the float32 chain decides, and the record's `choice` is what the next lane round applies.
// n, q: kernel arguments behind WalkArgs (read from the kernarg segment like walk_kernel's)
__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE, PW_MIN_WAVES)
lanes_eager_kernel(WalkArgs a_unused, SuspRec *q_unused, uint64_t n_unused, unsigned long long *stats_unused, const uint32_t *edge_row_unused) {
    // (persistent: the wavefronts of a fixed grid stride over the records -- one workgroup per record is bound by the dispatch
    //  rate, see lanes_eager_weighted_kernel)
    __shared__ uint32_t s_mask_all[WAVES_PER_BLOCK][MASK_WORDS];
    __shared__ uint16_t s_rank_all[WAVES_PER_BLOCK][MASK_WORDS + 2];
    __shared__ uint32_t s_queue_all[WAVES_PER_BLOCK][2 * QCAP];
    const int wave = readfirst_u32(threadIdx.x / WAVE);
    uint32_t *const s_mask = s_mask_all[wave], *const s_queue = s_queue_all[wave];
    uint16_t *const s_rank = s_rank_all[wave];
    constexpr size_t XARG = (sizeof(WalkArgs) + 7) & ~(size_t)7;
    SuspRec *q = (SuspRec *)kernarg<uint64_t>(XARG);
    const uint64_t n = kernarg<uint64_t>(XARG + 8);
    unsigned long long n_done = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (uint64_t)wave; i < n; i += (uint64_t)gridDim.x * WAVES_PER_BLOCK) {
    const uint4 *qp = (const uint4 *)(q + i);
    const uint4 q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3];
    if (uni(q0.x) == NOT_FOUND || uni(q2.x) != LANE_EAGER_MARK) continue;
    WalkArgs la = reload_walk_args();
    const ELine *lines = (const ELine *)la.g.tri;
    const uint32_t e = uni(q1.z);
    const uint32_t cur = uni(lines[e].nxt);
    uint32_t prev;
    if (e >= la.g.nnz) prev = e - la.g.nnz;                     // an overflow line: the walk came from vertex e - nnz
    else if (kernarg<uint64_t>(XARG + 24) != 0ull) prev = uni(as_scalar<uint32_t>(kernarg<uint64_t>(XARG + 24))[e]);   // (kept by a partial index)
    else {                                                      // the row that holds CSR entry e
        uint32_t lo = 0, hi = la.g.n_nodes;                     // last v with indptr[v] <= e
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (uni(la.g.indptr[mid]) <= e) lo = mid; else hi = mid; }
        prev = lo;
    }
    const uint32_t s0 = uni(la.g.indptr[cur]), d = uni(la.g.indptr[cur + 1]) - s0;
    const uint32_t t0 = uni(la.g.indptr[prev]), dp = uni(la.g.indptr[prev + 1]) - t0;
    const double r = __longlong_as_double((long long)(((unsigned long long)uni(q3.w) << 32) | uni(q3.z)));
    // the LAZY step first (walk_sparse.hip.h: progressive membership, 64 keys of the shorter row at a time, and the
    // exact-arithmetic decision -- it stops as soon as the target lies inside the classified prefix; the record carries
    // what it needs about the edge: the common-neighbour count and prev's position in cur's row); the eager step -- full
    // mask, float32 chain -- only when that declines
    uint32_t choice = LAZY_FALLBACK;
    {
        const u32x4 rc = as_scalar<u32x4>(PW_KARG(uint64_t, g.vrec))[cur], rp = as_scalar<u32x4>(PW_KARG(uint64_t, g.vrec))[prev];
        const VertexCtx vc{rc.x, rc.y, rc.z, rc.w}, vp{rp.x, rp.y, rp.z, rp.w};
        Prof pf;
        choice = sample_step_unit_lazy(s_mask, s_rank, cur, prev, uni(q1.x), uni(q1.y), false, 0ull, vc, vp, r, pf);
    }
    if (choice == LAZY_FALLBACK) {
        la.g.step_edge = NOT_FOUND;
        choice = sample_step_unit<float, false>(la, s_mask, s_rank, s_queue, cur, true, prev, t0, dp, r, s0, d);
    }
    choice = uni(choice);
    if (lane_id() == 0) q[i].choice = choice;      // (>= d: the mirrored overflow read -- the lane kernel's apply handles it like any other)
    n_done++;
    wave_lds_fence();
    }
    if (lane_id() == 0 && n_done) atomicAdd((unsigned long long *)kernarg<uint64_t>(XARG + 16), n_done);
}

// ---- WEIGHTED form: eager steps and the per-(p, q, extend) tables ------------------------------------------------------------
// A parked step of the weighted lane kernel: the wave-per-walk scan (walk_sparse.hip.h: sample_step_weighted -- membership
// mask from the arriving entry's list, exact sequential float32 chain) with the normaliser from the per-entry table.
// The scan does not start at element 0: the record carries k_safe (every partial sum before it stays below the draw --
// lane_decide_weighted) and the chain's exact value after every CHAIN_CKPT-th element of (prev, cur)'s row was recorded
// with the tables (wckpt_kernel), so it starts at the last recorded value at or before k_safe: at most CHAIN_CKPT + the
// width of the ambiguity instead of half a hub row.  ck_off[e] = first record of entry e; edge_row[e] = its source vertex.
template <bool EXTEND>
__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE, EXTEND ? PW_MIN_WAVES - 1 : PW_MIN_WAVES)
lanes_eager_weighted_kernel(WalkArgs a_unused, SuspRec *q_unused, uint64_t n_unused, unsigned long long *stats_unused,
                            const uint32_t *edge_row_unused, const unsigned long long *ck_off_unused, const float *ck_unused) {
    // PERSISTENT: one workgroup per record (a single wavefront each) was bound by the workgroup dispatch rate -- 81 M launches/s,
    // 1.1 resident wavefronts per SIMD for a kernel whose records take 14 us each (590 ms per C5 pass); the wavefronts of a
    // fixed grid stride over the records instead.
    __shared__ uint32_t s_mask_all[WAVES_PER_BLOCK][MASK_WORDS];
    __shared__ uint32_t s_in_all[EXTEND ? WAVES_PER_BLOCK : 1][EXTEND ? MASK_WORDS : 1];
    __shared__ uint32_t s_queue_all[WAVES_PER_BLOCK][2 * QCAP];
    const int wave = readfirst_u32(threadIdx.x / WAVE);
    uint32_t *const s_mask = s_mask_all[wave], *const s_in = s_in_all[EXTEND ? wave : 0], *const s_queue = s_queue_all[wave];
    constexpr size_t XARG = (sizeof(WalkArgs) + 7) & ~(size_t)7;
    SuspRec *q = (SuspRec *)kernarg<uint64_t>(XARG);
    const uint64_t n = kernarg<uint64_t>(XARG + 8);
    // (records by a fixed stride: a shared counter would serialise 5 M atomics per launch on one L2 channel -- 40 M/s)
    unsigned long long n_done = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (uint64_t)wave; i < n; i += (uint64_t)gridDim.x * WAVES_PER_BLOCK) {
    const uint4 *qp = (const uint4 *)(q + i);
    const uint4 q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3];
    if (uni(q0.x) == NOT_FOUND || uni(q2.x) != LANE_EAGER_MARK) continue;
    WalkArgs la = reload_walk_args();
    const ELine *lines = (const ELine *)la.g.tri;
    const uint32_t e = uni(q1.z), j = uni(q0.y);
    const bool has_prev = j >= 2u;
    uint32_t cur, prev = 0;
    if (!has_prev) cur = as_scalar<uint32_t>(PW_KARG(uint64_t, starts))[uni(q0.x)];
    else {
        cur = uni(lines[e].nxt);
        prev = e >= la.g.nnz ? e - la.g.nnz : uni(as_scalar<uint32_t>(kernarg<uint64_t>(XARG + 24))[e]);
    }
    const uint32_t s0 = uni(q0.z), d = uni(q0.w);               // (cur's row: the record carries it)
    const uint32_t t0 = has_prev ? uni(la.g.indptr[prev]) : 0u, dp = has_prev ? uni(la.g.indptr[prev + 1]) - t0 : 0u;
    const double r = __longlong_as_double((long long)(((unsigned long long)uni(q3.w) << 32) | uni(q3.z)));
    float ktot = 0.0f;
    bool have_tot = false;
    if (la.tot_e) {
        if (!has_prev) { ktot = la.tot_v[cur]; have_tot = true; }
        else if (e < la.g.nnz) { ktot = la.tot_e[e]; have_tot = true; }
    }
    ktot = __uint_as_float(uni(__float_as_uint(ktot)));
    // where the scan starts: the last recorded chain value at or before k_safe (records exist for real entries only)
    uint32_t k_start = 0;
    float c_start = 0.0f;
    if (has_prev && have_tot && e < la.g.nnz && kernarg<uint64_t>(XARG + 40) != 0ull) {
        const uint32_t m = uni(q3.x) / CHAIN_CKPT;                     // (k_safe <= d; record m - 1 = value after element m * CKPT - 1)
        if (m > 0u && m * CHAIN_CKPT < d) {
            const unsigned long long off = as_scalar<unsigned long long>(kernarg<uint64_t>(XARG + 32))[e];
            c_start = __uint_as_float(uni(__float_as_uint(((const float *)kernarg<uint64_t>(XARG + 40))[off + m - 1u])));
            k_start = m * CHAIN_CKPT;
        }
    }
    la.g.step_edge = (has_prev && e < la.g.nnz) ? e : NOT_FOUND;
    uint32_t choice = sample_step_weighted<float, false>(la, s_mask, EXTEND ? s_in : nullptr, s_queue, cur, has_prev, prev, t0, dp, r, s0, d,
                                                         have_tot ? &ktot : nullptr, nullptr, k_start, k_start ? &c_start : nullptr, nullptr,
                                                         2u * CHAIN_CKPT, uni(q1.y), has_prev);   // (q1.y: prev's position in cur's row)
    choice = uni(choice);
    if (lane_id() == 0) q[i].choice = choice;
    n_done++;          // (counted per wavefront: one atomic per RECORD on a single address runs at ~80 M/s -- it was the kernel's pace)
    wave_lds_fence();
    }
    if (lane_id() == 0 && n_done) atomicAdd((unsigned long long *)kernarg<uint64_t>(XARG + 16), n_done);
}

// Base value of every CSR entry (v -> x): the value of x in v's row for a walker whose prev is neither x nor adjacent to x --
// node2vec: fl32(f64(w) / q) (sparse_rw.py:86); node2vec+: fl32(f64(w) * alpha), alpha = 1/q, or min(1, 1/q) when the edge
// (v, x) itself is noisy (w < thr[v]) (sparse_rw.py:119-125 with t = 0).  Depends on cur alone.
template <bool EXTEND>
__global__ void __launch_bounds__(256)
wbase_kernel(const float *__restrict__ data, const uint32_t *__restrict__ edge_row, const float *__restrict__ thr, double q, uint32_t nnz,
             float *wb) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const float w = data[e];
    if (!EXTEND) { wb[e] = Arith<float>::bias_div(w, q); return; }
    const double inv_q = 1.0 / q;
    double alpha = inv_q + (1.0 - inv_q) * 0.0;
    if (Arith<float>::noisy(w, thr[edge_row[e]])) alpha = inv_q < 1.0 ? inv_q : 1.0;
    wb[e] = Arith<float>::bias_mul(w, alpha);
}

// Per-row inclusive float64 prefix sums of the base values, and the running sums of those prefix sums (one wavefront per row)
__global__ void __launch_bounds__(256)
wprefix_kernel(const uint32_t *__restrict__ indptr, const float *__restrict__ wb, uint32_t n_nodes, PrefixPair *pq) {
    const uint32_t v = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE);
    if (v >= n_nodes) return;
    const int lane = lane_id();
    const uint32_t s0 = indptr[v], d = indptr[v + 1] - s0;
    double carry = 0.0, carry2 = 0.0;
    for (uint32_t c0 = 0; c0 < d; c0 += WAVE) {
        const uint32_t k = c0 + (uint32_t)lane;
        double x = k < d ? (double)wb[s0 + k] : 0.0;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const double y = __shfl_up(x, (unsigned)off, WAVE);
            if (lane >= off) x += y;
        }
        const double pk = carry + x;                  // prefix sum at element k (lanes beyond the row: the row's total so far)
        double t = k < d ? pk : 0.0;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const double y = __shfl_up(t, (unsigned)off, WAVE);
            if (lane >= off) t += y;
        }
        if (k < d) pq[s0 + k] = PrefixPair{pk, carry2 + t};
        carry += __shfl(x, WAVE - 1, WAVE);
        carry2 += __shfl(t, WAVE - 1, WAVE);
    }
}

// Per-entry delta prefix sums: entry e = (u -> v), list P_0 < P_1 < ... of positions in row v of the common neighbours of u
// and v.  A walker that arrived by e (prev = u, cur = v) gives the common neighbour x = row_v[P_j] the value
//   node2vec : w(v, x)                                                                       (sparse_rw.py:84-86: not an out edge)
//   node2vec+: w(v, x) when w(u, x) >= thr[x] (in edge); else fl32(f64(w(v, x)) * alpha), alpha = 1/q + (1 - 1/q) * t,
//              t = fl32(w(u, x) / thr[x]), overridden by min(1, 1/q) when w(v, x) < thr[v]      (sparse_rw.py:93-130)
// -- the statements of RowVals::value / value_ext -- and dl[j] = sum_{i <= j} (f64(value_i) - f64(base_i)).  dprev[e] = the
// same difference for prev's own element (fl32(f64(w) / p)).  One thread per entry; entries whose list is not stored
// (partial index) or, for node2vec+, that have no reverse entry get offset ~0: their steps take the eager kernel.
template <bool EXTEND>
__global__ void __launch_bounds__(256)
wlist_kernel(CsrDev g, const uint32_t *__restrict__ edge_row, const float *__restrict__ wb, double p, double q,
             unsigned long long *wl_off, double *dl, double *dprev) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.nnz) return;
    const ELine *lines = (const ELine *)g.tri;
    const float *__restrict__ data = (const float *)g.data;
    const uint4 r0 = *(const uint4 *)(lines + e);
    const uint2 r1 = *((const uint2 *)(lines + e) + 2);
    const uint32_t v = r0.x, n_in = r0.y, rev = r0.z, d_v = r0.w, s_v = r1.x, coff = r1.y;
    const uint32_t u = edge_row[e];
    dprev[e] = rev != NOT_FOUND ? (double)Arith<float>::bias_div(data[s_v + rev], p) - (double)wb[s_v + rev] : 0.0;
    const unsigned long long off = wl_off[e];      // (in: exclusive prefix sum of n_in over the entries)
    if (n_in == 0u) return;
    if (!edge_list_stored(d_v, n_in, coff) || (EXTEND && rev == NOT_FOUND)) { wl_off[e] = ~0ull; return; }
    const ListView P = edge_list(lines, g.clist, (uint32_t)e, d_v, n_in, coff);
    ListView Q = P;
    uint32_t t0 = 0;
    if (EXTEND) {
        const uint32_t e2 = s_v + rev;
        const uint32_t d_u = g.indptr[u + 1] - g.indptr[u];
        if (!edge_list_stored(d_u, lines[e2].n_in, lines[e2].coff)) { wl_off[e] = ~0ull; return; }
        Q = edge_list(lines, g.clist, e2, d_u, lines[e2].n_in, lines[e2].coff);
        t0 = g.indptr[u];
    }
    const double inv_q = 1.0 / q;
    const float thr_cur = EXTEND ? g.thr[v] : 0.0f;
    double run = 0.0;
    // WeightedRow::margin (seqscan.h) relies on every difference below having the sign of q - 1 (value >= base for q >= 1:
    // rounding is monotone).  That holds for the positive finite weights the reference's reader keeps (graph.py:181-215); an
    // entry that breaks it (negative or NaN weights handed in through pw_csr_create) gets no table: its steps take the exact scan.
    const bool want_pos = q >= 1.0;
    bool sign_ok = true;
    for (uint32_t j = 0; j < n_in; j++) {
        const uint32_t pos = P.at(j);
        const float w = data[s_v + pos];
        float val = w;
        if (EXTEND) {
            const uint32_t x = g.indices[s_v + pos];
            const float wpx = data[t0 + Q.at(j)], tx = g.thr[x];
            if (!Arith<float>::in_edge(wpx, tx)) {
                double alpha = inv_q + (1.0 - inv_q) * Arith<float>::t_ratio(wpx, tx);
                if (Arith<float>::noisy(w, thr_cur)) alpha = inv_q < 1.0 ? inv_q : 1.0;
                val = Arith<float>::bias_mul(w, alpha);
            }
        }
        const double diff = (double)val - (double)wb[s_v + pos];
        if (want_pos ? !(diff >= 0.0) : !(diff <= 0.0)) sign_ok = false;
        run += diff;
        dl[off + j] = run;
    }
    if (!sign_ok) wl_off[e] = ~0ull;
}

// ---- verification of the interval decision: the float32 chain decides every recorded step again ------------------------// ---- verification of the interval decision: the float32 chain decides every recorded step again ------------------------
// counts: [0] records checked [1] MISMATCHES (lane_tight's position != the chain's) [2] chains that declined (tie budget)
// bad: the first few mismatching records, for the error message
// bad_jobs (optional, cap entries): the walks of the mismatching records -- the production sample hands them to walk_kernel
__global__ void __launch_bounds__(256)
lanes_verify_kernel(const VerRec *q, uint64_t n, const ELine *__restrict__ lines, const uint8_t *__restrict__ clist, float w_prev,
                    unsigned long long *counts, VerRec *bad, uint32_t bad_cap, uint32_t *bad_jobs = nullptr, uint32_t bad_jobs_cap = 0,
                    uint32_t floats = 0) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long chk = 0, mis = 0, tie = 0;
    if (i < n) {
        const uint4 *qp = (const uint4 *)(q + i);
        const uint4 q0 = qp[0], q1 = qp[1], q2 = qp[2];
        const float tot = __uint_as_float(q2.x), wo = __uint_as_float(q2.y);
        const double r = __longlong_as_double((long long)(((unsigned long long)q2.w << 32) | q2.z));
        const float x_in = 1.0f / tot;
        uint32_t reads = 0;
        uint32_t res;
        // (floats: the FLOATS form's values are three separately rounded quotients and its chain runs over the whole row)
        if (floats) res = lane_chain<true>(q0.x, q0.y, q0.z, r, x_in, wo / tot, w_prev / tot, edge_list(lines, clist, q1.x, q1.z, q0.y, q1.y), reads);
        else res = lane_chain(q0.x, q0.y, q0.z, r, x_in, x_in * wo, x_in * w_prev, edge_list(lines, clist, q1.x, q1.z, q0.y, q1.y), reads);
        chk = 1;
        if (res == LANE_TIE) tie = 1;
        else if (res != q0.w) {
            mis = 1;
            const unsigned long long slot = atomicAdd(counts + 3, 1ull);
            if (bad_jobs && slot < bad_jobs_cap) bad_jobs[slot] = q[i].job;
            if (slot < bad_cap) { bad[slot] = q[i]; bad[slot].job = res; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        chk += (unsigned long long)__shfl_down((long long)chk, (unsigned)off, WAVE);
        mis += (unsigned long long)__shfl_down((long long)mis, (unsigned)off, WAVE);
        tie += (unsigned long long)__shfl_down((long long)tie, (unsigned)off, WAVE);
    }
    if (lane_id() == 0 && chk) {
        atomicAdd(counts + 0, chk);
        if (mis) atomicAdd(counts + 1, mis);
        if (tie) atomicAdd(counts + 2, tie);
    }
}

// ---- index build ------------------------------------------------------------------------------------------------
// One set intersection per adjacent PAIR {h, k}, by the endpoint with the longer row (ties: the larger id): row h
// waits in LDS (sorted ids, 8192 at a time), the neighbours k of h stream their rows through it (coalesced reads,
// binary search in LDS -- no hash table, no filter, no global random access), and every common neighbour w yields both
// lists at once: its position in row k goes to the list of (h -> k), its position in row h to the list of (k -> h).
// Two passes of the same kernel: COUNT (n_in of both entries), then -- after the offsets of the long lists are known --
// FILL.  Round 2 ran four intersections per pair (count and fill for each direction) through a global hash index:
// 323 + 349 ms at RMAT-22.
//
// Entries whose reverse edge is absent (directed graphs) are handled by their source.  Rows longer than LB_SEG are
// processed LB_SEG positions at a time by separate workgroups ("segments": the ids of a segment are a contiguous id
// range, a neighbour's keys inside that range a contiguous sub-row found by two binary searches); the per-segment
// counts of such rows are kept (segcnt) so that FILL knows where a segment's block of a list starts.
constexpr int LB_SEG = 8192;      // row positions per workgroup (32 KB of LDS)
constexpr int LB_SMALL = 256;     // rows up to this length: one wavefront per row
constexpr uint32_t LB_SEQ = 32;   // neighbour rows up to this length are walked by ONE thread, longer ones by a wavefront

struct LaneBuildArgs {
    const uint32_t *__restrict__ indptr;
    const uint32_t *__restrict__ indices;
    ELine *lines;
    uint8_t *clist;                 // FILL only
    uint32_t *segcnt;               // per-(neighbour, segment) counts of the rows longer than LB_SEG
    uint32_t max_len;               // FILL: lists longer than this were left out of the index (partial index; 0xffffffff: none)
    // LOGGED build (round 5): the COUNT pass keeps its matches -- (position in row k) | (position in row h) << 16, in match order,
    // in a block of the log sized by the pair's upper bound d_k (log_off[e2] slots in; a segment's sub-block starts at the
    // first key of row k inside the segment's id range) -- and lane_scatter_kernel copies them to their places once the
    // offsets are known: the intersection of the long rows (50 GB of neighbour rows streamed through LDS at RMAT-22) runs
    // ONCE.  Rows beyond 65536 entries log 64-bit words (log_wide).
    uint32_t *log;
    const unsigned long long *log_off;
    uint32_t *seglo;                // per-(neighbour, segment): first key of row k inside the segment (sub-block start), like segcnt
    uint32_t logged;                // FILL: the loggable pairs were written by lane_scatter_kernel -- skip them
};
// a pair (h -> k) handled by h whose two lists can be logged: both positions fit 16 bits
// Log words: (position in row k) | (position in row h) << 16 while row h has at most 65536 entries (an entry without a reverse
// edge has no list (k -> h): its word is the position in row k alone, all 32 bits); rows h beyond that use 64-bit words,
// (position in row k) | (position in row h) << 32, two slots each.  Every pair a vertex takes can be logged.
__device__ __forceinline__ bool log_wide(uint32_t d_h) { return d_h > 65536u; }
__device__ __forceinline__ uint32_t log_pair_slots(uint32_t d_h, uint32_t d_k) { return log_wide(d_h) ? 2u * d_k : (d_k + 1u) & ~1u; }

// ELine[e] = { v, 0, position of u in row v, degree(v), indptr[v], 0 } for e = (u -> v); the reverse position comes
// from one probe of the adjacency index (walk_sparse.hip.h)
__global__ void __launch_bounds__(256)
eline_init_kernel(CsrDev g, const uint32_t *__restrict__ edge_row, ELine *lines) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.nnz) return;
    const uint32_t u = edge_row[e], v = g.indices[e];
    const uint4 vr = g.vrec[v];
    uint32_t rev = NOT_FOUND;
    if (vr.y) {   // a vertex without out-edges owns no index slots
        const uint64_t vtb = g.tab_off[v];
        const uint32_t vmask = (uint32_t)(g.tab_off[v + 1] - vtb) - 1u;
        rev = adj_lookup(g.slots + vtb, vmask, u, true);
    }
    uint4 *lp = (uint4 *)(lines + e);
    lp[0] = make_uint4(v, 0u, rev, vr.y);
    *(uint2 *)(lp + 1) = make_uint2(vr.x, 0u);
}

// OVERFLOW lines.  Vertex v's "choice == degree" read lands on x0(v) = indices[indptr[v + 1]] (first neighbour of the
// next non-empty row); the walk then stands on x0 having come from v -- a pair that need not be an edge, so no CSR entry
// carries its record.  lines[nnz + v] = { x0, |N(v) & N(x0)|, position of v in row x0, degree(x0), indptr[x0], list offset }
// + the list (positions in row x0 of the common neighbours), like any edge line; nxt = NOT_FOUND when v has no
// neighbours or the read would leave the index array (the walk kernel clamps it: redo path).
__global__ void __launch_bounds__(256)
vline_init_kernel(CsrDev g, ELine *lines) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.n_nodes) return;
    const uint32_t s0 = g.indptr[v], s1 = g.indptr[v + 1];
    uint4 l0 = make_uint4(NOT_FOUND, 0u, NOT_FOUND, 0u);
    uint2 l1 = make_uint2(0u, 0u);
    if (s1 > s0 && s1 < g.nnz) {
        const uint32_t x0 = g.indices[s1];
        const uint4 vr = g.vrec[x0];
        uint32_t rev = NOT_FOUND;
        if (vr.y) {
            const uint64_t tb = g.tab_off[x0];
            const uint32_t tmask = (uint32_t)(g.tab_off[x0 + 1] - tb) - 1u;
            rev = adj_lookup(g.slots + tb, tmask, v, true);
        }
        l0 = make_uint4(x0, 0u, rev, vr.y);
        l1 = make_uint2(vr.x, 0u);
    }
    uint4 *lp = (uint4 *)(lines + (uint64_t)g.nnz + v);
    lp[0] = l0;
    *(uint2 *)(lp + 1) = l1;
}

// The lists of the overflow lines: one wavefront per vertex v streams row v and looks every neighbour up in x0's
// adjacency index (one probe; x0 is a hub more often than not -- the smallest neighbour of the next vertex -- and its
// table stays in L2): the hits, in row order, ARE the ascending positions in row x0.  FILL = false: the count.
// (Round 5 tried keeping the count pass's hits and copying them by a per-vertex scatter instead of the second lookup pass: 15 ms
// against 6.4 -- one thread per vertex with scattered reads; the two passes stay.)
template <bool FILL>
__global__ void __launch_bounds__(256)
vline_lists_kernel(CsrDev g, ELine *lines, uint8_t *clist) {
    const uint32_t v = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE);
    if (v >= g.n_nodes) return;
    const int lane = lane_id();
    const uint32_t e = g.nnz + v;
    const uint4 r0 = *(const uint4 *)(lines + e);
    const uint32_t x0 = r0.x, d0 = r0.w;
    if (x0 == NOT_FOUND || d0 == 0) return;
    if (FILL && !(d0 <= 65536u && r0.y <= EL_INLINE) && lines[e].coff == EL_NO_LIST) return;   // (partial index: list left out)
    const uint32_t s_v = g.indptr[v], d_v = g.indptr[v + 1] - s_v;
    const uint64_t tb = g.tab_off[x0];
    const uint32_t tmask = (uint32_t)(g.tab_off[x0 + 1] - tb) - 1u;
    const bool narrow = d0 <= 65536u;
    uint8_t *p = nullptr;
    if (FILL) p = (uint8_t *)((narrow && r0.y <= EL_INLINE) ? (uint8_t *)(lines + e) + 24 : clist + (uint64_t)lines[e].coff * 16u);
    const uint64_t lane_lt = (1ull << lane) - 1ull;
    uint32_t run = 0;
    for (uint32_t c0 = 0; c0 < d_v; c0 += WAVE) {
        const uint32_t i = c0 + (uint32_t)lane;
        const bool valid = i < d_v;
        const uint32_t w = valid ? g.indices[s_v + i] : 0u;
        const uint32_t gpos = adj_lookup(g.slots + tb, tmask, w, valid);
        // (w == v -- a self loop at v that x0 is adjacent to -- is prev's own position in row x0: the reference takes prev out
        //  of the common neighbours, sparse_rw.py:79-87; see loop_fix_kernel)
        const bool hit = valid && gpos != NOT_FOUND && w != v;
        const uint64_t m = ballot(hit);
        if (FILL && hit) {
            const uint32_t rk = run + (uint32_t)__popcll(m & lane_lt);
            if (narrow) ((uint16_t *)p)[rk] = (uint16_t)gpos; else ((uint32_t *)p)[rk] = gpos;
        }
        run += (uint32_t)__popcll(m);
    }
    if (!FILL && lane == 0) lines[e].n_in = run;
}
__device__ __forceinline__ bool list_is_narrow(uint32_t deg) { return deg <= 65536u; }
__device__ __forceinline__ bool list_is_inline(uint32_t deg, uint32_t n_in) { return list_is_narrow(deg) && n_in <= EL_INLINE; }
__device__ __forceinline__ const uint8_t *list_base(const ELine *lines, const uint8_t *clist, uint32_t e, uint32_t deg,
                                                    uint32_t n_in, uint32_t coff) {
    return list_is_inline(deg, n_in) ? (const uint8_t *)(lines + e) + 24 : clist + (uint64_t)coff * 16u;
}

// first index in keys[0, P) (P a power of two, keys[len..P] = 0xffffffff) whose key is >= w
__device__ __forceinline__ uint32_t lds_lower_bound(const uint32_t *keys, uint32_t P, uint32_t w) {
    uint32_t lo = 0, len = P;
    while (len > 1) {
        const uint32_t half = len >> 1;
        lo = keys[lo + half - 1u] < w ? lo + half : lo;
        len -= half;
    }
    return lo + (keys[lo] < w ? 1u : 0u);
}

// work item of the build: vertex h, segment seg of nseg, base of its counts in segcnt (nseg > 1), neighbour range
struct LaneBuildItem {
    uint32_t h, seg, nseg, m0;
    uint32_t j0, j1;   // neighbours (positions of row h) this workgroup takes: the longest rows are split further
};

template <int THREADS, int CAP, bool FILL, bool LOG = false>
__global__ void __launch_bounds__(THREADS)
lane_lists_kernel(LaneBuildArgs a, const LaneBuildItem *__restrict__ items) {
    static_assert(!(FILL && LOG), "the log is written by the COUNT pass");
    constexpr int NW = THREADS / WAVE;
    __shared__ uint32_t keys[CAP + 1];
    __shared__ uint32_t qj[THREADS], qlo[THREADS], qhi[THREADS];
    __shared__ uint32_t qn;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const LaneBuildItem it = items[blockIdx.x];
    const uint32_t h = it.h, nseg = it.nseg;
    const uint32_t s_h = a.indptr[h], d_h = a.indptr[h + 1] - s_h;
    const uint32_t a0 = it.seg * (uint32_t)CAP;
    const uint32_t len = d_h - a0 < (uint32_t)CAP ? d_h - a0 : (uint32_t)CAP;
    uint32_t P = 1;
    while (P < len) P <<= 1;
    for (uint32_t i = tid; i <= P; i += THREADS) keys[i] = i < len ? a.indices[s_h + a0 + i] : 0xffffffffu;
    __syncthreads();
    const uint32_t id_lo = keys[0], id_hi = keys[len - 1u];
    const bool h_narrow = list_is_narrow(d_h);
    const uint64_t lane_lt = (1ull << lane) - 1ull;

    for (uint32_t base = it.j0; base < it.j1; base += THREADS) {
        if (tid == 0) qn = 0;
        __syncthreads();
        const uint32_t j = base + (uint32_t)tid;
        if (j < it.j1) {
            const uint32_t e2 = s_h + j;
            const uint4 r0 = *(const uint4 *)(a.lines + e2);          // { k, n_in, position of h in row k, degree(k) }
            const uint2 r1 = *((const uint2 *)(a.lines + e2) + 2);    // { indptr[k], coff }
            // (a SELF LOOP h -> h is a pair of its own: its list -- positions in row h of N(h) & N(h) -- has one direction only,
            //  so it is written like an entry without a reverse edge)
            const uint32_t k = r0.x, rev = r0.x == h ? NOT_FOUND : r0.z, d_k = r0.w, s_k = r1.x;
            const bool mine = rev == NOT_FOUND || d_h > d_k || (d_h == d_k && h > k);
            // single-segment rows: the COUNT pass already wrote the lists that fit their lines (below) -- nothing to fill.
            // "Fits its line" includes the row of k being narrow (uint16 positions): a DIRECTED entry h -> k into a row of
            // more than 65536 entries that has no reverse entry is taken by h whatever the degrees, and its short list
            // lives in the overflow array as uint32 positions, which only the FILL pass writes.
            // Partial index: a list longer than max_len is stored nowhere (both directions: same length) -- nothing to fill.
            const bool done = FILL && ((nseg == 1 && list_is_inline(d_k, r0.y)) || r0.y > a.max_len || a.logged);
            if (mine && d_k && !done) {
                uint32_t lo_i = 0, hi_i = d_k;
                if (nseg > 1) {   // keys of row k inside this segment's id range
                    lo_i = lower_bound_u32(a.indices + s_k, d_k, id_lo);
                    hi_i = id_hi == 0xffffffffu ? d_k : lower_bound_u32(a.indices + s_k, d_k, id_hi + 1u);
                }
                if (hi_i > lo_i) {
                    if (hi_i - lo_i <= LB_SEQ) {
                        // ---- a short neighbour row: this thread alone ----
                        uint32_t cnt = 0;
                        uint8_t *p2 = nullptr, *p1 = nullptr;
                        bool k_narrow = true;
                        // COUNT pass, single-segment row (both rows <= 8192 entries: uint16 positions): the first
                        // EL_INLINE matches go into the two lines right away; if the list turns out longer the line's
                        // inline area is simply not used (the list then lives in the overflow array: FILL pass)
                        const bool spec = !FILL && nseg == 1;
                        if (spec) {
                            p2 = (uint8_t *)(a.lines + e2) + 24;
                            if (rev != NOT_FOUND) p1 = (uint8_t *)(a.lines + (s_k + rev)) + 24;
                        }
                        if (FILL) {
                            if (nseg > 1) for (uint32_t sg = 0; sg < it.seg; sg++) cnt += a.segcnt[it.m0 + j * nseg + sg];
                            k_narrow = list_is_narrow(d_k);
                            p2 = (uint8_t *)list_base(a.lines, a.clist, e2, d_k, r0.y, r1.y);
                            if (rev != NOT_FOUND) {
                                const uint32_t e1 = s_k + rev;
                                const uint32_t n1 = a.lines[e1].n_in, c1 = a.lines[e1].coff;
                                p1 = (uint8_t *)list_base(a.lines, a.clist, e1, d_h, n1, c1);
                            }
                        }
                        uint32_t *lg = nullptr;
                        if (LOG) lg = a.log + a.log_off[e2] + (log_wide(d_h) ? 2u * lo_i : lo_i);
                        for (uint32_t i = lo_i; i < hi_i; i++) {
                            const uint32_t w = a.indices[s_k + i];
                            const uint32_t idx = lds_lower_bound(keys, P, w);
                            if (keys[idx] == w) {
                                if (LOG) {
                                    if (log_wide(d_h)) ((unsigned long long *)lg)[cnt] = (unsigned long long)i | ((unsigned long long)(a0 + idx) << 32);
                                    else lg[cnt] = rev == NOT_FOUND ? i : (i | ((a0 + idx) << 16));
                                }
                                if (FILL) {
                                    if (k_narrow) ((uint16_t *)p2)[cnt] = (uint16_t)i; else ((uint32_t *)p2)[cnt] = i;
                                    if (p1) { if (h_narrow) ((uint16_t *)p1)[cnt] = (uint16_t)(a0 + idx); else ((uint32_t *)p1)[cnt] = a0 + idx; }
                                } else if (spec && cnt < EL_INLINE) {
                                    ((uint16_t *)p2)[cnt] = (uint16_t)i;
                                    if (p1) ((uint16_t *)p1)[cnt] = (uint16_t)(a0 + idx);
                                }
                                cnt++;
                            }
                        }
                        if (!FILL) {
                            if (nseg > 1) {
                                a.segcnt[it.m0 + j * nseg + it.seg] = cnt;
                                if (LOG) a.seglo[it.m0 + j * nseg + it.seg] = lo_i;
                                if (cnt) {
                                    atomicAdd(&a.lines[e2].n_in, cnt);
                                    if (rev != NOT_FOUND) atomicAdd(&a.lines[s_k + rev].n_in, cnt);
                                }
                            } else {
                                a.lines[e2].n_in = cnt;
                                if (rev != NOT_FOUND) a.lines[s_k + rev].n_in = cnt;
                            }
                        }
                    } else {
                        const uint32_t slot = atomicAdd(&qn, 1u);
                        qj[slot] = j; qlo[slot] = lo_i; qhi[slot] = hi_i;
                    }
                } else if (!FILL && nseg > 1) a.segcnt[it.m0 + j * nseg + it.seg] = 0;
            }
        }
        __syncthreads();
        // ---- longer neighbour rows: one wavefront each, 64 keys per trip ----
        const uint32_t nq = qn;
        for (uint32_t qi = (uint32_t)wv; qi < nq; qi += NW) {
            const uint32_t jq = qj[qi], lo_i = qlo[qi], hi_i = qhi[qi];
            const uint32_t e2 = s_h + jq;
            const uint4 r0 = *(const uint4 *)(a.lines + e2);
            const uint2 r1 = *((const uint2 *)(a.lines + e2) + 2);
            const uint32_t rev = r0.x == h ? NOT_FOUND : r0.z, d_k = r0.w, s_k = r1.x;   // (self loop: as above)
            uint32_t run = 0;
            uint8_t *p2 = nullptr, *p1 = nullptr;
            bool k_narrow = true;
            const bool spec = !FILL && nseg == 1;   // (as above: lists that fit their lines are written by the COUNT pass)
            if (spec) {
                p2 = (uint8_t *)(a.lines + e2) + 24;
                if (rev != NOT_FOUND) p1 = (uint8_t *)(a.lines + (s_k + rev)) + 24;
            }
            if (FILL) {
                if (nseg > 1) for (uint32_t sg = 0; sg < it.seg; sg++) run += a.segcnt[it.m0 + jq * nseg + sg];
                k_narrow = list_is_narrow(d_k);
                p2 = (uint8_t *)list_base(a.lines, a.clist, e2, d_k, r0.y, r1.y);
                if (rev != NOT_FOUND) {
                    const uint32_t e1 = s_k + rev;
                    const uint32_t n1 = a.lines[e1].n_in, c1 = a.lines[e1].coff;
                    p1 = (uint8_t *)list_base(a.lines, a.clist, e1, d_h, n1, c1);
                }
            }
            uint32_t *lg = nullptr;
            if (LOG) lg = a.log + a.log_off[e2] + (log_wide(d_h) ? 2u * lo_i : lo_i);
            for (uint32_t c0 = lo_i; c0 < hi_i; c0 += WAVE) {
                const uint32_t i = c0 + (uint32_t)lane;
                const bool valid = i < hi_i;
                const uint32_t w = valid ? a.indices[s_k + i] : 0xffffffffu;
                const uint32_t idx = lds_lower_bound(keys, P, w);
                const bool hit = valid && keys[idx] == w;
                const uint64_t m = ballot(hit);
                if (LOG && hit) {
                    const uint32_t rk = run + (uint32_t)__popcll(m & lane_lt);
                    if (log_wide(d_h)) ((unsigned long long *)lg)[rk] = (unsigned long long)i | ((unsigned long long)(a0 + idx) << 32);
                    else lg[rk] = rev == NOT_FOUND ? i : (i | ((a0 + idx) << 16));
                }
                if (FILL && hit) {
                    const uint32_t rk = run + (uint32_t)__popcll(m & lane_lt);
                    if (k_narrow) ((uint16_t *)p2)[rk] = (uint16_t)i; else ((uint32_t *)p2)[rk] = i;
                    if (p1) { if (h_narrow) ((uint16_t *)p1)[rk] = (uint16_t)(a0 + idx); else ((uint32_t *)p1)[rk] = a0 + idx; }
                } else if (spec && hit) {
                    const uint32_t rk = run + (uint32_t)__popcll(m & lane_lt);
                    if (rk < EL_INLINE) {
                        ((uint16_t *)p2)[rk] = (uint16_t)i;
                        if (p1) ((uint16_t *)p1)[rk] = (uint16_t)(a0 + idx);
                    }
                }
                run += (uint32_t)__popcll(m);
            }
            if (!FILL && lane == 0) {
                if (nseg > 1) {
                    a.segcnt[it.m0 + jq * nseg + it.seg] = run;
                    if (LOG) a.seglo[it.m0 + jq * nseg + it.seg] = lo_i;
                    if (run) {
                        atomicAdd(&a.lines[e2].n_in, run);
                        if (rev != NOT_FOUND) atomicAdd(&a.lines[s_k + rev].n_in, run);
                    }
                } else {
                    a.lines[e2].n_in = run;
                    if (rev != NOT_FOUND) a.lines[s_k + rev].n_in = run;
                }
            }
        }
        __syncthreads();
    }
}

constexpr int CL_BLOCK = 256;
constexpr int CL_ITEMS = 16;
constexpr int CL_TILE = CL_BLOCK * CL_ITEMS;

// ---- LOGGED build: slots of the log per entry, and the copy of the logged matches to their places ---------------------
// slots of entry e2 = (h -> k): d_k when h takes the pair (the longer row; ties: the larger id; entries without a reverse
// edge: their source) and both positions fit 16 bits, else 0
__device__ __forceinline__ uint32_t log_slots(const ELine *lines, const uint32_t *indptr, const uint32_t *edge_row, uint64_t e2) {
    const uint4 r0 = *(const uint4 *)(lines + e2);
    const uint32_t h = edge_row[e2], k = r0.x, rev = r0.z, d_k = r0.w;
    const uint32_t d_h = indptr[h + 1] - indptr[h];
    const bool mine = rev == NOT_FOUND || d_h > d_k || (d_h == d_k && h > k);
    return (mine && d_h >= 2u && d_k) ? log_pair_slots(d_h, d_k) : 0u;
}
__global__ void __launch_bounds__(CL_BLOCK)
log_tile_sums_kernel(const ELine *__restrict__ lines, const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ edge_row, uint32_t nnz,
                     uint64_t *tile_sums) {
    __shared__ uint64_t sh[CL_BLOCK];
    const uint64_t base = (uint64_t)blockIdx.x * CL_TILE + (uint64_t)threadIdx.x * CL_ITEMS;
    uint64_t s = 0;
    for (int k = 0; k < CL_ITEMS; k++)
        if (base + k < nnz) s += log_slots(lines, indptr, edge_row, base + k);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = CL_BLOCK / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(CL_BLOCK)
log_offsets_kernel(const ELine *__restrict__ lines, const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ edge_row, uint32_t nnz,
                   const uint64_t *__restrict__ tile_sums, unsigned long long *off_out) {
    __shared__ uint64_t sh[CL_BLOCK];
    const int t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * CL_TILE + (uint64_t)t * CL_ITEMS;
    uint32_t loc[CL_ITEMS];
    uint64_t s = 0;
    for (int k = 0; k < CL_ITEMS; k++) {
        loc[k] = base + k < nnz ? log_slots(lines, indptr, edge_row, base + k) : 0u;
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
        if (base + k < nnz) off_out[base + k] = run;
        run += loc[k];
    }
}

// The logged matches of every pair to their places: list (h -> k) = the low halves (positions in row k), list (k -> h) = the
// high halves (positions in row h), both uint16.  One wavefront per 64 consecutive entries: lists of up to 32 matches are
// copied by their lane, longer ones by the whole wavefront, one after the other.  vm0[h] = base of h's per-segment counts
// (rows longer than LB_SEG: their matches sit in one sub-block per segment, concatenated here in segment order).
__global__ void __launch_bounds__(256)
lane_scatter_kernel(LaneBuildArgs a, const uint32_t *__restrict__ edge_row, const uint32_t *__restrict__ vm0, uint32_t nnz) {
    const int lane = lane_id();
    const uint64_t e2 = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool todo = false;
    uint32_t n = 0, nseg = 1, segbase = 0, s_k = 0, rev = NOT_FOUND, k_wide = 0, h_wide = 0;
    uint8_t *p2 = nullptr, *p1 = nullptr;
    const uint32_t *src = nullptr;
    if (e2 < nnz && log_slots(a.lines, a.indptr, edge_row, e2)) {
        const uint4 r0 = *(const uint4 *)(a.lines + e2);
        const uint2 r1 = *((const uint2 *)(a.lines + e2) + 2);
        const uint32_t h = edge_row[e2], d_k = r0.w;
        const uint32_t s_h = a.indptr[h], d_h = a.indptr[h + 1] - s_h;
        n = r0.y; rev = r0.z; s_k = r1.x;
        k_wide = list_is_narrow(d_k) ? 0u : 1u;
        h_wide = log_wide(d_h) ? 1u : 0u;
        nseg = d_h > (uint32_t)LB_SEG ? (d_h + LB_SEG - 1) / LB_SEG : 1u;
        // (single-segment rows: the COUNT pass wrote the lists that fit their lines itself; partial index: lists left out)
        todo = n != 0u && !(nseg == 1u && list_is_inline(d_k, n)) && n <= a.max_len;
        if (todo) {
            p2 = (uint8_t *)list_base(a.lines, a.clist, (uint32_t)e2, d_k, n, r1.y);
            if (rev != NOT_FOUND) {
                const uint32_t e1 = s_k + rev;
                p1 = (uint8_t *)list_base(a.lines, a.clist, e1, d_h, a.lines[e1].n_in, a.lines[e1].coff);
            }
            src = a.log + a.log_off[e2];
            if (nseg > 1u) segbase = vm0[h] + ((uint32_t)e2 - s_h) * nseg;
        }
    }
    // word i of a block -> (position in row k, position in row h)
    auto put = [](uint8_t *q2, uint8_t *q1, uint32_t at, uint32_t pk, uint32_t ph, uint32_t kw, uint32_t hw) {
        if (kw) ((uint32_t *)q2)[at] = pk; else ((uint16_t *)q2)[at] = (uint16_t)pk;
        if (q1) { if (hw) ((uint32_t *)q1)[at] = ph; else ((uint16_t *)q1)[at] = (uint16_t)ph; }
    };
    // ... and the PIVOTS of both lists into the inline areas of their lines (seqscan.h: entry (t + 1) * step of the list, 20 of them,
    // 10 for uint32 lists) -- the words are at hand here; eline_pivots_kernel is left with the overflow lines
    const bool small_one = todo && nseg == 1u && n <= 32u;    // (single-segment rows are narrow: 32-bit words)
    if (small_one) {
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t v = src[i];
            put(p2, p1, i, rev == NOT_FOUND ? v : (v & 0xffffu), v >> 16, k_wide, 0u);
        }
        if (list_has_pivots(k_wide, n)) {                     // (n > 20: the list (h -> k) lives in the overflow array)
            const uint32_t np = list_pivot_count(k_wide), step = list_pivot_step(k_wide, n);
            uint8_t *dst = (uint8_t *)(a.lines + e2) + 24;
            for (uint32_t t = 0; t < np; t++) {
                const uint32_t v = src[(t + 1u) * step];
                put(dst, nullptr, t, rev == NOT_FOUND ? v : (v & 0xffffu), 0u, k_wide, 0u);
            }
        }
        if (p1 && list_has_pivots(0u, n)) {
            const uint32_t step = list_pivot_step(0u, n);
            uint16_t *dst = (uint16_t *)((uint8_t *)(a.lines + (s_k + rev)) + 24);
            for (uint32_t t = 0; t < LIST_PIVOTS_NARROW; t++) dst[t] = (uint16_t)(src[(t + 1u) * step] >> 16);
        }
    }
    uint64_t big = ballot(todo && !small_one);
    while (big) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1ull;
        const uint32_t nsg = readlane_u32(nseg, l), sb = readlane_u32(segbase, l), nn = readlane_u32(n, l);
        const uint32_t kw = readlane_u32(k_wide, l), hw = readlane_u32(h_wide, l), norev = readlane_u32(rev, l) == NOT_FOUND ? 1u : 0u;
        const uint32_t *sp = (const uint32_t *)readlane_u64((uint64_t)src, l);
        uint8_t *q2 = (uint8_t *)readlane_u64((uint64_t)p2, l), *q1 = (uint8_t *)readlane_u64((uint64_t)p1, l);
        uint32_t out = 0;
        for (uint32_t sg = 0; sg < nsg; sg++) {
            const uint32_t cnt = nsg > 1u ? a.segcnt[sb + sg] : nn;
            const uint32_t lo = nsg > 1u ? a.seglo[sb + sg] : 0u;
            for (uint32_t i = (uint32_t)lane; i < cnt; i += WAVE) {
                if (hw) {
                    const unsigned long long v = ((const unsigned long long *)sp)[lo + i];
                    put(q2, q1, out + i, (uint32_t)v, (uint32_t)(v >> 32), kw, 1u);
                } else {
                    const uint32_t v = sp[lo + i];
                    put(q2, q1, out + i, norev ? v : (v & 0xffffu), v >> 16, kw, 0u);
                }
            }
            out += cnt;
        }
        // pivots: lane t looks entry (t + 1) * step up through the segment table
        const uint32_t e2l = (uint32_t)((uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u) + (uint32_t)l);
        const uint32_t s_kl = readlane_u32(s_k, l), revl = readlane_u32(rev, l);
        for (int which = 0; which < 2; which++) {
            const uint32_t wide = which == 0 ? kw : hw;
            if (which == 1 && !q1) break;
            // (the list must live in the overflow array: its inline area is free for the pivots)
            if (!list_has_pivots(wide, nn)) continue;
            const uint32_t np = list_pivot_count(wide), step = list_pivot_step(wide, nn);
            if ((uint32_t)lane < np) {
                uint32_t target = ((uint32_t)lane + 1u) * step, base = 0, pk = 0, ph = 0;
                for (uint32_t sg = 0; sg < nsg; sg++) {
                    const uint32_t cnt = nsg > 1u ? a.segcnt[sb + sg] : nn;
                    if (target < base + cnt) {
                        const uint32_t lo = nsg > 1u ? a.seglo[sb + sg] : 0u;
                        if (hw) { const unsigned long long v = ((const unsigned long long *)sp)[lo + target - base]; pk = (uint32_t)v; ph = (uint32_t)(v >> 32); }
                        else { const uint32_t v = sp[lo + target - base]; pk = norev ? v : (v & 0xffffu); ph = v >> 16; }
                        break;
                    }
                    base += cnt;
                }
                uint8_t *dst = (uint8_t *)(a.lines + (which == 0 ? e2l : s_kl + revl)) + 24;
                if (wide) ((uint32_t *)dst)[lane] = which == 0 ? pk : ph; else ((uint16_t *)dst)[lane] = (uint16_t)(which == 0 ? pk : ph);
            }
        }
    }
}

// ---- SELF LOOPS (round 6) --------------------------------------------------------------------------------------------
// The reference accepts them (graph.py:238-268: no check).  For a walker that stands on v having come from u the reference
// classes are disjoint: prev's own position in row v gets 1/p and is taken OUT of the common neighbours
// (`non_com_nbr[prev_ptr] = False; ... unnormalized_probs[prev_ptr] /= p`, sparse_rw.py:79-87) -- but when u has a self loop
// (u in N(u)) and v -> u is an edge, u IS a member of N(u) & N(v), and the symmetric intersection above puts its position
// into the list of (u -> v).  Every consumer of a list (lane_decide / lane_tight / lane_chain, the float64-bounded forms, the
// wave kernel's mask scatter) assumes the three classes disjoint, so the build REMOVES that one entry afterwards:
// list(u -> v) = positions in row v of (N(u) & N(v)) \ {u}.  A self loop at v needs nothing: v's position in its own row is
// an ordinary common neighbour whenever v is adjacent to prev.  selfbits: bit u = vertex u has a self loop.
__global__ void __launch_bounds__(256)
self_loop_bits_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices, uint32_t n_nodes, uint32_t *bits) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes) return;
    const uint32_t s0 = indptr[v], d = indptr[v + 1] - s0;
    const uint32_t i = lower_bound_u32(indices + s0, d, v);
    if (i < d && indices[s0 + i] == v) atomicOr(bits + (v >> 5), 1u << (v & 31u));
}
// One thread per CSR entry e = (u -> v) whose source has a self loop and whose reverse entry exists: the entry rev_pos leaves
// the list, the tail moves up, n_in drops by one (a list that now fits the line moves into it).  After the FILL pass, before
// the pivots.  removed[0] counts the entries taken out.  (Lists must be stored: a partial index is not built for such graphs.)
__global__ void __launch_bounds__(256)
loop_fix_kernel(const uint32_t *__restrict__ edge_row, const uint32_t *__restrict__ selfbits, ELine *lines, uint8_t *clist, uint32_t nnz,
                unsigned long long *removed) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const uint32_t u = edge_row[e];
    if (!((selfbits[u >> 5] >> (u & 31u)) & 1u)) return;
    const uint4 r0 = *(const uint4 *)(lines + e);
    const uint32_t n_in = r0.y, rev = r0.z, d_v = r0.w;
    if (rev == NOT_FOUND || n_in == 0u) return;
    const uint32_t coff = lines[e].coff;
    const bool narrow = d_v <= 65536u, old_inl = narrow && n_in <= EL_INLINE;
    if (!old_inl && coff == EL_NO_LIST) return;
    uint8_t *src = old_inl ? (uint8_t *)(lines + e) + 24 : clist + (uint64_t)coff * 16u;
    auto get = [&](uint32_t i) -> uint32_t { return narrow ? (uint32_t)((const uint16_t *)src)[i] : ((const uint32_t *)src)[i]; };
    uint32_t lo = 0, hi = n_in;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (get(mid) < rev) lo = mid + 1u; else hi = mid; }
    if (lo >= n_in || get(lo) != rev) return;             // (prev is not among the common neighbours: nothing to take out)
    const uint32_t n = n_in - 1u;
    const bool new_inl = narrow && n <= EL_INLINE;
    if (!old_inl && new_inl) {                            // 21 -> 20 entries: the list moves into its line
        uint16_t *dst = (uint16_t *)((uint8_t *)(lines + e) + 24);
        for (uint32_t i = 0, o = 0; i < n_in; i++)
            if (i != lo) dst[o++] = (uint16_t)get(i);
    } else {
        for (uint32_t i = lo; i < n; i++) {
            if (narrow) ((uint16_t *)src)[i] = ((const uint16_t *)src)[i + 1u];
            else ((uint32_t *)src)[i] = ((const uint32_t *)src)[i + 1u];
        }
    }
    lines[e].n_in = n;
    atomicAdd(removed, 1ull);
}

// 16-byte units the list of entry e takes in the overflow array (0: it lives inside the edge line)
__device__ __forceinline__ uint32_t list_units(const ELine *lines, uint64_t e, uint32_t max_len = 0xffffffffu) {
    const uint4 r0 = *(const uint4 *)(lines + e);
    if (list_is_inline(r0.w, r0.y) || r0.y > max_len) return 0u;
    return (r0.y * (list_is_narrow(r0.w) ? 2u : 4u) + 15u) >> 4;
}

// Partial index: histogram of the overflow array's 16-byte units by list length (lists of hist_len entries or more share
// the last bin) -- the host picks the largest length whose cumulative units fit the byte budget.
__global__ void __launch_bounds__(256)
clist_length_hist_kernel(const ELine *__restrict__ lines, uint32_t n_lines, unsigned long long *hist, uint32_t hist_len) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_lines) return;
    const uint32_t u = list_units(lines, e);
    if (!u) return;
    const uint32_t n = lines[e].n_in;
    atomicAdd(hist + (n < hist_len ? n : hist_len - 1u), (unsigned long long)u);
}

// PIVOTS of the lists that live in the overflow array, written into the unused inline area of their lines after the FILL
// pass (seqscan.h: ListView / list_search_pivots): entry (k + 1) * step of the list, k = 0 .. 19 (9 for uint32 lists)
__global__ void __launch_bounds__(256)
eline_pivots_kernel(ELine *lines, const uint8_t *__restrict__ clist, uint32_t n_lines, uint32_t first = 0) {
    const uint64_t e = (uint64_t)first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_lines) return;
    const uint4 r0 = *(const uint4 *)(lines + e);
    if (r0.x == NOT_FOUND) return;                    // (an overflow line without a target)
    const uint32_t n_in = r0.y, d = r0.w;
    if (list_is_inline(d, n_in) || lines[e].coff == EL_NO_LIST) return;
    const uint32_t wide = list_is_narrow(d) ? 0u : 1u;
    if (!list_has_pivots(wide, n_in)) return;
    const uint32_t np = list_pivot_count(wide), step = list_pivot_step(wide, n_in);
    const uint8_t *p = clist + (uint64_t)lines[e].coff * 16u;
    uint8_t *dst = (uint8_t *)(lines + e) + 24;
    for (uint32_t k = 0; k < np; k++) {
        const uint32_t idx = (k + 1u) * step;
        if (wide) ((uint32_t *)dst)[k] = ((const uint32_t *)p)[idx];
        else ((uint16_t *)dst)[k] = ((const uint16_t *)p)[idx];
    }
}


// tile_sums[b] = 16-byte units of tile b; entry_sums[b] = list entries of tile b
__global__ void __launch_bounds__(CL_BLOCK)
clist_tile_sums_kernel(const ELine *__restrict__ lines, uint32_t nnz, uint32_t nnz_real, uint64_t *tile_sums, uint64_t *entry_sums,
                       uint32_t max_len) {
    __shared__ uint64_t sh[CL_BLOCK], sh2[CL_BLOCK];
    const uint64_t base = (uint64_t)blockIdx.x * CL_TILE + (uint64_t)threadIdx.x * CL_ITEMS;
    uint64_t s = 0, s2 = 0;
    for (int k = 0; k < CL_ITEMS; k++)
        if (base + k < nnz) { s += list_units(lines, base + k, max_len); if (base + k < nnz_real) s2 += lines[base + k].n_in; }
    sh[threadIdx.x] = s;
    sh2[threadIdx.x] = s2;
    __syncthreads();
    for (int st = CL_BLOCK / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { sh[threadIdx.x] += sh[threadIdx.x + st]; sh2[threadIdx.x] += sh2[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { tile_sums[blockIdx.x] = sh[0]; entry_sums[blockIdx.x] = sh2[0]; }
}

// lines[e].coff = exclusive prefix sum of the units (tile_sums already scanned: scan_tile_sums_kernel)
__global__ void __launch_bounds__(CL_BLOCK)
clist_offsets_kernel(ELine *lines, uint32_t nnz, const uint64_t *__restrict__ tile_sums, uint32_t max_len) {
    __shared__ uint64_t sh[CL_BLOCK];
    const int t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * CL_TILE + (uint64_t)t * CL_ITEMS;
    uint32_t loc[CL_ITEMS];
    uint64_t s = 0;
    for (int k = 0; k < CL_ITEMS; k++) {
        loc[k] = base + k < nnz ? list_units(lines, base + k, max_len) : 0u;
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
        if (base + k < nnz) {
            const uint4 r0_ = *(const uint4 *)(lines + base + k);
            lines[base + k].coff = (!list_is_inline(r0_.w, r0_.y) && r0_.y > max_len) ? EL_NO_LIST : (uint32_t)run;
        }
        run += loc[k];
    }
}

// Exclusive prefix sums over the CSR entries of a per-entry count (the weighted lane form's tables): MODE 0 = the list length
// n_in (one float64 per list entry: wlist_kernel), MODE 1 = the recorded chain values of the entry's target row,
// (degree - 1) / CHAIN_CKPT (wckpt_kernel).  Tile sums -> scan_tile_sums_kernel -> offsets.
template <int MODE> __device__ __forceinline__ uint32_t entry_count(const ELine *lines, uint64_t e) {
    const uint4 r0 = *(const uint4 *)(lines + e);
    return MODE == 0 ? r0.y : (r0.w ? (r0.w - 1u) / CHAIN_CKPT : 0u);
}
template <int MODE>
__global__ void __launch_bounds__(CL_BLOCK)
entry_tile_sums_kernel(const ELine *__restrict__ lines, uint32_t nnz, uint64_t *tile_sums) {
    __shared__ uint64_t sh[CL_BLOCK];
    const uint64_t base = (uint64_t)blockIdx.x * CL_TILE + (uint64_t)threadIdx.x * CL_ITEMS;
    uint64_t s = 0;
    for (int k = 0; k < CL_ITEMS; k++)
        if (base + k < nnz) s += entry_count<MODE>(lines, base + k);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = CL_BLOCK / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = sh[0];
}
template <int MODE>
__global__ void __launch_bounds__(CL_BLOCK)
entry_offsets_kernel(const ELine *__restrict__ lines, uint32_t nnz, const uint64_t *__restrict__ tile_sums, unsigned long long *off_out) {
    __shared__ uint64_t sh[CL_BLOCK];
    const int t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * CL_TILE + (uint64_t)t * CL_ITEMS;
    uint32_t loc[CL_ITEMS];
    uint64_t s = 0;
    for (int k = 0; k < CL_ITEMS; k++) {
        loc[k] = base + k < nnz ? entry_count<MODE>(lines, base + k) : 0u;
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
        if (base + k < nnz) off_out[base + k] = run;
        run += loc[k];
    }
}

// Recorded chain values: for every CSR entry e = (u -> v) whose target row has more than CHAIN_CKPT entries, the exact
// float32 value of the reference's cumsum(w / tot) for a walker that arrived by e (prev = u, cur = v), after every
// CHAIN_CKPT-th element -- computed by the walk step's own code (sample_step_weighted, normaliser from the per-entry
// table) -- so that an exact scan can start in the middle of a hub row.  One wavefront per entry; Sigma_v d_v^2 element
// visits, like the normaliser table.
template <bool EXTEND>
__global__ void __launch_bounds__(WAVES_PER_BLOCK *WAVE, EXTEND ? PW_MIN_WAVES - 1 : PW_MIN_WAVES)
wckpt_kernel(WalkArgs a_unused, const uint32_t *__restrict__ edge_row_unused, const unsigned long long *ck_off_unused, float *ck_unused) {
    __shared__ uint32_t s_mask[WAVES_PER_BLOCK][MASK_WORDS];
    __shared__ uint32_t s_in[EXTEND ? WAVES_PER_BLOCK : 1][EXTEND ? MASK_WORDS : 1];
    __shared__ uint32_t s_queue[WAVES_PER_BLOCK][2 * QCAP];
    const int wave = threadIdx.x / WAVE;
    constexpr size_t XARG = (sizeof(WalkArgs) + 7) & ~(size_t)7;
    const uint32_t nnz = PW_KARG(uint32_t, g.nnz);
    const uint64_t e = (uint64_t)blockIdx.x * WAVES_PER_BLOCK + (uint64_t)wave;
    if (e >= nnz) return;
    const sptr<uint32_t> indptr = as_scalar<uint32_t>(PW_KARG(uint64_t, g.indptr));
    const uint32_t cur = uni(as_scalar<uint32_t>(PW_KARG(uint64_t, g.indices))[e]);
    const uint32_t s0 = indptr[cur], d = indptr[cur + 1] - s0;
    if (d <= CHAIN_CKPT) return;
    const uint32_t prev = uni(as_scalar<uint32_t>(kernarg<uint64_t>(XARG))[e]);
    const uint32_t t0 = indptr[prev], dp = indptr[prev + 1] - t0;
    WalkArgs la = reload_walk_args();
    la.g.step_edge = (uint32_t)e;
    float ktot = __uint_as_float(uni(__float_as_uint(la.tot_e[e])));
    const unsigned long long off = as_scalar<unsigned long long>(kernarg<uint64_t>(XARG + 8))[e];
    float *out = (float *)kernarg<uint64_t>(XARG + 16) + off;
    (void)sample_step_weighted<float, false>(la, s_mask[wave], EXTEND ? s_in[EXTEND ? wave : 0] : nullptr, s_queue[wave], cur, true, prev, t0, dp,
                                             0.0, s0, d, &ktot, nullptr, 0u, nullptr, out);
}

// test hook (pw_lane_index_export): the list of every entry, decoded to uint32, at off[e] of `out`
__global__ void __launch_bounds__(256)
lane_index_export_kernel(const ELine *__restrict__ lines, const uint8_t *__restrict__ clist, uint32_t nnz,
                         const uint64_t *__restrict__ off, uint32_t *out) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const ELine &ln = lines[e];
    if (!edge_list_stored(ln.deg, ln.n_in, ln.coff)) {   // partial index: this list was left out -- every entry reads 0xffffffff
        for (uint32_t i = 0; i < ln.n_in; i++) out[off[e] + i] = 0xffffffffu;
        return;
    }
    const uint8_t *p = list_base(lines, clist, (uint32_t)e, ln.deg, ln.n_in, ln.coff);
    const bool narrow = list_is_narrow(ln.deg);
    for (uint32_t i = 0; i < ln.n_in; i++) out[off[e] + i] = narrow ? (uint32_t)((const uint16_t *)p)[i] : ((const uint32_t *)p)[i];
}

}  // namespace pw
