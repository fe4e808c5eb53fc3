// pecanpy_amd.hip -- C ABI (include/pecanpy_amd.h) over the gfx950 walk kernels.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// (-ffp-contract=off is mandatory: the reference's Numba code never fuses a*b+c, and the
// bit-exact float chain depends on separately rounded operations.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <random>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pecanpy_amd.h"
#include "aux_kernels.hip.h"
#include "edgelist.hpp"
#include "thresholds.hpp"
#include "mtjump.hpp"
#include "seqscan.h"
#include "walk_dense.hip.h"
#include "walk_dense_w.hip.h"
#include "walk_seq.hip.h"
#include "walk_sparse.hip.h"
#include "walk_lanes.hip.h"
#include "sgns.hip.h"

#define PW_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_err;

// pw_graph::counters layout: [0] job counter [1..4] stats [5] changed count [6] redo count [7] list entries read
// by the lane kernel [8] ambiguous steps (float chain) of the lane kernel [9] first bad start [10] ambiguous steps the
// per-lane chain left to the wave-cooperative chain (rounding ties)
constexpr int N_COUNTERS = 48;   // ... [32] parked walks (lane kernel's chain queue; a cache line of its own)

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

// opt-in switches documented as NAME=1: on when set to a non-zero number (NAME=0 and an empty value leave them off)
bool env_on(const char *name) {
    const char *v = getenv(name);
    return v != nullptr && atoi(v) != 0;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(PW_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    } while (0)

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e != hipSuccess) return fail(PW_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        cap = n;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct pw_graph {
    int device = 0;
    int kind = 0;  // 0 = CSR, 1 = dense
    uint32_t n_nodes = 0, nnz = 0;
    bool unit = false;  // all weights are 1.0f (data not stored)
    uint32_t max_degree = 0;
    uint32_t *d_indptr = nullptr, *d_indices = nullptr;
    uint32_t *d_hasnbr = nullptr;                       // bit v: vertex v has neighbours (stream offsets; built by the first call)
    void *d_data = nullptr;          // float32 (CSR graphs) or float64 (dense graphs); null when unit
    float *d_thr = nullptr;
    uint64_t *d_adjbits = nullptr;   // dense graphs: bit-packed adjacency rows
    uint32_t *d_deg = nullptr;       // dense graphs: row degrees
    bool bits_only = false;          // dense graph created from packed bits: no compressed rows
    bool dense_nonneg = false;       // weighted dense graph: every stored value is finite and > 0 (walk_dense_w.hip.h)
    uint32_t *d_foff = nullptr;      // CSR graphs: per-row membership filters (offsets, bits)
    uint64_t *d_fbits = nullptr;
    uint2 *d_kf = nullptr;           // CSR graphs: (neighbour id, filter word) per CSR entry
    uint64_t *d_tab_off = nullptr, *d_slots = nullptr;  // CSR graphs: adjacency index (exact lookups)
    uint4 *d_vrec = nullptr;                            // CSR graphs: per-vertex record (row start, degree, filter, index)
    pw::ELine *d_lines = nullptr;                       // lane index (walk_lanes.hip.h): 64-byte edge line per CSR entry; its first 16 bytes
                                                        // {neighbour, common-neighbour count, reverse position, degree} also serve walk_kernel's lazy step
    uint8_t *d_clist = nullptr;                         // lane index: the lists too long for their edge line
    uint64_t clist_bytes = 0, line_bytes = 0;
    uint64_t fbits_words = 0, slot_words = 0;           // sizes of d_fbits / d_slots (pw_graph_replicate copies the buffers)
    bool vlines = false;                                // lines[nnz + v]: the line of vertex v's mirrored overflow read
    // TWIN (round 6): a second call context on the SAME device that aliases this handle's graph, index and per-(p, q) tables
    // (own streams, counters, queues, stream buffers): the weighted lane form walks the two halves of a job array on the two
    // contexts side by side, so that one half's eager kernel runs beside the other half's lane round (simulate_twin)
    pw_graph *twin = nullptr;
    bool alias = false;                                 // this handle IS such a twin: the shared buffers are not its to free
    bool twin_active = false;                           // the current call runs on both contexts: each takes a share of the GPU
    std::function<void()> on_tables_ready;              // simulate_twin: called once the call's per-(p, q) tables are in place
    bool has_loop = false;                              // the CSR has a self loop (unit graphs: lists fixed up, wave kernel's lazy step off)
    bool lanes_off = false;                             // PECANPY_AMD_NO_LANES was set when the handle was created: the index is
                                                        // built (the wave kernel's lazy step reads it) but the lane kernel is not used
    float *d_tot_e = nullptr, *d_tot_v = nullptr;       // weighted CSR graphs: per-edge / per-vertex normalisers
    float *d_utot = nullptr;                            // unit graphs, 1/p or 1/q not a power of two: row total per arriving line
    float utot_wo = 0, utot_wp = 0;                     // ... built for these biases (0: none)
    bool utot_failed = false;
    // weighted lane form (walk_lanes.hip.h: WEIGHTED): base values, their per-row float64 prefix sums, per-entry delta prefix sums
    float *d_wb = nullptr;
    pw::PrefixPair *d_wpq = nullptr;
    double *d_wdl = nullptr, *d_wl_dprev = nullptr;
    unsigned long long *d_wl_off = nullptr;
    uint32_t *d_wedge_row = nullptr;                    // ... source vertex of every CSR entry
    pw::PrefixPair *d_wp1 = nullptr;                    // ... per-row prefix sums of the raw weights (first steps; graph-static)
    unsigned long long *d_wck_off = nullptr;            // ... recorded chain values (wckpt_kernel): first record of entry e
    float *d_wck = nullptr;
    uint64_t wdl_cap = 0, wck_cap = 0;
    double wl_p = 0, wl_q = 0;
    int wl_extend = -1;
    uint64_t wl_thr_version = 0;
    bool wl_failed = false;
    bool wl_active = false;                             // the current call may run the weighted lane form (tables are there)
    bool wl_used = false;                               // ... and did
    double tot_p = 0, tot_q = 0;                        // ... built for these parameters
    int tot_extend = -1;                                // -1: none yet
    uint64_t tot_thr_version = 0, thr_version = 0;      // thresholds uploaded since the table was built?
    bool tot_failed = false;
    double tot_build_ms = 0;
    double param_ms_call = 0;                           // (p, q)-dependent index time of the current call
    uint64_t n_clist = 0;
    uint32_t list_max_len = 0xffffffffu;                // partial index: lists longer than this were left out (EL_NO_LIST)
    double index_build_ms = 0;                          // device time of all index KERNELS of pw_csr_create (event pairs around them)
    double create_wall_ms = 0;                          // wall clock of pw_csr_create: runtime start-up, host passes, H2D, allocations, kernels
    uint64_t index_bytes = 0;                           // device bytes of the membership / lane index
    uint32_t words_per_row = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // side stream: the zero-fill of the walk matrix runs under the stream expansion
    hipEvent_t ev_side = nullptr;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> round_ev;          // lane rounds: a pair of events per round, read once behind the last round (no sync per round)
    // pw_simulate() walks a large job array in parts: the stream of the WHOLE array is expanded once, and while this is
    // set expand_stream() serves the parts' sub-ranges from g->rng as it stands (never kept across API calls)
    struct { bool valid = false, user = false; uint32_t seed = 0; uint64_t first_block = 0, n_blocks = 0; } rng_hold;   // (user: pw_stream_hold)
    hipStream_t copy_stream = nullptr;     // pw_simulate: the D2H of one part of the walk matrix runs here, under the walks of the next
    static constexpr int N_STAGE = 12;
    hipEvent_t ev_copy[N_STAGE] = {};
    void *stage[N_STAGE] = {};             // pinned staging buffers of pw_simulate's copy out
    uint32_t *seed_state = nullptr;        // pinned: the seed's MT19937 state on its way to the device
    double lane_ms = 0;              // lane kernel time of the current call
    int n_cu = 0;
    // scratch reused across calls
    DevBuf<uint64_t> stream_off, tile_sums;
    DevBuf<double> rng;
    DevBuf<uint32_t> mt_state, changed, redo;
    DevBuf<pw::SuspRec> susp[2];      // lane kernel: walks parked for the float chain (two queues, swapped per round)
    uint32_t lane_rounds = 0;         // lane kernel launches of the last call
    uint64_t call_runnable = 0;       // whole-array call: jobs whose start has neighbours (the stream's nominal draws / walk_length)
    // generator states of recent calls, keyed by (seed, first block, blocks per generator, generators): a repeated
    // call (every pass of a benchmark, every chunk of a sharded run) skips the ~20 sequential jump-ahead launches
    struct MtCache { bool valid = false; uint32_t seed = 0, n_gen = 0; uint64_t first_block = 0; int per_gen_log = 0; uint64_t stamp = 0;
                     DevBuf<uint32_t> states; } mt_cache[8];
    uint64_t mt_stamp = 0;
    // verification mode of the lane kernel (PECANPY_AMD_VERIFY_TIGHT=1): steps the interval decision settled
    DevBuf<pw::VerRec> ver, ver_bad;
    DevBuf<uint32_t> ver_jobs;        // sampled verification (production): walks of mismatching records, redone by walk_kernel
    uint64_t ver_checked = 0, ver_mismatch = 0, ver_dropped = 0, ver_ties = 0;   // ... of the current call
    DevBuf<uint64_t> jump_table;  // MtJump::pow2_table() on the device
    DevBuf<uint32_t> jump_tmp;    // partial results of jumps whose taps are split over several workgroups (kept zeroed)
    bool jump_table_ready = false;
    DevBuf<unsigned long long> counters;  // [0] job counter [1..4] stats [5] changed count
    // alias tables (PreComp modes)
    DevBuf<uint64_t> alias_indptr;
    DevBuf<uint32_t> alias_j, alias_s, alias_l, edge_row;
    DevBuf<float> alias_q, probs_scratch;
    uint64_t n_alias = 0;
    int alias_kind = -1;  // -1 none, 0 second order, 1 first order
    double alias_p = 0, alias_q_param = 0;
    int alias_extend = 0;
};

namespace {

int set_device(const pw_graph *g) {
    HIP_TRY(hipSetDevice(g->device));
    return 0;
}

uint32_t os_seed() {
    std::random_device rd;
    return (uint32_t)rd();
}

// host emulation of seq_scan (walk_sparse.hip.h) used by the self test: identical arithmetic,
// `chunk` elements per pass instead of one wavefront pass.
template <typename T>
uint32_t seqscan_emulate(const T *x, uint32_t n, double r, bool use_target, uint32_t chunk, T *sum) {
    using B = pw::Binade<T>;
    using U = typename B::UInt;
    T c = (T)0;
    uint32_t k = 0;
    std::vector<pw::Inc<T>> f(chunk);
    while (k < n) {
        const int eb = B::eb_of(c);
        const U C = B::sig_of(c);
        const U Tt = use_target ? B::threshold(r, eb) : B::TOP;
        uint32_t m = n - k < chunk ? n - k : chunk;
        pw::Inc<T> run = {0, 0};
        uint32_t first = m;
        U Cprev = C, Cn = C;
        for (uint32_t l = 0; l < m; l++) {
            f[l] = B::quantize(x[k + l], eb);
            pw::Inc<T> incl = l ? B::compose(run, f[l]) : f[l];
            U Ci = B::apply(C, incl);
            if (Ci >= Tt) { first = l; Cn = Ci; break; }
            Cprev = Ci;
            run = incl;
        }
        if (first == m) {
            c = B::make(Cprev, eb);
            k += m;
            continue;
        }
        uint32_t kf = k + first;
        if (Cn < B::TOP) { c = B::make(Cn, eb); *sum = c; return kf; }
        c = B::make(Cprev, eb) + x[kf];
        if (use_target && (double)c >= r) { *sum = c; return kf; }
        k = kf + 1;
    }
    *sum = c;
    return n;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
PW_EXPORT const char *pw_version(void) { return "pecanpy_amd 0.1.0 (gfx950)"; }
PW_EXPORT const char *pw_last_error(void) { return g_err.c_str(); }

PW_EXPORT int pw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

PW_EXPORT int pw_warmup(int device, double *ms) {
    const auto t0 = std::chrono::steady_clock::now();
    const int n = pw_device_count();
    if (n <= 0) return fail(PW_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(PW_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIP_TRY(hipStreamDestroy(s));
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PW_OK;
}

PW_EXPORT void pw_graph_destroy(pw_graph *g) {
    if (!g) return;
    if (g->twin) { pw_graph_destroy(g->twin); g->twin = nullptr; }
    (void)hipSetDevice(g->device);
    if (g->alias) {   // the graph, its index and the per-(p, q) tables belong to the handle this one aliases
        g->d_indptr = g->d_indices = nullptr; g->d_data = nullptr; g->d_thr = nullptr; g->d_adjbits = nullptr; g->d_deg = nullptr;
        g->d_foff = nullptr; g->d_fbits = nullptr; g->d_kf = nullptr; g->d_tab_off = nullptr; g->d_slots = nullptr; g->d_vrec = nullptr;
        g->d_lines = nullptr; g->d_clist = nullptr; g->d_tot_e = nullptr; g->d_tot_v = nullptr; g->d_utot = nullptr; g->d_hasnbr = nullptr;
        g->d_wb = nullptr; g->d_wpq = nullptr; g->d_wdl = nullptr; g->d_wl_dprev = nullptr; g->d_wl_off = nullptr; g->d_wedge_row = nullptr;
        g->d_wp1 = nullptr; g->d_wck_off = nullptr; g->d_wck = nullptr;
    }
    if (g->d_indptr) (void)hipFree(g->d_indptr);
    if (g->d_indices) (void)hipFree(g->d_indices);
    if (g->d_data) (void)hipFree(g->d_data);
    if (g->d_thr) (void)hipFree(g->d_thr);
    if (g->d_adjbits) (void)hipFree(g->d_adjbits);
    if (g->d_deg) (void)hipFree(g->d_deg);
    if (g->d_foff) (void)hipFree(g->d_foff);
    if (g->d_fbits) (void)hipFree(g->d_fbits);
    if (g->d_kf) (void)hipFree(g->d_kf);
    if (g->d_tab_off) (void)hipFree(g->d_tab_off);
    if (g->d_slots) (void)hipFree(g->d_slots);
    if (g->d_vrec) (void)hipFree(g->d_vrec);
    if (g->d_lines) (void)hipFree(g->d_lines);
    if (g->d_clist) (void)hipFree(g->d_clist);
    if (g->d_tot_e) (void)hipFree(g->d_tot_e);
    if (g->d_utot) (void)hipFree(g->d_utot);
    if (g->d_hasnbr) (void)hipFree(g->d_hasnbr);
    if (g->d_tot_v) (void)hipFree(g->d_tot_v);
    for (void *q : {(void *)g->d_wb, (void *)g->d_wpq, (void *)g->d_wdl, (void *)g->d_wl_dprev, (void *)g->d_wl_off, (void *)g->d_wedge_row, (void *)g->d_wp1,
                    (void *)g->d_wck_off, (void *)g->d_wck})
        if (q) (void)hipFree(q);
    g->redo.release();
    g->susp[0].release();
    g->susp[1].release();
    g->ver.release();
    g->ver_bad.release();
    g->ver_jobs.release();
    g->stream_off.release();
    g->tile_sums.release();
    g->rng.release();
    g->mt_state.release();
    for (auto &c : g->mt_cache) c.states.release();
    g->jump_table.release();
    g->jump_tmp.release();
    g->changed.release();
    g->counters.release();
    g->alias_indptr.release();
    g->alias_j.release();
    g->alias_s.release();
    g->alias_l.release();
    g->edge_row.release();
    g->alias_q.release();
    g->probs_scratch.release();
    for (auto &b : g->stage)
        if (b) (void)hipHostFree(b);
    if (g->seed_state) (void)hipHostFree(g->seed_state);
    for (auto &e : g->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : g->round_ev)
        if (e) (void)hipEventDestroy(e);
    if (g->ev_side) (void)hipEventDestroy(g->ev_side);
    for (auto &e : g->ev_copy)
        if (e) (void)hipEventDestroy(e);
    if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
    if (g->stream2) (void)hipStreamDestroy(g->stream2);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

static int graph_common_init(pw_graph *g, int device) {
    const bool dbg = getenv("PECANPY_AMD_CREATE_DEBUG") != nullptr;
    auto nowms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = nowms();
    auto lap = [&](const char *what) { if (dbg) { const double t = nowms(); fprintf(stderr, "[create]   init: %-28s %8.2f ms\n", what, t - t0); t0 = t; } };
    int n = pw_device_count();
    if (n <= 0) return fail(PW_ERR_NO_DEVICE, "no HIP device visible (libpecanpy_amd needs a GPU; there is no CPU fallback)");
    if (device < 0 || device >= n) return fail(PW_ERR_INVALID, "device index out of range");
    g->device = device;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    g->n_cu = prop.multiProcessorCount;
    lap("device count / properties");
    HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    lap("stream");
    HIP_TRY(hipStreamCreateWithFlags(&g->stream2, hipStreamNonBlocking));
    lap("stream2");
    HIP_TRY(hipEventCreateWithFlags(&g->ev_side, hipEventDisableTiming));
    for (auto &e : g->ev) HIP_TRY(hipEventCreate(&e));
    lap("events");
    HIP_TRY(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
    for (auto &e : g->ev_copy) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    lap("copy stream + events");
    return 0;
}

static pw::CsrDev csr_dev(const pw_graph *g);

// Lane-kernel index (walk_lanes.hip.h): one 64-byte edge line per CSR entry (the record of the edge + its list of
// common-neighbour positions when that is short) and the overflow array of the longer lists.  One set intersection per
// adjacent pair, row of the larger endpoint in LDS (lane_lists_kernel: count pass, offsets, fill pass).  Skipped (the
// wave-per-walk kernel then serves every call, eager step) when lines + lists would take more than half of the free
// device memory.  `indptr` = the caller's host array.
// Work items of the lane-index build: one per (vertex, segment of its row, chunk of its neighbours), the longest rows
// first.  Needs the host indptr only: pw_csr_create computes them on a helper thread while the runtime starts up and the
// CSR travels to the device.
struct LaneWorkItems {
    std::vector<pw::LaneBuildItem> small, large;
    uint64_t segcnt_total = 0;
    std::vector<uint32_t> vm0;      // per vertex: base of its per-segment counts (rows longer than LB_SEG; 0 otherwise)
};
static void make_lane_work_items(const uint32_t *indptr, const uint32_t *indices, uint32_t n_nodes, LaneWorkItems &w) {
    const uint32_t JCHUNK = 16384;
    w.vm0.assign((size_t)n_nodes + 1, 0u);
    // The vertex pass on up to four threads, a contiguous range each (round 6: since the library's start-up no longer hides it --
    // pw_warmup -- this pass, 185 ms on one thread at RMAT-22 with its scattered look-ups behind the one-neighbour rows, was what
    // pw_csr_create waited for): the rows of up to LB_SMALL entries become items at once, the longer ones are listed and turned
    // into items below, in vertex order (their per-segment count bases are a running sum).
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt >= 8 ? 4u : (nt >= 4 ? 2u : 1u);
    if (n_nodes < (1u << 16)) nt = 1u;
    std::vector<std::vector<pw::LaneBuildItem>> smalls(nt);
    std::vector<std::vector<uint32_t>> bigs(nt);
    auto scan = [&](unsigned t) {
        const uint32_t h0 = (uint32_t)((uint64_t)n_nodes * t / nt), h1 = (uint32_t)((uint64_t)n_nodes * (t + 1) / nt);
        std::vector<pw::LaneBuildItem> &sm = smalls[t];
        sm.reserve((size_t)(h1 - h0) / 2 + 16);
        for (uint32_t h = h0; h < h1; h++) {
            const uint32_t d = indptr[h + 1] - indptr[h];
            if (d < 2) {
                // one neighbour k: N(h) & N(k) = {k} & N(k) is empty -- unless k has a SELF LOOP (then k's position in its own
                // row is the one common neighbour of h and k; the pair is h's to take only when k -> h is no edge, but an item
                // too many costs nothing).  (The indices are validated later, on the device: nothing is assumed of them here.)
                if (d == 1 && indices) {
                    const uint32_t k = indices[indptr[h]];
                    if (k != h && k < n_nodes) {
                        const uint32_t *row = indices + indptr[k], *end = indices + indptr[k + 1];
                        if (std::binary_search(row, end, k)) sm.push_back({h, 0u, 1u, 0u, 0u, d});
                    }
                }
                continue;
            }
            if (d <= (uint32_t)pw::LB_SMALL) sm.push_back({h, 0u, 1u, 0u, 0u, d});
            else bigs[t].push_back(h);
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) {
            try { th.emplace_back(scan, t); } catch (const std::system_error &) { scan(t); }
        }
        scan(0);
        for (auto &x : th) x.join();
    }
    size_t n_small = 0;
    for (auto &v : smalls) n_small += v.size();
    w.small.reserve(n_small);
    for (auto &v : smalls) { w.small.insert(w.small.end(), v.begin(), v.end()); std::vector<pw::LaneBuildItem>().swap(v); }
    for (unsigned t = 0; t < nt; t++)
        for (uint32_t h : bigs[t]) {
            const uint32_t d = indptr[h + 1] - indptr[h];
            const uint32_t nseg = (d + pw::LB_SEG - 1) / pw::LB_SEG;
            uint32_t m0 = 0;
            if (nseg > 1) { m0 = (uint32_t)w.segcnt_total; w.segcnt_total += (uint64_t)d * nseg; w.vm0[h] = m0; }
            for (uint32_t sg = 0; sg < nseg; sg++)
                for (uint32_t j0 = 0; j0 < d; j0 += JCHUNK) w.large.push_back({h, sg, nseg, m0, j0, j0 + JCHUNK < d ? j0 + JCHUNK : d});
        }
    // the longest rows first (their workgroups run longest)
    std::stable_sort(w.large.begin(), w.large.end(), [&](const pw::LaneBuildItem &x, const pw::LaneBuildItem &y) {
        return indptr[x.h + 1] - indptr[x.h] > indptr[y.h + 1] - indptr[y.h];
    });
}

// device time of a group of index kernels: g->ev[2] / g->ev[3] around it, added to g->index_build_ms once it has run
#define INDEX_KERNELS_BEGIN(g) (void)hipEventRecord((g)->ev[2], (g)->stream)
#define INDEX_KERNELS_END(g)                                                                                     \
    do {                                                                                                         \
        (void)hipEventRecord((g)->ev[3], (g)->stream);                                                           \
        if (hipEventSynchronize((g)->ev[3]) == hipSuccess) {                                                     \
            float ms_ = 0;                                                                                       \
            if (hipEventElapsedTime(&ms_, (g)->ev[2], (g)->ev[3]) == hipSuccess) (g)->index_build_ms += ms_;     \
        }                                                                                                        \
    } while (0)

static int build_lane_index(pw_graph *g, LaneWorkItems &items, const uint32_t *d_edge_row, bool has_loop, void *pre_lines = nullptr,
                            uint64_t pre_bytes = 0) {
    // (pre_lines: the edge lines' memory, allocated by pw_csr_create's helper thread beside the host pass and the first kernels
    //  -- device memory a fresh box hands out for the first time costs up to ~25 ms per GB of hipMalloc; owned by g from here on)
    if (pre_lines) g->d_lines = (pw::ELine *)pre_lines;
    auto no_index = [&]() { if (g->d_lines) { (void)hipFree(g->d_lines); g->d_lines = nullptr; } return 0; };
    if (!g->nnz) return no_index();
    const uint32_t nnz = g->nnz, n_nodes = g->n_nodes;
    const bool dbg = getenv("PECANPY_AMD_CREATE_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = now();
    auto stamp = [&](const char *what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(g->stream);
        const double t = now();
        fprintf(stderr, "[create]   lane index: %-22s %8.2f ms\n", what, (t - t_last) * 1e3);
        t_last = t;
    };
    std::vector<pw::LaneBuildItem> &small = items.small, &large = items.large;
    const uint64_t segcnt_total = items.segcnt_total;
    if (segcnt_total >= 0xffffffffull) return no_index();   // (rows this long and this many: no lane index)
    // + one OVERFLOW line per vertex (walk_lanes.hip.h: vline_init_kernel): the pair of its mirrored choice == degree read
    const bool vlines = (uint64_t)nnz + n_nodes < 0xffffffffull && !getenv("PECANPY_AMD_NO_VLINES");
    const uint32_t n_lines = vlines ? nnz + n_nodes : nnz;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    stamp("hipMemGetInfo");
    const uint64_t line_bytes = (uint64_t)n_lines * sizeof(pw::ELine) + 64;
    if (g->d_lines && pre_bytes != line_bytes) no_index();   // (cannot happen: the same formula sized it)
    if (!g->d_lines && line_bytes > free_b / 2) return 0;
    pw::LaneBuildItem *d_small = nullptr, *d_large = nullptr;
    uint32_t *d_segcnt = nullptr;
    uint64_t *d_tiles = nullptr, *d_etiles = nullptr;
    uint32_t *d_log = nullptr, *d_seglo = nullptr, *d_vm0 = nullptr;    // LOGGED build (below)
    unsigned long long *d_logoff = nullptr;
    const uint64_t n_tiles = ((uint64_t)n_lines + pw::CL_TILE - 1) / pw::CL_TILE;
    auto cleanup = [&]() {
        for (void *q : {(void *)d_small, (void *)d_large, (void *)d_segcnt, (void *)d_tiles, (void *)d_etiles, (void *)d_log,
                        (void *)d_seglo, (void *)d_vm0, (void *)d_logoff})
            if (q) (void)hipFree(q);
    };
    auto drop = [&](int rc) {   // no lane index
        cleanup();
        if (g->d_lines) (void)hipFree(g->d_lines);
        if (g->d_clist) (void)hipFree(g->d_clist);
        g->d_lines = nullptr;
        g->d_clist = nullptr;
        (void)hipGetLastError();
        return rc;
    };
    hipError_t e = g->d_lines ? hipSuccess : hipMalloc((void **)&g->d_lines, line_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_small, sizeof(pw::LaneBuildItem) * (small.size() + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_large, sizeof(pw::LaneBuildItem) * (large.size() + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_segcnt, sizeof(uint32_t) * (size_t)(segcnt_total + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_tiles, sizeof(uint64_t) * (n_tiles + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_etiles, sizeof(uint64_t) * (n_tiles + 1));
    if (e == hipSuccess && !small.empty())
        e = hipMemcpyAsync(d_small, small.data(), sizeof(pw::LaneBuildItem) * small.size(), hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess && !large.empty())
        e = hipMemcpyAsync(d_large, large.data(), sizeof(pw::LaneBuildItem) * large.size(), hipMemcpyHostToDevice, g->stream);
    if (e != hipSuccess) return drop(e == hipErrorOutOfMemory ? 0 : fail(PW_ERR_HIP, std::string("lane index: ") + hipGetErrorString(e)));
    stamp("hipMalloc + item upload");
    pw::CsrDev c = csr_dev(g);
    pw::LaneBuildArgs ba;
    ba.indptr = g->d_indptr;
    ba.indices = g->d_indices;
    ba.lines = g->d_lines;
    ba.clist = nullptr;
    ba.segcnt = d_segcnt;
    ba.max_len = 0xffffffffu;
    ba.log = nullptr;
    ba.log_off = nullptr;
    ba.seglo = nullptr;
    ba.logged = 0;
    INDEX_KERNELS_BEGIN(g);
    hipLaunchKernelGGL(pw::eline_init_kernel, dim3((unsigned)(((uint64_t)nnz + 255) / 256)), dim3(256), 0, g->stream, c, d_edge_row, g->d_lines);
    if (vlines) hipLaunchKernelGGL(pw::vline_init_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, g->stream, c, g->d_lines);
    const unsigned vgrid = (unsigned)(((uint64_t)n_nodes * pw::WAVE + 255) / 256);
    // LOGGED build (round 5; walk_lanes.hip.h: LaneBuildArgs): the COUNT pass keeps its matches in a log -- one block of d_k
    // four-byte slots per pair, the pair's upper bound: 50 GB of address space at RMAT-22 of which the matches touch 5 -- and
    // lane_scatter_kernel copies them to their places once the offsets are known, so the intersection of the long rows runs
    // ONCE (the FILL pass streamed the same 50 GB of neighbour rows through LDS a second time: 61 of the index's 172 ms).
    // (The lists of the overflow lines keep their two lookup passes: a per-vertex scatter measured 15 ms against 6.4.)
    // OPT-IN (PECANPY_AMD_INDEX_LOGGED=1), and only when the log fits a third of the free memory: device memory that no process
    // has used since the box started costs ~23 ms per GB of hipMalloc (the driver clears it on first use; measured on fresh
    // MI355X boxes: 50 GB of log = 1.0-1.3 s, the lists' 5.2 GB = 121 ms), memory that has been used and freed microseconds.  On a
    // fresh box the log costs twenty times what it saves; in a long-lived service whose device memory has been touched it saves
    // 45 ms per graph.  The default is the two-pass build.
    uint64_t log_slots = 0;
    bool logged = false;
    const uint64_t nnz_tiles = ((uint64_t)nnz + pw::CL_TILE - 1) / pw::CL_TILE;
    if (env_on("PECANPY_AMD_INDEX_LOGGED") && !getenv("PECANPY_AMD_INDEX_TWO_PASS") && !has_loop) {   // (self loops: the two-pass build)
        e = hipMalloc((void **)&d_logoff, sizeof(unsigned long long) * ((size_t)nnz + 1));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pw::log_tile_sums_kernel, dim3((unsigned)nnz_tiles), dim3(pw::CL_BLOCK), 0, g->stream, g->d_lines, g->d_indptr,
                               d_edge_row, nnz, d_tiles);
            hipLaunchKernelGGL(pw::scan_tile_sums_kernel, dim3(1), dim3(pw::SCAN_BLOCK), 0, g->stream, d_tiles, nnz_tiles);
            hipLaunchKernelGGL(pw::log_offsets_kernel, dim3((unsigned)nnz_tiles), dim3(pw::CL_BLOCK), 0, g->stream, g->d_lines, g->d_indptr,
                               d_edge_row, nnz, d_tiles, d_logoff);
            e = hipMemcpyAsync(&log_slots, d_tiles + nnz_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream);
        }
        INDEX_KERNELS_END(g);                       // (the allocations below are host time, not index-kernel time)
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        size_t free_now = 0, total_now = 0;
        (void)hipMemGetInfo(&free_now, &total_now);
        if (e == hipSuccess && log_slots && (log_slots + 64) * sizeof(uint32_t) <= free_now / 3) {
            e = hipMalloc((void **)&d_log, (log_slots + 64) * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMalloc((void **)&d_seglo, sizeof(uint32_t) * (size_t)(segcnt_total + 1));
            if (e == hipSuccess) e = hipMalloc((void **)&d_vm0, sizeof(uint32_t) * ((size_t)n_nodes + 1));
            if (e == hipSuccess) e = hipMemcpyAsync(d_vm0, items.vm0.data(), sizeof(uint32_t) * ((size_t)n_nodes + 1), hipMemcpyHostToDevice, g->stream);
            logged = e == hipSuccess;
        }
        if (!logged) {   // (no room: the two passes)
            (void)hipGetLastError();
            for (void **q : {(void **)&d_log, (void **)&d_seglo, (void **)&d_vm0, (void **)&d_logoff})
                if (*q) { (void)hipFree(*q); *q = nullptr; }
        } else {
            ba.log = d_log;
            ba.log_off = d_logoff;
            ba.seglo = d_seglo;
        }
        stamp("log offsets + allocation");
        INDEX_KERNELS_BEGIN(g);
    }
    auto lists = [&](bool fill) {
        if (!large.empty()) {
            // wavefronts per workgroup of the long rows' kernel (PECANPY_AMD_INDEX_THREADS = 256 / 512 / 1024 lanes): a workgroup holds
            // 8192 row positions in LDS (33 KB: four workgroups per CU) and its wavefronts wait for the 13 dependent LDS reads of
            // a search most of the time (SQ_WAIT_ANY 74 % of the wave cycles at 3.7 wavefronts per SIMD, profiles/r06_pmc_headline.txt)
            // -- with 512 lanes the same LDS serves EIGHT wavefronts per SIMD instead of four (26-32 VGPRs: the registers allow
            // it).  RMAT-22, same box: index kernels 170.6 -> 117.0 ms with 512 (the default since round 6's fourth session),
            // 122.7 with 1024; COUNT + FILL of the long rows 130 -> ~77 ms.
            static const int ith = getenv("PECANPY_AMD_INDEX_THREADS") ? atoi(getenv("PECANPY_AMD_INDEX_THREADS")) : 512;
            const dim3 lg((unsigned)large.size());
            if (ith == 512) {
                if (fill) hipLaunchKernelGGL((pw::lane_lists_kernel<512, pw::LB_SEG, true>), lg, dim3(512), 0, g->stream, ba, d_large);
                else if (logged) hipLaunchKernelGGL((pw::lane_lists_kernel<512, pw::LB_SEG, false, true>), lg, dim3(512), 0, g->stream, ba, d_large);
                else hipLaunchKernelGGL((pw::lane_lists_kernel<512, pw::LB_SEG, false>), lg, dim3(512), 0, g->stream, ba, d_large);
            } else if (ith == 1024) {
                if (fill) hipLaunchKernelGGL((pw::lane_lists_kernel<1024, pw::LB_SEG, true>), lg, dim3(1024), 0, g->stream, ba, d_large);
                else if (logged) hipLaunchKernelGGL((pw::lane_lists_kernel<1024, pw::LB_SEG, false, true>), lg, dim3(1024), 0, g->stream, ba, d_large);
                else hipLaunchKernelGGL((pw::lane_lists_kernel<1024, pw::LB_SEG, false>), lg, dim3(1024), 0, g->stream, ba, d_large);
            } else {
                if (fill) hipLaunchKernelGGL((pw::lane_lists_kernel<256, pw::LB_SEG, true>), lg, dim3(256), 0, g->stream, ba, d_large);
                else if (logged) hipLaunchKernelGGL((pw::lane_lists_kernel<256, pw::LB_SEG, false, true>), lg, dim3(256), 0, g->stream, ba, d_large);
                else hipLaunchKernelGGL((pw::lane_lists_kernel<256, pw::LB_SEG, false>), lg, dim3(256), 0, g->stream, ba, d_large);
            }
        }
        if (!small.empty()) {
            if (fill) hipLaunchKernelGGL((pw::lane_lists_kernel<64, pw::LB_SMALL, true>), dim3((unsigned)small.size()), dim3(64), 0, g->stream, ba, d_small);
            else if (logged) hipLaunchKernelGGL((pw::lane_lists_kernel<64, pw::LB_SMALL, false, true>), dim3((unsigned)small.size()), dim3(64), 0, g->stream, ba, d_small);
            else hipLaunchKernelGGL((pw::lane_lists_kernel<64, pw::LB_SMALL, false>), dim3((unsigned)small.size()), dim3(64), 0, g->stream, ba, d_small);
        }
        if (vlines) {
            if (fill) hipLaunchKernelGGL(pw::vline_lists_kernel<true>, dim3(vgrid), dim3(256), 0, g->stream, c, g->d_lines, g->d_clist);
            else hipLaunchKernelGGL(pw::vline_lists_kernel<false>, dim3(vgrid), dim3(256), 0, g->stream, c, g->d_lines, (uint8_t *)nullptr);
        }
    };
    lists(false);
    uint64_t units = 0, entries = 0;
    auto offsets = [&](uint32_t max_len) -> hipError_t {   // list offsets (16-byte units) of the lists of at most max_len entries
        hipLaunchKernelGGL(pw::clist_tile_sums_kernel, dim3((unsigned)n_tiles), dim3(pw::CL_BLOCK), 0, g->stream, g->d_lines, n_lines, nnz, d_tiles, d_etiles, max_len);
        hipLaunchKernelGGL(pw::scan_tile_sums_kernel, dim3(1), dim3(pw::SCAN_BLOCK), 0, g->stream, d_tiles, n_tiles);
        hipLaunchKernelGGL(pw::scan_tile_sums_kernel, dim3(1), dim3(pw::SCAN_BLOCK), 0, g->stream, d_etiles, n_tiles);
        hipLaunchKernelGGL(pw::clist_offsets_kernel, dim3((unsigned)n_tiles), dim3(pw::CL_BLOCK), 0, g->stream, g->d_lines, n_lines, d_tiles, max_len);
        hipError_t e2 = hipGetLastError();
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(&units, d_tiles + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(&entries, d_etiles + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream);
        return e2;
    };
    e = offsets(0xffffffffu);
    INDEX_KERNELS_END(g);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) return drop(fail(PW_ERR_HIP, std::string("lane index (count pass): ") + hipGetErrorString(e)));
    stamp("count pass + offsets");
    (void)hipMemGetInfo(&free_b, &total_b);
    // BYTE BUDGET of the overflow lists: PECANPY_AMD_INDEX_BUDGET (bytes) or half of the free device memory (room for the
    // stream and the walk matrix).  Over budget the index is PARTIAL: the longest lists are left out (EL_NO_LIST) -- the
    // longest length whose cumulative bytes fit, from a histogram of the units by list length -- and a step that arrives
    // by such an entry is taken by lanes_eager_kernel (membership searched, as without an index) while the walk stays in
    // the lane kernel.  The reference's SparseOTF is O(nnz) memory (pecanpy.py:510-561); this bounds the distance.
    uint64_t budget = free_b / 2;
    if (const char *be = getenv("PECANPY_AMD_INDEX_BUDGET")) budget = (uint64_t)strtoull(be, nullptr, 10);
    uint32_t max_len = 0xffffffffu;
    if (units * 16 + 64 > budget || units >= 0xffffffffull) {
        const uint32_t HL = 1u << 17;
        unsigned long long *d_hist = nullptr;
        std::vector<unsigned long long> hist(HL);
        e = hipMalloc((void **)&d_hist, sizeof(unsigned long long) * HL);
        if (e == hipSuccess) e = hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * HL, g->stream);
        if (e == hipSuccess) {
            INDEX_KERNELS_BEGIN(g);
            hipLaunchKernelGGL(pw::clist_length_hist_kernel, dim3((unsigned)(((uint64_t)n_lines + 255) / 256)), dim3(256), 0, g->stream,
                               g->d_lines, n_lines, d_hist, HL);
            e = hipGetLastError();
            INDEX_KERNELS_END(g);
        }
        if (e == hipSuccess) e = hipMemcpy(hist.data(), d_hist, sizeof(unsigned long long) * HL, hipMemcpyDeviceToHost);
        if (d_hist) (void)hipFree(d_hist);
        if (e != hipSuccess) return drop(fail(PW_ERR_HIP, std::string("lane index (length histogram): ") + hipGetErrorString(e)));
        const uint64_t cap_units = std::min<uint64_t>(budget > 64 ? (budget - 64) / 16 : 0, 0xfffffffeull);
        uint64_t run = 0;
        max_len = pw::EL_INLINE;                        // (lists inside their lines are always there)
        for (uint32_t n = 0; n + 1 < HL; n++) {         // (the last bin pools every longer list: never kept when over budget)
            if (run + hist[n] > cap_units) break;
            run += hist[n];
            if (n > max_len) max_len = n;
        }
        INDEX_KERNELS_BEGIN(g);
        e = offsets(max_len);
        INDEX_KERNELS_END(g);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (e != hipSuccess) return drop(fail(PW_ERR_HIP, std::string("lane index (partial offsets): ") + hipGetErrorString(e)));
        stamp("partial index: histogram + offsets");
    }
    const uint64_t list_bytes = units * 16 + 64;
    if (units >= 0xffffffffull || list_bytes > free_b - free_b / 8) return drop(0);
    if (has_loop && max_len != 0xffffffffu) return drop(0);   // (loop_fix_kernel edits stored lists: no partial index with self loops)
    e = hipMalloc((void **)&g->d_clist, list_bytes);
    if (e != hipSuccess) return drop(e == hipErrorOutOfMemory ? 0 : fail(PW_ERR_HIP, std::string("lane index (lists): ") + hipGetErrorString(e)));
    stamp("hipMalloc of the lists");
    ba.clist = g->d_clist;
    ba.max_len = max_len;
    INDEX_KERNELS_BEGIN(g);
    if (logged) {   // the logged matches to their places; the FILL pass is left with the pairs of rows beyond 65536 entries
        hipLaunchKernelGGL(pw::lane_scatter_kernel, dim3((unsigned)(((uint64_t)nnz + 255) / 256)), dim3(256), 0, g->stream, ba, d_edge_row, d_vm0, nnz);
        ba.logged = 1u;
        if (vlines) hipLaunchKernelGGL(pw::vline_lists_kernel<true>, dim3(vgrid), dim3(256), 0, g->stream, c, g->d_lines, g->d_clist);
    } else
    lists(true);
    unsigned long long loop_removed = 0;
    if (has_loop) {   // self loops: prev's own position leaves the lists of the entries whose source has one (walk_lanes.hip.h)
        uint32_t *d_self = nullptr;
        unsigned long long *d_removed = nullptr;
        const size_t words = ((size_t)n_nodes + 31) / 32 + 1;
        e = hipMalloc((void **)&d_self, sizeof(uint32_t) * words);
        if (e == hipSuccess) e = hipMalloc((void **)&d_removed, sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemsetAsync(d_self, 0, sizeof(uint32_t) * words, g->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_removed, 0, sizeof(unsigned long long), g->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pw::self_loop_bits_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, g->stream, g->d_indptr, g->d_indices, n_nodes, d_self);
            hipLaunchKernelGGL(pw::loop_fix_kernel, dim3((unsigned)(((uint64_t)nnz + 255) / 256)), dim3(256), 0, g->stream, d_edge_row,
                               (const uint32_t *)d_self, g->d_lines, g->d_clist, nnz, d_removed);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&loop_removed, d_removed, sizeof(loop_removed), hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        if (d_self) (void)hipFree(d_self);
        if (d_removed) (void)hipFree(d_removed);
        if (e != hipSuccess) return drop(fail(PW_ERR_HIP, std::string("lane index (self loops): ") + hipGetErrorString(e)));
    }
    {   // (logged build: lane_scatter_kernel wrote the pivots of the CSR entries' lists with the lists; the overflow lines are left)
        const uint32_t first = logged ? nnz : 0u;
        if (n_lines > first)
            hipLaunchKernelGGL(pw::eline_pivots_kernel, dim3((unsigned)(((uint64_t)(n_lines - first) + 255) / 256)), dim3(256), 0, g->stream,
                               g->d_lines, g->d_clist, n_lines, first);
    }
    e = hipGetLastError();
    INDEX_KERNELS_END(g);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) return drop(fail(PW_ERR_HIP, std::string("lane index (fill pass): ") + hipGetErrorString(e)));
    stamp("fill pass + pivots");
    cleanup();
    stamp("hipFree of the scratch");
    g->n_clist = entries - loop_removed;
    g->list_max_len = max_len;
    g->vlines = vlines;
    g->clist_bytes = list_bytes;
    g->line_bytes = line_bytes;
    g->index_bytes += line_bytes + list_bytes;
    return 0;
}

PW_EXPORT int pw_csr_create(const uint32_t *indptr, const uint32_t *indices, const float *data,
                            uint32_t n_nodes, uint32_t nnz, int device, pw_graph **out) {
    if (!indptr || !out || (nnz && !indices)) return fail(PW_ERR_INVALID, "null pointer");
    if (indptr[0] != 0) return fail(PW_ERR_INVALID, "indptr[0] != 0");
    if (indptr[n_nodes] != nnz) return fail(PW_ERR_INVALID, "indptr[n_nodes] != nnz");
    const bool dbg = getenv("PECANPY_AMD_CREATE_DEBUG") != nullptr;   // wall-clock stamps of the stages on stderr
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_last = t_begin;
    auto stamp = [&](const char *what) {
        if (!dbg) return;
        const double t = now();
        fprintf(stderr, "[create] %-34s %8.2f ms (at %8.2f)\n", what, (t - t_last) * 1e3, (t - t_begin) * 1e3);
        t_last = t;
    };
    // the lane index's work items need the host indptr only: a helper thread makes them while the runtime starts and the CSR
    // is copied (validated below before anyone relies on them: a non-monotone indptr returns before the items are used)
    LaneWorkItems items;
    bool monotone = true;
    for (uint32_t i = 0; i < n_nodes && monotone; i++) monotone = indptr[i + 1] >= indptr[i];
    if (!monotone) return fail(PW_ERR_INVALID, "indptr not monotone");
    std::thread item_thread;
    try {
        item_thread = std::thread([&]() { if (nnz && !getenv("PECANPY_AMD_NO_LAZY")) make_lane_work_items(indptr, indices, n_nodes, items); });
    } catch (const std::system_error &) {   // (thread limit of the process: the items are made on this thread)
        if (nnz && !getenv("PECANPY_AMD_NO_LAZY")) make_lane_work_items(indptr, indices, n_nodes, items);
    }
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{item_thread};
    pw_graph *g = new pw_graph();
    int rc = graph_common_init(g, device);
    if (rc) { pw_graph_destroy(g); return rc; }
    stamp("runtime / streams / events");
    // Two helper threads, from here on beside the host pass below: (1) the CSR arrays go to the device (261 MB of column indices
    // at RMAT-22 from pageable memory: ~25 ms); (2) the edge lines of the lane index (64 bytes per CSR entry + one per vertex:
    // 4.4 GB at RMAT-22) are allocated -- on a box whose device memory has not been handed out before, that hipMalloc alone
    // costs ~120 ms (EXPERIMENTS.md, rounds 5-6); it runs beside the validation / membership kernels too.
    struct Pre {
        std::thread t_up, t_lines;
        void *indptr = nullptr, *indices = nullptr, *data = nullptr;   // (1): handed to g once t_up is joined
        hipError_t err = hipSuccess;
        void *lines = nullptr; uint64_t line_bytes = 0; bool lines_taken = false, csr_taken = false;   // (2)
        ~Pre() {
            if (t_up.joinable()) t_up.join();
            if (t_lines.joinable()) t_lines.join();
            if (lines && !lines_taken) (void)hipFree(lines);
            if (!csr_taken) for (void *q : {indptr, indices, data}) if (q) (void)hipFree(q);
        }
    } pre;
    auto up_work = [&pre, device, indptr, indices, data, n_nodes, nnz]() {
        if ((pre.err = hipSetDevice(device)) != hipSuccess) return;
        auto up1 = [&](void **dst, const void *src, size_t bytes) {
            if (pre.err != hipSuccess) return;
            pre.err = hipMalloc(dst, bytes ? bytes : 4);
            if (pre.err == hipSuccess && bytes) pre.err = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        };
        up1(&pre.indptr, indptr, sizeof(uint32_t) * ((size_t)n_nodes + 1));
        up1(&pre.indices, indices, sizeof(uint32_t) * (size_t)nnz);
        if (data) up1(&pre.data, data, sizeof(float) * (size_t)nnz);
    };
    try { pre.t_up = std::thread(up_work); } catch (const std::system_error &) { up_work(); }   // (thread limit: on this thread)
    if (nnz && !getenv("PECANPY_AMD_NO_LAZY") && !getenv("PECANPY_AMD_NO_PREALLOC")) {
        const bool vl = (uint64_t)nnz + n_nodes < 0xffffffffull && !getenv("PECANPY_AMD_NO_VLINES");
        pre.line_bytes = (uint64_t)(vl ? nnz + n_nodes : nnz) * sizeof(pw::ELine) + 64;
        try {
            pre.t_lines = std::thread([&pre, device]() {
                size_t free_b = 0, total_b = 0;
                if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
                if (pre.line_bytes > free_b / 2) return;               // (build_lane_index's rule: no lane index then)
                if (hipMalloc(&pre.lines, pre.line_bytes) != hipSuccess) { pre.lines = nullptr; (void)hipGetLastError(); }
            });
        } catch (const std::system_error &) {}   // (thread limit: build_lane_index allocates)
    }
    g->kind = 0;
    g->n_nodes = n_nodes;
    g->nnz = nnz;
    // O(n_nodes) host pass: monotone offsets, maximum degree, sizes of the per-row filters and index tables
    std::vector<uint32_t> foff((size_t)n_nodes + 1);
    std::vector<uint64_t> off((size_t)n_nodes + 1);
    uint32_t md = 0;
    uint64_t frun = 0, trun = 0;
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (indptr[i + 1] < indptr[i]) { pw_graph_destroy(g); return fail(PW_ERR_INVALID, "indptr not monotone"); }
        const uint32_t d = indptr[i + 1] - indptr[i];
        if (d > md) md = d;
        foff[i] = (uint32_t)frun;
        frun += pw::filter_words_for_degree(d);
        off[i] = trun;
        if (d) {
            uint64_t sz = 2;
            while (sz < 2ull * d) sz <<= 1;
            trun += sz;
        }
    }
    if (frun >= 0xffffffffull) { pw_graph_destroy(g); return fail(PW_ERR_INVALID, "graph too large for 32-bit filter offsets"); }
    if ((trun >> 1) > 0xffffffffull) { pw_graph_destroy(g); return fail(PW_ERR_INVALID, "graph too large for 32-bit index offsets"); }
    foff[n_nodes] = (uint32_t)frun;
    off[n_nodes] = trun;
    g->max_degree = md;
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        HIP_TRY(hipMalloc(dst, bytes ? bytes : 4));
        if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    // the CSR arrays, uploaded by the helper thread meanwhile (the lines' allocation may still be running: joined further down)
    if (pre.t_up.joinable()) pre.t_up.join();
    if (pre.err != hipSuccess) { pw_graph_destroy(g); return fail(pre.err == hipErrorOutOfMemory ? PW_ERR_NOMEM : PW_ERR_HIP, std::string("CSR upload: ") + hipGetErrorString(pre.err)); }
    g->d_indptr = (uint32_t *)pre.indptr; g->d_indices = (uint32_t *)pre.indices; g->d_data = pre.data;
    pre.csr_taken = true;
    stamp("host pass + H2D of the CSR");

    // ---- device side: validation, weight scan, membership index ------------------------------------------------
    uint32_t *d_edge_row = nullptr;
    unsigned long long *d_flags = nullptr;   // [0] first entry with index >= n_nodes  [1] first entry out of order
                                             // [2] some weight != 1.0f  [3] self loop present  [4] some weight negative / NaN / inf
    auto bail = [&](int code, const std::string &msg) {
        if (d_edge_row) (void)hipFree(d_edge_row);
        if (d_flags) (void)hipFree(d_flags);
        pw_graph_destroy(g);
        return fail(code, msg);
    };
    hipError_t e = hipMalloc((void **)&d_edge_row, sizeof(uint32_t) * (size_t)(nnz ? nnz : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_flags, 5 * sizeof(unsigned long long));
    unsigned long long h_flags[5] = {~0ull, ~0ull, 0ull, 0ull, 0ull};
    if (e == hipSuccess) e = hipMemcpyAsync(d_flags, h_flags, sizeof(h_flags), hipMemcpyHostToDevice, g->stream);
    g->index_build_ms = 0;
    INDEX_KERNELS_BEGIN(g);
    if (e == hipSuccess && nnz) {
        hipLaunchKernelGGL(pw::csr_edge_rows_kernel, dim3(g->n_cu * 8), dim3(256), 0, g->stream, g->d_indptr, n_nodes, d_edge_row);
        hipLaunchKernelGGL(pw::csr_validate_kernel, dim3((unsigned)(((uint64_t)nnz + 255) / 256)), dim3(256), 0, g->stream,
                           g->d_indptr, g->d_indices, (const float *)g->d_data, d_edge_row, n_nodes, nnz, d_flags);
        e = hipGetLastError();
    }
    INDEX_KERNELS_END(g);
    if (e == hipSuccess) e = hipMemcpyAsync(h_flags, d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) return bail(PW_ERR_HIP, std::string("CSR validation: ") + hipGetErrorString(e));
    if (h_flags[0] != ~0ull) {
        uint32_t row = 0;
        (void)hipMemcpy(&row, d_edge_row + h_flags[0], sizeof(row), hipMemcpyDeviceToHost);
        return bail(PW_ERR_INVALID, "CSR entry " + std::to_string(h_flags[0]) + " (row " + std::to_string(row) +
                                        "): column index >= n_nodes");
    }
    if (h_flags[1] != ~0ull) {
        uint32_t row = 0;
        (void)hipMemcpy(&row, d_edge_row + h_flags[1], sizeof(row), hipMemcpyDeviceToHost);
        return bail(PW_ERR_INVALID, "row " + std::to_string(row) + " (CSR entry " + std::to_string(h_flags[1]) +
                                        "): column indices must be strictly ascending within a row (sorted, no duplicates), "
                                        "as the reference's to_csr produces them (graph.py:336)");
    }
    stamp("validation kernels");
    g->unit = !(data && h_flags[2]);
    if (g->unit && g->d_data) { (void)hipFree(g->d_data); g->d_data = nullptr; }   // unit weights are never read
    const bool has_loop = h_flags[3] != 0;
    // A negative, NaN or infinite weight: the reference turns it into meaningless "probabilities" (w / w.sum(), cumsum, searchsorted:
    // pecanpy.py:556-557) without complaint; the exact scans here (wave kernel: monotone partial sums; weighted lane form: a float64
    // bound that takes the sign of a common neighbour's delta from q alone, seqscan.h: WeightedRow::margin) would NOT reproduce that
    // garbage bit for bit, so such a graph is refused instead of walked differently (error behaviour)
    if (data && h_flags[4] != 0)
        return bail(PW_ERR_INVALID, "edge weights must be finite and >= 0 (a negative, NaN or infinite weight makes the reference's "
                                    "transition probabilities w / w.sum() meaningless; this library does not reproduce them)");

    rc = up((void **)&g->d_foff, foff.data(), sizeof(uint32_t) * foff.size());
    if (!rc) rc = up((void **)&g->d_tab_off, off.data(), sizeof(uint64_t) * off.size());
    if (rc) { if (d_edge_row) (void)hipFree(d_edge_row); (void)hipFree(d_flags); pw_graph_destroy(g); return rc; }
    e = hipMalloc((void **)&g->d_fbits, sizeof(uint64_t) * (frun ? frun : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_kf, sizeof(uint2) * (size_t)(nnz ? nnz : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_slots, sizeof(uint64_t) * (trun ? trun : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_vrec, sizeof(uint4) * ((size_t)n_nodes + 1));
    if (e == hipSuccess) e = hipMemsetAsync(g->d_fbits, 0, sizeof(uint64_t) * (frun ? frun : 1), g->stream);
    if (e == hipSuccess) e = hipMemsetAsync(g->d_slots, 0xff, sizeof(uint64_t) * (trun ? trun : 1), g->stream);
    if (e != hipSuccess) return bail(PW_ERR_NOMEM, std::string("membership index: ") + hipGetErrorString(e));
    g->index_bytes = sizeof(uint64_t) * (frun + trun) + sizeof(uint2) * (uint64_t)nnz + sizeof(uint4) * ((uint64_t)n_nodes + 1);
    g->fbits_words = frun;
    g->slot_words = trun;
    INDEX_KERNELS_BEGIN(g);
    hipLaunchKernelGGL(pw::vrec_build_kernel, dim3((n_nodes + 256) / 256), dim3(256), 0, g->stream, g->d_indptr, g->d_foff,
                       g->d_tab_off, n_nodes, g->d_vrec);
    if (n_nodes && nnz)
        hipLaunchKernelGGL(pw::membership_build_kernel, dim3((unsigned)(((uint64_t)nnz + 255) / 256)), dim3(256), 0, g->stream,
                           g->d_indptr, g->d_indices, d_edge_row, g->d_foff, g->d_tab_off, (unsigned long long *)g->d_fbits, g->d_kf,
                           (unsigned long long *)g->d_slots, nnz, 4294967296.0 / (double)nnz);
    e = hipGetLastError();
    INDEX_KERNELS_END(g);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) return bail(PW_ERR_HIP, std::string("membership index build: ") + hipGetErrorString(e));
    stamp("membership index");
    g->has_loop = has_loop;
    if (nnz && (!has_loop || g->unit) && !getenv("PECANPY_AMD_NO_LAZY")) {   // (weighted graphs too: membership does not depend on the weights)
        // per-edge records and common-neighbour lists (lane kernel; lazy membership of the wave kernel).  SELF LOOPS (round 6):
        // unit-weight graphs keep the index -- prev's own position is taken out of the lists of the entries whose source has a
        // loop (loop_fix_kernel) -- and only the wave kernel's LAZY step, which counts common neighbours while it classifies
        // keys, stays off for them (launch_wave_walks); weighted graphs with self loops get no index, as before (the
        // node2vec+ tables pair the two directions' lists entry by entry, and the fix makes their lengths differ).
        g->lanes_off = getenv("PECANPY_AMD_NO_LANES") != nullptr;
        item_thread.join();
        if (pre.t_lines.joinable()) pre.t_lines.join();
        pre.lines_taken = pre.lines != nullptr;
        rc = build_lane_index(g, items, d_edge_row, has_loop, pre.lines, pre.line_bytes);
        if (rc) { (void)hipFree(d_edge_row); (void)hipFree(d_flags); pw_graph_destroy(g); return rc; }
    }
    (void)hipStreamSynchronize(g->stream);
    stamp("lane index");
    g->create_wall_ms = (now() - t_begin) * 1e3;
    // a PARTIAL index keeps the source vertex of every CSR entry (4 bytes per entry): lanes_eager_kernel reads the vertex a
    // parked step came from there instead of bisecting indptr for it (22 dependent loads per step)
    if (g->d_lines && g->list_max_len != 0xffffffffu && !g->d_wedge_row) g->d_wedge_row = d_edge_row;
    else (void)hipFree(d_edge_row);
    (void)hipFree(d_flags);
    *out = g;
    return PW_OK;
}

PW_EXPORT int pw_graph_index_info(const pw_graph *g, double *build_ms, uint64_t *index_bytes, uint64_t *lane_list_entries) {
    if (!g) return fail(PW_ERR_INVALID, "null pointer");
    if (build_ms) *build_ms = g->index_build_ms;
    if (index_bytes) *index_bytes = g->index_bytes;
    if (lane_list_entries) *lane_list_entries = g->d_lines ? g->n_clist : 0;
    return PW_OK;
}

// Test hook: the lane index decoded to flat arrays -- per CSR entry e = (u -> v) the number of common neighbours of u
// and v, the position of u in row v (0xffffffff: no reverse entry), and -- concatenated in entry order -- the positions
// in row v of the common neighbours.
PW_EXPORT int pw_lane_index_export(pw_graph *g, uint32_t *n_in, uint32_t *rev_pos, uint32_t *entries) {
    if (!g) return fail(PW_ERR_INVALID, "null pointer");
    if (!g->d_lines) return fail(PW_ERR_UNSUPPORTED, "this graph has no lane index");
    if (set_device(g)) return PW_ERR_HIP;
    const uint32_t nnz = g->nnz;
    std::vector<pw::ELine> lines(nnz);
    HIP_TRY(hipMemcpy(lines.data(), g->d_lines, sizeof(pw::ELine) * (size_t)nnz, hipMemcpyDeviceToHost));
    std::vector<uint64_t> off((size_t)nnz + 1, 0);
    for (uint32_t e = 0; e < nnz; e++) {
        if (n_in) n_in[e] = lines[e].n_in;
        if (rev_pos) rev_pos[e] = lines[e].rev_pos;
        off[e + 1] = off[e] + lines[e].n_in;
    }
    if (off[nnz] != g->n_clist) return fail(PW_ERR_HIP, "lane index: entry count mismatch");
    if (!entries || !off[nnz]) return PW_OK;
    uint64_t *d_off = nullptr;
    uint32_t *d_out = nullptr;
    hipError_t e = hipMalloc((void **)&d_off, sizeof(uint64_t) * off.size());
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof(uint32_t) * (size_t)off[nnz]);
    if (e == hipSuccess) e = hipMemcpy(d_off, off.data(), sizeof(uint64_t) * off.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pw::lane_index_export_kernel, dim3((unsigned)(((uint64_t)nnz + 255) / 256)), dim3(256), 0, g->stream,
                           g->d_lines, g->d_clist, nnz, d_off, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e == hipSuccess) e = hipMemcpy(entries, d_out, sizeof(uint32_t) * (size_t)off[nnz], hipMemcpyDeviceToHost);
    if (d_off) (void)hipFree(d_off);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(PW_ERR_HIP, std::string("pw_lane_index_export: ") + hipGetErrorString(e));
    return PW_OK;
}

PW_EXPORT int pw_dense_create(const double *data, uint32_t n_nodes, int device, pw_graph **out) {
    if (!data || !out) return fail(PW_ERR_INVALID, "null pointer");
    // HBM layout of a dense graph (declared format): bit-packed adjacency rows for O(1)
    // membership + the non-zero entries of every row compressed in ascending column order
    // (uint32 column, float64 value -- values dropped when they are all 1.0).
    const uint64_t n = n_nodes;
    const uint32_t wpr = (uint32_t)((n + 63) / 64);
    std::vector<uint32_t> indptr(n + 1, 0);
    std::vector<uint32_t> cols;
    std::vector<double> vals;
    std::vector<uint64_t> bits((size_t)n * wpr, 0);
    bool unit = true, nonneg = true;
    uint64_t nnz = 0;
    for (uint64_t i = 0; i < n; i++) {
        const double *row = data + i * n;
        for (uint64_t x = 0; x < n; x++) {
            const double v = row[x];
            if (v != 0.0) {
                if (nnz >= 0xffffffffull) return fail(PW_ERR_INVALID, "dense graph has more than 2^32-1 edges");
                cols.push_back((uint32_t)x);
                vals.push_back(v);
                bits[i * wpr + (x >> 6)] |= 1ull << (x & 63);
                if (v != 1.0) unit = false;
                if (!(v > 0.0) || !(v < 0x1p1000)) nonneg = false;
                nnz++;
            }
        }
        indptr[i + 1] = (uint32_t)nnz;
    }
    pw_graph *g = new pw_graph();
    int rc = graph_common_init(g, device);
    if (rc) { pw_graph_destroy(g); return rc; }
    g->kind = 1;
    g->n_nodes = n_nodes;
    g->nnz = (uint32_t)nnz;
    g->unit = unit;
    g->dense_nonneg = nonneg;
    g->words_per_row = wpr;
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        HIP_TRY(hipMalloc(dst, bytes ? bytes : 8));
        if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    rc = up((void **)&g->d_indptr, indptr.data(), sizeof(uint32_t) * indptr.size());
    if (!rc) rc = up((void **)&g->d_indices, cols.data(), sizeof(uint32_t) * cols.size());
    if (!rc && !unit) rc = up((void **)&g->d_data, vals.data(), sizeof(double) * vals.size());
    if (!rc) rc = up((void **)&g->d_adjbits, bits.data(), sizeof(uint64_t) * bits.size());
    if (!rc) {
        std::vector<uint32_t> deg(n);
        for (uint64_t i = 0; i < n; i++) {
            deg[i] = indptr[i + 1] - indptr[i];
            if (deg[i] > g->max_degree) g->max_degree = deg[i];
        }
        rc = up((void **)&g->d_deg, deg.data(), sizeof(uint32_t) * deg.size());
    }
    if (rc) { pw_graph_destroy(g); return rc; }
    *out = g;
    return PW_OK;
}

PW_EXPORT int pw_dense_create_bits(const uint64_t *adjbits, uint32_t n_nodes, int on_device, int device,
                                   pw_graph **out) {
    if (!adjbits || !out || n_nodes == 0) return fail(PW_ERR_INVALID, "null pointer / empty graph");
    pw_graph *g = new pw_graph();
    int rc = graph_common_init(g, device);
    if (rc) { pw_graph_destroy(g); return rc; }
    const uint64_t n = n_nodes;
    const uint32_t wpr = (uint32_t)((n + 63) / 64);
    g->kind = 1;
    g->n_nodes = n_nodes;
    g->unit = true;
    g->bits_only = true;
    g->words_per_row = wpr;
    const size_t bytes = sizeof(uint64_t) * (size_t)n * wpr;
    hipError_t e = hipMalloc((void **)&g->d_adjbits, bytes);
    // the copy must be ordered with the kernels of this handle's (non-blocking) stream
    if (e == hipSuccess) e = hipMemcpyAsync(g->d_adjbits, adjbits, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_deg, sizeof(uint32_t) * n);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pw::dense_degree_kernel, dim3(g->n_cu * 8), dim3(256), 0, g->stream, g->d_adjbits, n_nodes, wpr, g->d_deg);
        e = hipGetLastError();
    }
    std::vector<uint32_t> deg(n), indptr(n + 1, 0);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e == hipSuccess) e = hipMemcpy(deg.data(), g->d_deg, sizeof(uint32_t) * n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { pw_graph_destroy(g); return fail(PW_ERR_HIP, std::string("pw_dense_create_bits: ") + hipGetErrorString(e)); }
    uint64_t nnz = 0;
    for (uint64_t i = 0; i < n; i++) {
        nnz += deg[i];
        if (nnz >= 0xffffffffull) { pw_graph_destroy(g); return fail(PW_ERR_INVALID, "dense graph has more than 2^32-1 edges"); }
        indptr[i + 1] = (uint32_t)nnz;
        if (deg[i] > g->max_degree) g->max_degree = deg[i];
    }
    g->nnz = (uint32_t)nnz;
    e = hipMalloc((void **)&g->d_indptr, sizeof(uint32_t) * (n + 1));   // only used for stream offsets
    if (e == hipSuccess) e = hipMemcpy(g->d_indptr, indptr.data(), sizeof(uint32_t) * (n + 1), hipMemcpyHostToDevice);
    if (e != hipSuccess) { pw_graph_destroy(g); return fail(PW_ERR_HIP, std::string("pw_dense_create_bits: ") + hipGetErrorString(e)); }
    *out = g;
    return PW_OK;
}

PW_EXPORT int pw_graph_set_thresholds(pw_graph *g, const float *thr) {
    if (!g || !thr) return fail(PW_ERR_INVALID, "null pointer");
    if (set_device(g)) return PW_ERR_HIP;
    if (!g->d_thr) HIP_TRY(hipMalloc((void **)&g->d_thr, sizeof(float) * (size_t)g->n_nodes));
    HIP_TRY(hipMemcpy(g->d_thr, thr, sizeof(float) * (size_t)g->n_nodes, hipMemcpyHostToDevice));
    g->thr_version++;
    return PW_OK;
}

// ---- in-process multi-device (round 6; SURVEY.md section 8(b)/(e): the reference is ONE process, cli.py:340-351) -----------
// A replica of a graph handle on another device (or the same one again): the CSR and the whole per-graph index are COPIED
// device to device (hipMemcpyPeer: xGMI between GPUs) instead of being built again -- RMAT-22: 12 GB at link speed against
// 170 ms of index kernels + the host passes per device.  Per-(p, q) tables are built by each replica's first call.
PW_EXPORT int pw_graph_replicate(const pw_graph *src, int device, pw_graph **out) {
    if (!src || !out) return fail(PW_ERR_INVALID, "null pointer");
    pw_graph *g = new pw_graph();
    int rc = graph_common_init(g, device);
    if (rc) { pw_graph_destroy(g); return rc; }
    g->kind = src->kind; g->n_nodes = src->n_nodes; g->nnz = src->nnz; g->unit = src->unit; g->max_degree = src->max_degree;
    g->bits_only = src->bits_only; g->dense_nonneg = src->dense_nonneg; g->words_per_row = src->words_per_row; g->has_loop = src->has_loop; g->lanes_off = src->lanes_off;
    g->vlines = src->vlines; g->clist_bytes = src->clist_bytes; g->line_bytes = src->line_bytes; g->fbits_words = src->fbits_words;
    g->slot_words = src->slot_words; g->n_clist = src->n_clist; g->list_max_len = src->list_max_len; g->index_bytes = src->index_bytes;
    g->thr_version = src->d_thr ? 1 : 0;
    const uint64_t n = src->n_nodes, nnz = src->nnz;
    const bool rows = !src->bits_only;   // (dense graphs created from packed bits have no compressed rows)
    struct Buf { void **dst; const void *from; uint64_t bytes; };
    const Buf bufs[] = {
        {(void **)&g->d_indptr, src->d_indptr, sizeof(uint32_t) * (n + 1)},
        {(void **)&g->d_indices, src->d_indices, rows ? sizeof(uint32_t) * nnz : 0},
        {(void **)&g->d_data, src->d_data, (src->kind == 0 ? sizeof(float) : sizeof(double)) * nnz},
        {(void **)&g->d_thr, src->d_thr, sizeof(float) * n},
        {(void **)&g->d_adjbits, src->d_adjbits, sizeof(uint64_t) * n * src->words_per_row},
        {(void **)&g->d_deg, src->d_deg, sizeof(uint32_t) * n},
        {(void **)&g->d_foff, src->d_foff, sizeof(uint32_t) * (n + 1)},
        {(void **)&g->d_fbits, src->d_fbits, sizeof(uint64_t) * src->fbits_words},
        {(void **)&g->d_kf, src->d_kf, sizeof(uint2) * nnz},
        {(void **)&g->d_tab_off, src->d_tab_off, sizeof(uint64_t) * (n + 1)},
        {(void **)&g->d_slots, src->d_slots, sizeof(uint64_t) * src->slot_words},
        {(void **)&g->d_vrec, src->d_vrec, sizeof(uint4) * (n + 1)},
        {(void **)&g->d_lines, src->d_lines, src->line_bytes},
        {(void **)&g->d_clist, src->d_clist, src->clist_bytes},
    };
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipSuccess;
    for (const Buf &b : bufs) {
        if (!b.from) continue;
        e = hipMalloc(b.dst, b.bytes ? b.bytes : 8);
        if (e == hipSuccess && b.bytes) e = hipMemcpyPeerAsync(*b.dst, device, b.from, src->device, b.bytes, g->stream);
        if (e != hipSuccess) break;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) {
        pw_graph_destroy(g);
        return fail(e == hipErrorOutOfMemory ? PW_ERR_NOMEM : PW_ERR_HIP, std::string("pw_graph_replicate: ") + hipGetErrorString(e));
    }
    g->create_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g->index_build_ms = 0;     // (nothing was built here)
    *out = g;
    return PW_OK;
}

PW_EXPORT int pw_device_mask_to_list(uint64_t device_mask, int *devices, int cap) {
    if (!devices && cap > 0) return fail(PW_ERR_INVALID, "null pointer");
    const int nd = pw_device_count();
    int k = 0;
    for (int d = 0; d < 64; d++) {
        if (!((device_mask >> d) & 1ull)) continue;
        if (d >= nd) return fail(PW_ERR_INVALID, "device_mask names device " + std::to_string(d) + ", " + std::to_string(nd) + " visible");
        if (k < cap) devices[k] = d;
        k++;
    }
    return k;
}

PW_EXPORT int pw_csr_create_multi(const uint32_t *indptr, const uint32_t *indices, const float *data, uint32_t n_nodes, uint32_t nnz,
                                  const int *devices, int n_devices, pw_graph **out_handles) {
    if (!devices || !out_handles || n_devices < 1) return fail(PW_ERR_INVALID, "null pointer / no device named");
    for (int i = 0; i < n_devices; i++) out_handles[i] = nullptr;
    int rc = pw_csr_create(indptr, indices, data, n_nodes, nnz, devices[0], &out_handles[0]);
    // the replicas: one helper thread per device, so that the copies run side by side (one xGMI link per destination)
    std::vector<int> rcs((size_t)n_devices, 0);
    std::vector<std::string> errs((size_t)n_devices);
    std::vector<std::thread> th;
    for (int i = 1; i < n_devices && !rc; i++)
        th.emplace_back([&, i]() {
            rcs[i] = pw_graph_replicate(out_handles[0], devices[i], &out_handles[i]);
            if (rcs[i]) errs[i] = g_err;
        });
    for (auto &t : th) t.join();
    for (int i = 1; i < n_devices && !rc; i++)
        if (rcs[i]) rc = fail(rcs[i], errs[i]);
    if (rc)
        for (int i = 0; i < n_devices; i++) { if (out_handles[i]) pw_graph_destroy(out_handles[i]); out_handles[i] = nullptr; }
    return rc;
}

// exclusive prefix of per-job draw counts -> g->stream_off; returns total through *total.
// Block-wise repair: a WINDOW of the job array -- jobs [j0, j0 + n_jobs), `skip` = the stream offset of job j0 (d_starts / d_walks
// are the WHOLE arrays; the buffers are sized for n_all jobs) -- so that a round costs O(window), not O(job array).
// *first_mismatch (optional) = smallest index of a job whose offset differs from the one it was last walked with (~0: none).
static int compute_offsets(pw_graph *g, const uint32_t *d_starts, const uint32_t *d_walks, uint32_t L,
                           uint64_t n_jobs, uint64_t skip, bool track_changes, uint64_t *total,
                           uint64_t *n_changed, uint64_t j0 = 0, uint64_t n_all = 0, uint64_t *first_mismatch = nullptr) {
    if (n_all < j0 + n_jobs) n_all = j0 + n_jobs;
    uint64_t n_tiles = (n_jobs + pw::SCAN_TILE - 1) / pw::SCAN_TILE;
    if (g->stream_off.ensure(n_all + 1)) return PW_ERR_NOMEM;
    if (g->tile_sums.ensure((n_all + pw::SCAN_TILE - 1) / pw::SCAN_TILE + 1)) return PW_ERR_NOMEM;
    if (track_changes && g->changed.ensure(n_all)) return PW_ERR_NOMEM;
    if (!g->d_hasnbr) {
        const uint32_t words = (g->n_nodes + 31u) / 32u;
        HIP_TRY(hipMalloc((void **)&g->d_hasnbr, sizeof(uint32_t) * (size_t)(words ? words : 1)));
        hipLaunchKernelGGL(pw::has_nbr_bits_kernel, dim3((words + 255) / 256 ? (words + 255) / 256 : 1), dim3(256), 0, g->stream, g->d_indptr, g->n_nodes, g->d_hasnbr);
    }
    const uint32_t *w_starts = d_starts + j0;
    const uint32_t *w_walks = d_walks ? d_walks + j0 * ((uint64_t)L + 2) : nullptr;
    unsigned long long *cc = g->counters.p + 5;
    if (track_changes) HIP_TRY(hipMemsetAsync(cc, 0, sizeof(unsigned long long), g->stream));
    unsigned long long *fm = first_mismatch ? g->counters.p + 14 : nullptr;
    if (fm) HIP_TRY(hipMemsetAsync(fm, 0xff, sizeof(unsigned long long), g->stream));
    hipLaunchKernelGGL(pw::draws_tile_sums_kernel, dim3((unsigned)n_tiles), dim3(pw::SCAN_BLOCK), 0, g->stream,
                       g->d_hasnbr, w_starts, w_walks, L, n_jobs, g->tile_sums.p);
    hipLaunchKernelGGL(pw::scan_tile_sums_kernel, dim3(1), dim3(pw::SCAN_BLOCK), 0, g->stream, g->tile_sums.p, n_tiles);
    hipLaunchKernelGGL(pw::draws_offsets_kernel, dim3((unsigned)n_tiles), dim3(pw::SCAN_BLOCK), 0, g->stream,
                       g->d_hasnbr, w_starts, w_walks, L, n_jobs, g->tile_sums.p, skip, g->stream_off.p + j0,
                       track_changes ? g->changed.p : nullptr, cc, j0, fm);
    HIP_TRY(hipGetLastError());
    uint64_t tot = 0;
    HIP_TRY(hipMemcpyAsync(&tot, g->tile_sums.p + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
    unsigned long long nc = 0, fmv = ~0ull;
    if (track_changes) HIP_TRY(hipMemcpyAsync(&nc, cc, sizeof(nc), hipMemcpyDeviceToHost, g->stream));
    if (fm) HIP_TRY(hipMemcpyAsync(&fmv, fm, sizeof(fmv), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    *total = tot;
    if (n_changed) *n_changed = nc;
    if (first_mismatch) *first_mismatch = fmv;
    return 0;
}

// every start must be a vertex (the kernels index indptr / the vertex records with it)
static int check_starts(pw_graph *g, const uint32_t *d_starts, uint64_t n_jobs) {
    if (!n_jobs) return 0;
    unsigned long long bad = ~0ull;
    HIP_TRY(hipMemcpyAsync(g->counters.p + 9, &bad, sizeof(bad), hipMemcpyHostToDevice, g->stream));
    hipLaunchKernelGGL(pw::starts_check_kernel, dim3((unsigned)((n_jobs + 255) / 256)), dim3(256), 0, g->stream, d_starts,
                       n_jobs, g->n_nodes, g->counters.p + 9);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&bad, g->counters.p + 9, sizeof(bad), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (bad != ~0ull) return fail(PW_ERR_INVALID, "start vertex out of range at job " + std::to_string(bad));
    return 0;
}

PW_EXPORT int pw_count_stream_draws(pw_graph *g, const uint32_t *starts, uint64_t n_jobs,
                                    uint32_t walk_length, uint64_t *out_draws) {
    if (!g || !starts || !out_draws) return fail(PW_ERR_INVALID, "null pointer");
    if (set_device(g)) return PW_ERR_HIP;
    if (g->counters.ensure(N_COUNTERS)) return PW_ERR_NOMEM;
    uint32_t *d_starts = nullptr;
    HIP_TRY(hipMalloc((void **)&d_starts, sizeof(uint32_t) * (n_jobs ? n_jobs : 1)));
    hipError_t e = hipMemcpy(d_starts, starts, sizeof(uint32_t) * n_jobs, hipMemcpyHostToDevice);
    int rc = 0;
    uint64_t tot = 0;
    if (e != hipSuccess) rc = fail(PW_ERR_HIP, hipGetErrorString(e));
    if (!rc) rc = check_starts(g, d_starts, n_jobs);
    if (!rc && n_jobs) rc = compute_offsets(g, d_starts, nullptr, walk_length, n_jobs, 0, false, &tot, nullptr);
    (void)hipFree(d_starts);
    *out_draws = tot;
    return rc;
}

static pw::CsrDev csr_dev(const pw_graph *g) {
    pw::CsrDev c;
    c.indptr = g->d_indptr;
    c.indices = g->d_indices;
    c.data = g->d_data;
    c.thr = g->d_thr;
    c.adjbits = g->d_adjbits;
    c.foff = g->d_foff;
    c.fbits = g->d_fbits;
    c.kf = g->d_kf;
    c.tab_off = g->d_tab_off;
    c.slots = g->d_slots;
    c.tri = (const uint4 *)g->d_lines;   // (stride: one 64-byte line per CSR entry)
    c.clist = g->d_clist;
    c.step_edge = 0xffffffffu;
    c.vrec = g->d_vrec;
    c.words_per_row = g->words_per_row;
    c.n_nodes = g->n_nodes;
    c.nnz = g->nnz;
    return c;
}

PW_EXPORT int pw_precomp_build(pw_graph *g, double p, double q, int extend, int first_order) {
    if (!g) return fail(PW_ERR_INVALID, "null pointer");
    if (g->kind != 0) return fail(PW_ERR_UNSUPPORTED, "alias tables need a CSR graph handle");
    if (!(p > 0) || !(q > 0)) return fail(PW_ERR_INVALID, "p and q must be positive");
    if (extend && !g->d_thr) return fail(PW_ERR_INVALID, "extend: call pw_graph_set_thresholds() first");
    if (set_device(g)) return PW_ERR_HIP;
    first_order = first_order ? 1 : 0;
    if (g->alias_kind == first_order && (first_order || (g->alias_p == p && g->alias_q_param == q && g->alias_extend == extend)))
        return PW_OK;
    const uint32_t n = g->n_nodes;
    std::vector<uint32_t> indptr((size_t)n + 1);
    HIP_TRY(hipMemcpy(indptr.data(), g->d_indptr, sizeof(uint32_t) * indptr.size(), hipMemcpyDeviceToHost));
    std::vector<uint64_t> aptr((size_t)n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        uint64_t d = indptr[i + 1] - indptr[i];
        aptr[i + 1] = aptr[i] + (first_order ? d : d * d);
    }
    const uint64_t n_alias = aptr[n];
    const size_t cap = n_alias ? n_alias : 1;
    if (g->alias_indptr.ensure((size_t)n + 1) || g->alias_j.ensure(cap) || g->alias_q.ensure(cap) ||
        g->alias_s.ensure(cap) || g->alias_l.ensure(cap) || g->edge_row.ensure(g->nnz ? g->nnz : 1))
        return PW_ERR_NOMEM;
    HIP_TRY(hipMemcpy(g->alias_indptr.p, aptr.data(), sizeof(uint64_t) * aptr.size(), hipMemcpyHostToDevice));
    pw::CsrDev c = csr_dev(g);
    if (n) hipLaunchKernelGGL(pw::edge_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, g->stream, g->d_indptr, n, g->edge_row.p);
    const uint64_t work = first_order ? (uint64_t)n : (uint64_t)g->nnz;
    if (work)
        hipLaunchKernelGGL(pw::alias_tables_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, g->stream, c, p, q,
                           extend, first_order, g->edge_row.p, g->alias_indptr.p, g->alias_j.p, g->alias_q.p,
                           g->alias_s.p, g->alias_l.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->alias_s.release();   // scratch of the build only (2 x sum(deg^2) words)
    g->alias_l.release();
    g->n_alias = n_alias;
    g->alias_kind = first_order;
    g->alias_p = p;
    g->alias_q_param = q;
    g->alias_extend = extend;
    return PW_OK;
}

PW_EXPORT int pw_precomp_export(pw_graph *g, uint64_t *alias_indptr, uint32_t *alias_j, float *alias_q,
                                uint64_t *n_entries) {
    if (!g) return fail(PW_ERR_INVALID, "null pointer");
    if (g->alias_kind < 0) return fail(PW_ERR_INVALID, "no alias tables built");
    if (set_device(g)) return PW_ERR_HIP;
    if (n_entries) *n_entries = g->n_alias;
    if (alias_indptr)
        HIP_TRY(hipMemcpy(alias_indptr, g->alias_indptr.p, sizeof(uint64_t) * ((size_t)g->n_nodes + 1), hipMemcpyDeviceToHost));
    if (alias_j && g->n_alias) HIP_TRY(hipMemcpy(alias_j, g->alias_j.p, sizeof(uint32_t) * g->n_alias, hipMemcpyDeviceToHost));
    if (alias_q && g->n_alias) HIP_TRY(hipMemcpy(alias_q, g->alias_q.p, sizeof(float) * g->n_alias, hipMemcpyDeviceToHost));
    return PW_OK;
}

// Sequential-stream modes (variable word consumption): one lane walks every job in order.
static int simulate_sequential(pw_graph *g, int mode, double p, double q, int extend, const uint32_t *d_starts,
                               uint64_t n_jobs, uint32_t L, int has_seed, uint32_t seed, uint32_t *d_out,
                               pw_stats *st) {
    if (g->kind != 0) return fail(PW_ERR_UNSUPPORTED, "this mode needs a CSR graph handle");
    if (mode == PW_MODE_PRECOMP) { int rc = pw_precomp_build(g, p, q, extend, 0); if (rc) return rc; }
    if (mode == PW_MODE_PRECOMP_FIRST_ORDER) { int rc = pw_precomp_build(g, 1.0, 1.0, 0, 1); if (rc) return rc; }
    if (g->mt_state.ensure(pw::MT_N)) return PW_ERR_NOMEM;
    if (g->probs_scratch.ensure((size_t)g->max_degree + 1)) return PW_ERR_NOMEM;
    uint32_t st0[pw::MT_N];
    pw::mt_seed_state(st0, seed);
    HIP_TRY(hipMemcpy(g->mt_state.p, st0, sizeof(st0), hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(g->counters.p, 0, 8 * sizeof(unsigned long long), g->stream));
    pw::SeqArgs a;
    a.g = csr_dev(g);
    a.p = p;
    a.q = q;
    a.mode = mode;
    a.L = L;
    a.n_jobs = n_jobs;
    a.starts = d_starts;
    a.mt_seed_state = g->mt_state.p;
    a.alias_indptr = g->alias_indptr.p;
    a.alias_j = g->alias_j.p;
    a.alias_q = g->alias_q.p;
    a.n_alias = g->n_alias;
    a.probs_scratch = g->probs_scratch.p;
    a.out = d_out;
    a.stats = g->counters.p + 1;
    HIP_TRY(hipEventRecord(g->ev[2], g->stream));
    if (has_seed)  // reproducible: the reference's single sequential stream
        hipLaunchKernelGGL(pw::walk_seq_kernel, dim3(1), dim3(64), 0, g->stream, a);
    else           // random_state=None: nothing to reproduce, all walks in parallel
        hipLaunchKernelGGL(pw::walk_alias_parallel_kernel, dim3((unsigned)((n_jobs + 255) / 256)), dim3(256), 0,
                           g->stream, a, (uint64_t)seed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(g->ev[3], g->stream));
    unsigned long long h[8];
    HIP_TRY(hipMemcpyAsync(h, g->counters.p, sizeof(h), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev[2], g->ev[3]));
    st->walk_kernel_ms = ms;
    st->walk_kernel_launches = 1;
    st->total_steps = h[1];
    st->overflow_reads = h[2];
    st->clamped_reads = h[3];
    st->dead_end_walks = h[4];
    return PW_OK;
}

typedef void (*walk_kernel_fn)(pw::WalkArgs);

static walk_kernel_fn pick_kernel(const pw_graph *g, bool extend) {
    if (g->kind == 0) {
        if (g->unit) return pw::walk_kernel<float, false, true, false>;
        return extend ? pw::walk_kernel<float, false, false, true> : pw::walk_kernel<float, false, false, false>;
    }
    if (g->unit) return pw::walk_kernel<double, true, true, false>;
    return extend ? pw::walk_kernel<double, true, false, true> : pw::walk_kernel<double, true, false, false>;
}

// unweighted dense graphs: column-space kernel on the packed adjacency (walk_dense.hip.h)
static bool is_pow2_double(double x) {
    int e = 0;
    return x > 0 && std::frexp(x, &e) == 0.5;
}

static int launch_dense_bits(pw_graph *g, const pw::WalkArgs &wa, uint64_t *redo_total) {
    pw::DenseArgs da;
    da.adjbits = g->d_adjbits;
    da.deg = g->d_deg;
    da.n = g->n_nodes;
    da.wpr = g->words_per_row;
    da.p = wa.p;
    da.q = wa.q;
    da.L = wa.L;
    da.n_jobs = wa.n_jobs;
    da.starts = wa.starts;
    da.stream_off = wa.stream_off;
    da.job_list = wa.job_list;
    da.n_list = wa.n_list;
    da.rng = wa.rng;
    da.rng_base = wa.rng_base;
    da.out = wa.out;
    da.job_counter = wa.job_counter;
    da.stats = wa.stats;
    // rows of up to 131 072 columns: the row of `cur` is kept in registers for the next step (WPL words per lane)
    typedef void (*dense_fn)(pw::DenseArgs);
    dense_fn fn = pw::walk_dense_bits_kernel<0>;
    if (!getenv("PECANPY_AMD_DENSE_NO_KEEP")) {
        if (da.wpr <= 64 * 8) fn = pw::walk_dense_bits_kernel<8>;
        else if (da.wpr <= 64 * 16) fn = pw::walk_dense_bits_kernel<16>;
        else if (da.wpr <= 64 * 25) fn = pw::walk_dense_bits_kernel<25>;
        else if (da.wpr <= 64 * 32) fn = pw::walk_dense_bits_kernel<32>;
    }
    uint64_t n_work = wa.job_list ? wa.n_list : wa.n_jobs;
    // dyadic 1/p, 1/q: the decisive path alone, in registers (walk_dense_fast_kernel); what it leaves (a handful of
    // walks per 10^8 steps) is walked again by the complete kernel
    typedef void (*fast_fn)(pw::DenseArgs, uint32_t *, unsigned long long *, uint32_t);
    fast_fn ff = nullptr;
    const bool dyadic = is_pow2_double(1.0 / wa.p) && is_pow2_double(1.0 / wa.q);
    if (fn != pw::walk_dense_bits_kernel<0> && dyadic && !getenv("PECANPY_AMD_DENSE_NO_FAST")) {
        // (round 6: prev's row in LDS instead of registers -- three wavefronts per SIMD instead of two; PECANPY_AMD_DENSE_KEEP_REGS=1:
        //  the register form, rounds 3-5)
        const bool ldsk = getenv("PECANPY_AMD_DENSE_KEEP_REGS") == nullptr;
        if (da.wpr <= 64 * 8) ff = ldsk ? pw::walk_dense_fast_kernel<8, 0, true> : pw::walk_dense_fast_kernel<8, 0>;
        else if (da.wpr <= 64 * 16) ff = ldsk ? pw::walk_dense_fast_kernel<16, 8, true> : pw::walk_dense_fast_kernel<16, 8>;
        else if (da.wpr <= 64 * 25) ff = ldsk ? pw::walk_dense_fast_kernel<25, 16, true> : pw::walk_dense_fast_kernel<25, 16>;
        else ff = pw::walk_dense_fast_kernel<32, 25>;
    } else if (fn != pw::walk_dense_bits_kernel<0> && !dyadic && !getenv("PECANPY_AMD_DENSE_NO_FAST") && !getenv("PECANPY_AMD_DENSE_NO_BOUNDED")) {
        // 1/p or 1/q not a power of two: the same kernel with float64 masses and the float64-bounded decision (round 6; before:
        // walk_dense_bits_kernel alone: 22.7 against 467 M steps/s at ER-100k)
        if (da.wpr <= 64 * 8) ff = pw::walk_dense_fast_kernel<8, 0, true, true>;
        else if (da.wpr <= 64 * 16) ff = pw::walk_dense_fast_kernel<16, 8, true, true>;
        else if (da.wpr <= 64 * 25) ff = pw::walk_dense_fast_kernel<25, 16, true, true>;
        else ff = pw::walk_dense_fast_kernel<32, 25, false, true>;
    }
    auto grid_for = [&](const void *f, uint64_t work, unsigned *out) -> int {
        int occ = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, f, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
        if (occ < 1) occ = 1;
        const uint64_t want = (work + pw::WAVES_PER_BLOCK - 1) / pw::WAVES_PER_BLOCK;
        uint64_t grid = (uint64_t)g->n_cu * (uint64_t)occ;
        if (grid > want) grid = want;
        if (grid < 1) grid = 1;
        *out = (unsigned)grid;
        return 0;
    };
    unsigned grid = 1;
    if (ff && n_work) {
        if (g->redo.ensure(n_work)) return PW_ERR_NOMEM;
        if (grid_for((const void *)ff, n_work, &grid)) return PW_ERR_HIP;
        HIP_TRY(hipMemsetAsync(g->counters.p, 0, sizeof(unsigned long long), g->stream));
        HIP_TRY(hipMemsetAsync(g->counters.p + 6, 0, sizeof(unsigned long long), g->stream));
        const char *rt = getenv("PECANPY_AMD_DENSE_REDO_TEST");   // tests: every k-th walk is handed over at its third step
        hipLaunchKernelGGL(ff, dim3(grid), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, da, g->redo.p, g->counters.p + 6,
                           rt ? (uint32_t)strtoul(rt, nullptr, 10) : 0u);
        HIP_TRY(hipGetLastError());
        unsigned long long nr = 0;
        HIP_TRY(hipMemcpyAsync(&nr, g->counters.p + 6, sizeof(nr), hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (!nr) return 0;
        if (redo_total) *redo_total += nr;
        da.job_list = g->redo.p;
        da.n_list = nr;
        n_work = nr;
    }
    if (grid_for((const void *)fn, n_work, &grid)) return PW_ERR_HIP;
    HIP_TRY(hipMemsetAsync(g->counters.p, 0, sizeof(unsigned long long), g->stream));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, da);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Per-edge normaliser table of a weighted CSR graph (walk_sparse.hip.h: tot_build_kernel), built on the first call
// with a given (p, q, extend, thresholds) and kept in the handle.  PECANPY_AMD_NO_TOT=1 disables it.
static int ensure_tot_table(pw_graph *g, pw::WalkArgs &wa, bool extend) {
    wa.tot_e = nullptr;
    wa.tot_v = nullptr;
    wa.resume = 0;
    if (g->kind != 0 || g->unit || !g->nnz || g->tot_failed || getenv("PECANPY_AMD_NO_TOT")) return 0;
    const bool fresh = g->tot_extend == (extend ? 1 : 0) && g->tot_p == wa.p && g->tot_q == wa.q &&
                       (!extend || g->tot_thr_version == g->thr_version);
    // The table costs one row scan per CSR entry and saves one per sampled step: a call that samples fewer steps than
    // the graph has entries is faster through the two-pass step (an existing table is used whatever the call's size)
    if (!fresh && wa.n_jobs * (uint64_t)wa.L < (uint64_t)g->nnz && !getenv("PECANPY_AMD_FORCE_TOT")) return 0;
    if (!fresh) {
        if (!g->d_tot_e) {
            hipError_t e = hipMalloc((void **)&g->d_tot_e, sizeof(float) * (size_t)g->nnz);
            if (e == hipSuccess) e = hipMalloc((void **)&g->d_tot_v, sizeof(float) * (size_t)(g->n_nodes ? g->n_nodes : 1));
            if (e != hipSuccess) { g->tot_failed = true; (void)hipGetLastError(); return 0; }
        }
        uint32_t *d_edge_row = nullptr;
        HIP_TRY(hipMalloc((void **)&d_edge_row, sizeof(uint32_t) * (size_t)g->nnz));
        hipLaunchKernelGGL(pw::csr_edge_rows_kernel, dim3(g->n_cu * 8), dim3(256), 0, g->stream, g->d_indptr, g->n_nodes, d_edge_row);
        pw::WalkArgs ba = wa;
        hipError_t e = hipEventRecord(g->ev[4], g->stream);
        if (e == hipSuccess) {
            const uint64_t items = (uint64_t)g->nnz + g->n_nodes;
            const unsigned grid = (unsigned)((items + pw::WAVES_PER_BLOCK - 1) / pw::WAVES_PER_BLOCK);
            if (extend) hipLaunchKernelGGL(pw::tot_build_kernel<true>, dim3(grid), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, ba, d_edge_row, g->d_tot_e, g->d_tot_v);
            else hipLaunchKernelGGL(pw::tot_build_kernel<false>, dim3(grid), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, ba, d_edge_row, g->d_tot_e, g->d_tot_v);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(g->ev[5], g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
        (void)hipFree(d_edge_row);
        if (e != hipSuccess) return fail(PW_ERR_HIP, std::string("normaliser table: ") + hipGetErrorString(e));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, g->ev[4], g->ev[5]));
        g->tot_build_ms = ms;
        g->param_ms_call += ms;
        g->tot_extend = extend ? 1 : 0;
        g->tot_p = wa.p;
        g->tot_q = wa.q;
        g->tot_thr_version = g->thr_version;
    }
    wa.tot_e = g->d_tot_e;
    wa.tot_v = g->d_tot_v;
    return 0;
}

// Tables of the WEIGHTED lane form (walk_lanes.hip.h), per (p, q, extend, thresholds), cached in the handle like the
// normaliser table they go with: base values wb[nnz], their per-row float64 prefix sums wpq[nnz], one float64 per list
// entry of the lane index (delta prefix sums) and the per-entry prev delta.  Returns with *ok = false (no error) when
// the form does not apply: the wave-per-walk kernel then serves the call.
static int ensure_wlane_tables(pw_graph *g, const pw::WalkArgs &wa, bool extend, bool *ok) {
    *ok = false;
    if (g->kind != 0 || g->unit || !g->nnz || !g->d_lines || g->lanes_off || !wa.tot_e || g->wl_failed) return 0;
    if (getenv("PECANPY_AMD_NO_LANES") || getenv("PECANPY_AMD_NO_WLANES")) return 0;
    if (extend && !g->d_thr) return 0;
    // (the float64 evaluation of the prefix differences is part of the decision's error budget: moderate biases only)
    if (!(wa.p >= 1.0 / 1024 && wa.p <= 1024.0 && wa.q >= 1.0 / 1024 && wa.q <= 1024.0)) return 0;
    const bool fresh = g->wl_extend == (extend ? 1 : 0) && g->wl_p == wa.p && g->wl_q == wa.q &&
                       (!extend || g->wl_thr_version == g->thr_version);
    if (fresh) { *ok = true; return 0; }
    const uint32_t nnz = g->nnz;
    auto give_up = [&]() { g->wl_failed = true; (void)hipGetLastError(); return 0; };
    if (!g->d_wb) {
        hipError_t e = hipMalloc((void **)&g->d_wb, sizeof(float) * (size_t)nnz);
        if (e == hipSuccess) e = hipMalloc((void **)&g->d_wpq, sizeof(pw::PrefixPair) * (size_t)nnz);
        if (e == hipSuccess) e = hipMalloc((void **)&g->d_wl_dprev, sizeof(double) * (size_t)nnz);
        if (e == hipSuccess) e = hipMalloc((void **)&g->d_wl_off, sizeof(unsigned long long) * (size_t)nnz);
        if (e == hipSuccess) e = hipMalloc((void **)&g->d_wck_off, sizeof(unsigned long long) * (size_t)nnz);
        const bool have_rows = g->d_wedge_row != nullptr;   // (a handle with a PARTIAL index kept them from its creation: pw_csr_create)
        if (e == hipSuccess && !have_rows) e = hipMalloc((void **)&g->d_wedge_row, sizeof(uint32_t) * (size_t)nnz);
        if (e == hipSuccess) e = hipMalloc((void **)&g->d_wp1, sizeof(pw::PrefixPair) * (size_t)nnz);
        if (e != hipSuccess) return give_up();
        if (!have_rows) hipLaunchKernelGGL(pw::csr_edge_rows_kernel, dim3(g->n_cu * 8), dim3(256), 0, g->stream, g->d_indptr, g->n_nodes, g->d_wedge_row);
        hipLaunchKernelGGL(pw::wprefix_kernel, dim3((unsigned)(((uint64_t)g->n_nodes * pw::WAVE + 255) / 256)), dim3(256), 0, g->stream, g->d_indptr,
                           (const float *)g->d_data, g->n_nodes, g->d_wp1);
    }
    uint64_t *d_tiles = nullptr;
    const uint64_t n_tiles = ((uint64_t)nnz + pw::CL_TILE - 1) / pw::CL_TILE;
    auto cleanup = [&]() { if (d_tiles) (void)hipFree(d_tiles); };
    hipError_t e = hipMalloc((void **)&d_tiles, sizeof(uint64_t) * (n_tiles + 1));
    if (e != hipSuccess) { cleanup(); return give_up(); }
    HIP_TRY(hipEventRecord(g->ev[4], g->stream));
    // offsets of the per-entry delta lists (one float64 per list entry) and of the recorded chain values
    uint64_t entries = 0, records = 0;
    hipLaunchKernelGGL(pw::entry_tile_sums_kernel<0>, dim3((unsigned)n_tiles), dim3(pw::CL_BLOCK), 0, g->stream, g->d_lines, nnz, d_tiles);
    hipLaunchKernelGGL(pw::scan_tile_sums_kernel, dim3(1), dim3(pw::SCAN_BLOCK), 0, g->stream, d_tiles, n_tiles);
    hipLaunchKernelGGL(pw::entry_offsets_kernel<0>, dim3((unsigned)n_tiles), dim3(pw::CL_BLOCK), 0, g->stream, g->d_lines, nnz, d_tiles, g->d_wl_off);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&entries, d_tiles + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pw::entry_tile_sums_kernel<1>, dim3((unsigned)n_tiles), dim3(pw::CL_BLOCK), 0, g->stream, g->d_lines, nnz, d_tiles);
        hipLaunchKernelGGL(pw::scan_tile_sums_kernel, dim3(1), dim3(pw::SCAN_BLOCK), 0, g->stream, d_tiles, n_tiles);
        hipLaunchKernelGGL(pw::entry_offsets_kernel<1>, dim3((unsigned)n_tiles), dim3(pw::CL_BLOCK), 0, g->stream, g->d_lines, nnz, d_tiles, g->d_wck_off);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&records, d_tiles + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e != hipSuccess) { cleanup(); return fail(PW_ERR_HIP, std::string("weighted lane tables (offsets): ") + hipGetErrorString(e)); }
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (entries + 1 > g->wdl_cap) {
        if (g->d_wdl) (void)hipFree(g->d_wdl);
        g->d_wdl = nullptr;
        g->wdl_cap = 0;
        if ((entries + 1) * sizeof(double) > free_b / 2 || hipMalloc((void **)&g->d_wdl, sizeof(double) * (size_t)(entries + 1)) != hipSuccess) {
            cleanup();
            return give_up();
        }
        g->wdl_cap = entries + 1;
    }
    if (records + 1 > g->wck_cap) {
        if (g->d_wck) (void)hipFree(g->d_wck);
        g->d_wck = nullptr;
        g->wck_cap = 0;
        if ((records + 1) * sizeof(float) > free_b / 4 || hipMalloc((void **)&g->d_wck, sizeof(float) * (size_t)(records + 1)) != hipSuccess) {
            cleanup();
            return give_up();
        }
        g->wck_cap = records + 1;
    }
    const unsigned egrid = (unsigned)(((uint64_t)nnz + 255) / 256);
    pw::CsrDev c = csr_dev(g);
    if (extend) hipLaunchKernelGGL(pw::wbase_kernel<true>, dim3(egrid), dim3(256), 0, g->stream, (const float *)g->d_data, g->d_wedge_row, g->d_thr, wa.q, nnz, g->d_wb);
    else hipLaunchKernelGGL(pw::wbase_kernel<false>, dim3(egrid), dim3(256), 0, g->stream, (const float *)g->d_data, g->d_wedge_row, (const float *)nullptr, wa.q, nnz, g->d_wb);
    hipLaunchKernelGGL(pw::wprefix_kernel, dim3((unsigned)(((uint64_t)g->n_nodes * pw::WAVE + 255) / 256)), dim3(256), 0, g->stream, g->d_indptr, g->d_wb,
                       g->n_nodes, g->d_wpq);
    if (extend) hipLaunchKernelGGL(pw::wlist_kernel<true>, dim3(egrid), dim3(256), 0, g->stream, c, g->d_wedge_row, g->d_wb, wa.p, wa.q, g->d_wl_off, g->d_wdl, g->d_wl_dprev);
    else hipLaunchKernelGGL(pw::wlist_kernel<false>, dim3(egrid), dim3(256), 0, g->stream, c, g->d_wedge_row, g->d_wb, wa.p, wa.q, g->d_wl_off, g->d_wdl, g->d_wl_dprev);
    if (records) {   // the chain's value after every CHAIN_CKPT-th element of the rows longer than that, per arriving entry
        const unsigned cgrid = (unsigned)(((uint64_t)nnz + pw::WAVES_PER_BLOCK - 1) / pw::WAVES_PER_BLOCK);
        if (extend) hipLaunchKernelGGL(pw::wckpt_kernel<true>, dim3(cgrid), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, wa, g->d_wedge_row, g->d_wck_off, g->d_wck);
        else hipLaunchKernelGGL(pw::wckpt_kernel<false>, dim3(cgrid), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, wa, g->d_wedge_row, g->d_wck_off, g->d_wck);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(g->ev[5], g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    cleanup();
    if (e != hipSuccess) return fail(PW_ERR_HIP, std::string("weighted lane tables: ") + hipGetErrorString(e));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev[4], g->ev[5]));
    g->param_ms_call += ms;
    g->wl_extend = extend ? 1 : 0;
    g->wl_p = wa.p;
    g->wl_q = wa.q;
    g->wl_thr_version = g->thr_version;
    *ok = true;
    return 0;
}

// weighted dense graphs: the float64-bounded decision, one row stream per step (walk_dense_w.hip.h); the walks it hands
// over (a partial sum inside the bound's interval: ~10^-11 of the steps) are walked again by the complete kernel
static bool dense_weighted_eligible(const pw_graph *g, const pw::WalkArgs &wa, bool extend) {
    if (g->kind != 1 || g->unit || g->bits_only || !g->dense_nonneg || !g->d_adjbits || !g->d_data) return false;
    if (extend && !g->d_thr) return false;
    if (wa.resume || getenv("PECANPY_AMD_DENSE_NO_WFAST")) return false;
    // LDS of one wavefront: prev's packed row + its prefix popcounts + the block prefixes
    const uint64_t lds = (uint64_t)g->words_per_row * 12u + ((uint64_t)g->max_degree / pw::DWBLK_MIN + 2u) * 8u;
    return lds <= 60u * 1024u;
}

static int launch_dense_weighted(pw_graph *g, const pw::WalkArgs &wa, bool extend, uint64_t *redo_total) {
    uint64_t n_work = wa.job_list ? wa.n_list : wa.n_jobs;
    if (!n_work) return 0;
    pw::DenseWArgs da;
    da.indptr = g->d_indptr;
    da.indices = g->d_indices;
    da.data = (const double *)g->d_data;
    da.adjbits = g->d_adjbits;
    da.thr = g->d_thr;
    da.n = g->n_nodes;
    da.wpr = g->words_per_row;
    da.p = wa.p;
    da.q = wa.q;
    da.L = wa.L;
    da.n_jobs = wa.n_jobs;
    da.starts = wa.starts;
    da.stream_off = wa.stream_off;
    da.job_list = wa.job_list;
    da.n_list = wa.n_list;
    da.rng = wa.rng;
    da.rng_base = wa.rng_base;
    da.out = wa.out;
    da.job_counter = wa.job_counter;
    da.stats = wa.stats;
    if (g->redo.ensure(n_work)) return PW_ERR_NOMEM;
    da.redo_list = g->redo.p;
    da.redo_count = g->counters.p + 6;
    const char *rt = getenv("PECANPY_AMD_DENSE_REDO_TEST");   // tests: every k-th walk is handed over at its third step
    da.redo_every = rt ? (uint32_t)strtoul(rt, nullptr, 10) : 0u;
    const char *xt = getenv("PECANPY_AMD_DENSE_EXACT_TEST");  // tests: steps with (job + step) % k == 0 are decided by the in-kernel chain
    da.exact_every = xt ? (uint32_t)strtoul(xt, nullptr, 10) : 0u;
    da.lds_blocks = g->max_degree / pw::DWBLK_MIN + 2u;
    const size_t lds = (size_t)da.wpr * 8u + (size_t)da.lds_blocks * 8u + (size_t)da.wpr * 4u;
    typedef void (*dw_fn)(pw::DenseWArgs);
    dw_fn fn = extend ? pw::walk_dense_weighted_kernel<true> : pw::walk_dense_weighted_kernel<false>;
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)fn, pw::WAVE, lds));
    if (occ < 1) occ = 1;
    if (const char *oe = getenv("PECANPY_AMD_DENSE_W_OCC")) { const int cap = atoi(oe); if (cap >= 1 && cap < occ) occ = cap; }
    uint64_t grid = (uint64_t)g->n_cu * (uint64_t)occ;
    if (grid > n_work) grid = n_work;
    HIP_TRY(hipMemsetAsync(g->counters.p, 0, sizeof(unsigned long long), g->stream));
    HIP_TRY(hipMemsetAsync(g->counters.p + 6, 0, sizeof(unsigned long long), g->stream));
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(pw::WAVE), lds, g->stream, da);
    HIP_TRY(hipGetLastError());
    unsigned long long nr = 0;
    HIP_TRY(hipMemcpyAsync(&nr, g->counters.p + 6, sizeof(nr), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    if (!nr) return 0;
    if (redo_total) *redo_total += nr;
    pw::WalkArgs wr = wa;
    wr.job_list = g->redo.p;
    wr.n_list = nr;
    wr.resume = 0;
    int occ2 = 0;
    walk_kernel_fn cf = pick_kernel(g, extend);
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, (const void *)cf, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    if (occ2 < 1) occ2 = 1;
    uint64_t want = (nr + pw::WAVES_PER_BLOCK - 1) / pw::WAVES_PER_BLOCK;
    uint64_t grid2 = (uint64_t)g->n_cu * (uint64_t)occ2;
    if (grid2 > want) grid2 = want;
    HIP_TRY(hipMemsetAsync(g->counters.p, 0, sizeof(unsigned long long), g->stream));
    hipLaunchKernelGGL(cf, dim3((unsigned)grid2), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, wr);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int launch_wave_walks(pw_graph *g, pw::WalkArgs &wa, bool extend, uint64_t *redo_total = nullptr) {
    // unweighted dense graphs: the column-space kernels at every size (round 6: with prev's row in LDS and DPP sums the packed rows
    // beat the compressed ones from N = 1 000 on -- ER-1k 364 -> 682, ER-8k 207 -> 1 135, ER-12k 158 -> 1 145 M steps/s; rounds 2-5
    // sent matrices of up to 12 000 rows through their compressed rows; PECANPY_AMD_DENSE_SMALL_ROWS=1: that rule)
    const bool small_rows = env_on("PECANPY_AMD_DENSE_SMALL_ROWS");
    if (g->kind == 1 && g->unit && g->d_deg && (g->bits_only || g->n_nodes > 12000 || !small_rows)) return launch_dense_bits(g, wa, redo_total);
    if (dense_weighted_eligible(g, wa, extend)) return launch_dense_weighted(g, wa, extend, redo_total);
    int occ = 0;
    walk_kernel_fn fn = pick_kernel(g, extend);
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)fn, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    if (occ < 1) occ = 1;
    uint64_t n_work = wa.job_list ? wa.n_list : wa.n_jobs;
    uint64_t want = (n_work + pw::WAVES_PER_BLOCK - 1) / pw::WAVES_PER_BLOCK;
    uint64_t grid = (uint64_t)g->n_cu * (uint64_t)occ;
    if (grid > want) grid = want;
    if (grid < 1) grid = 1;
    HIP_TRY(hipMemsetAsync(g->counters.p, 0, sizeof(unsigned long long), g->stream));
    pw::WalkArgs wk = wa;
    if (g->has_loop) wk.lazy_ok = 0;   // (self loops: the lazy step's key classification would meet prev among the common neighbours)
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, wk);
    HIP_TRY(hipGetLastError());
    return 0;
}

static bool lanes_eligible(const pw_graph *g, const pw::WalkArgs &wa) {
    return g->kind == 0 && g->unit && g->d_lines && !g->lanes_off && wa.lazy_ok && !getenv("PECANPY_AMD_NO_LANES");
}

// One lane per walk (walk_lanes.hip.h); the jobs it hands back (overflow reads, rows outside the exact range) are
// walked again by the wave-per-walk kernel.  *n_redo receives their number.
static int launch_lane_walks(pw_graph *g, pw::WalkArgs &wa, uint64_t *n_redo, bool weighted = false, bool extend = false) {
    const uint64_t n_work = wa.job_list ? wa.n_list : wa.n_jobs;
    if (g->redo.ensure(n_work ? n_work : 1)) return PW_ERR_NOMEM;
    pw::LanesArgs la;
    la.lines = g->d_lines;
    la.clist = g->d_clist;
    la.vlines = g->vlines ? 1u : 0u;
    la.vrec = g->d_vrec;
    la.nnz = g->nnz;
    la.L = wa.L;
    la.n_jobs = wa.n_jobs;
    la.starts = wa.starts;
    la.stream_off = wa.stream_off;
    la.job_list = wa.job_list;
    la.n_list = wa.n_list;
    la.rng = wa.rng;
    la.rng_base = wa.rng_base;
    la.out = wa.out;
    la.job_counter = g->counters.p;
    la.stats = g->counters.p + 1;
    la.redo_list = g->redo.p;
    la.redo_count = g->counters.p + 6;
    la.w_out = wa.w_out;
    la.w_prev = wa.w_prev;
    la.wpq = g->d_wpq;
    la.wdl = g->d_wdl;
    la.wl_off = g->d_wl_off;
    la.wl_dprev = g->d_wl_dprev;
    la.tot_e = wa.tot_e;
    la.wp1 = getenv("PECANPY_AMD_NO_WFIRST") ? nullptr : g->d_wp1;
    la.wl_pos = wa.q >= 1.0 ? 1u : 0u;
    la.tot_v = wa.tot_v;
    // TAILS form (round 4: the rest of the edge line staged in LDS by the lane itself; picked for graphs whose lines stay cache
    // resident): superseded by the QUAD form, which fetches the whole line by a quad of lanes -- RMAT-20, 10.5 M jobs: 34.1 ms
    // (TAILS) vs 32.1 (QUAD); kept behind PECANPY_AMD_LANE_TAILS=1 for A/B runs.
    bool tails = false;
    if (const char *te = getenv("PECANPY_AMD_LANE_TAILS")) tails = atoi(te) != 0;
    if (getenv("PECANPY_AMD_VERIFY_TIGHT") || weighted) tails = false;
    int occ = 0;
    if (weighted) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)pw::walk_lanes_kernel<false, false, false, false, true>, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    else if (tails) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)pw::walk_lanes_kernel<false, false, false, true>, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)pw::walk_lanes_kernel<false, false>, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    if (occ < 1) occ = 1;
    int occ_in = 0;
    if (weighted) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_in, (const void *)pw::walk_lanes_kernel<true, false, false, false, true>, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    else if (tails) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_in, (const void *)pw::walk_lanes_kernel<true, false, false, true>, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_in, (const void *)pw::walk_lanes_kernel<true, false>, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    if (occ_in < 1) occ_in = 1;
    // twin contexts (weighted halves side by side): a persistent lane kernel sized for the whole GPU would keep the other half's
    // eager kernel waiting for its workgroups to retire -- four of the six resident workgroups per CU here, six of eight for the
    // eager kernel: C5 341 -> 311 ms per pass (sweep: 3..5 x 4..12, profiles/r06_c5_twin_sweep.txt)
    if (weighted && g->twin_active) { if (occ > 4) occ = 4; if (occ_in > 4) occ_in = 4; }
    if (const char *oe = getenv("PECANPY_AMD_LANE_OCC")) {   // (experiments: fewer resident workgroups per CU than fit)
        const int cap = atoi(oe);
        if (cap >= 1 && cap < occ) occ = cap;
        if (cap >= 1 && cap < occ_in) occ_in = cap;
    }
    const uint64_t lanes_resident = (uint64_t)g->n_cu * (uint64_t)occ * pw::WAVES_PER_BLOCK * pw::WAVE;
    // Steps that need the float32 chain (~1 % on RMAT-22 after lane_tight) are not run in place -- a chain with a few
    // of the wavefront's 64 lanes enabled costs the other lanes ~300 us -- their walks are PARKED in a queue, a
    // separate launch runs all queued chains at full width, and the next round of the lane kernel resumes the walks.
    // Rounds go on while a queue is worth a launch; the last one runs its chains in place.
    const char *tail_env = getenv("PECANPY_AMD_CHAIN_TAIL");
    // (RMAT-22, 21 M walks with neighbours: tail = lanes_resident / 2 -> 6 rounds, 158.7 ms per pass; 4 x -> 4 rounds, 155.2;
    //  16 x -> 160.8: the chains of the last round run at a few lanes per wavefront)
    uint64_t tail = lanes_resident / 2;
    if (n_work / 16 > tail) tail = n_work / 16 < 4 * lanes_resident ? n_work / 16 : 4 * lanes_resident;
    // (weighted form: every walk is parked about nine times -- first step, ambiguous steps -- and the in-place form can only
    //  hand such walks to walk_kernel for good: rounds go on until few walks are left)
    if (weighted) tail = 8192;
    if (tail_env) tail = (uint64_t)strtoull(tail_env, nullptr, 10);
    bool use_queue = getenv("PECANPY_AMD_NO_CHAIN_QUEUE") == nullptr && n_work > tail;
    // Short job lists (at most nine jobs per resident lane) run in ONE in-place launch: with so few walks per lane the launch
    // lasts as long as its slowest walks, and a walk is slower in the queueing form (its deferred steps wait for a full
    // pass of the wavefront's pool, its chains for the next round) -- that form pays off through throughput only.
    // (RMAT-16 / -17 / -18, 0.66 / 1.3 / 2.6 M jobs: 5.1 -> 3.4, 9.7 -> 5.8, 13.0 -> 10.7 ms per pass; RMAT-19, 5.2 M jobs:
    //  18.4 with the queue, 22.1 in place.  On the RMAT-22 graph, 1.3 / 2.6 / 3.2 M jobs: 12.9 -> 10.3, 18.5 -> 17.5,
    //  19.1-20.5 -> 21.0: nine jobs per resident lane is the crossover.  PECANPY_AMD_CHAIN_TAIL set: the queue rule alone decides.)
    // (not with a partial index: without a queue the steps whose list was left out hand their walks to walk_kernel for good)
    if (!weighted && !tail_env && g->list_max_len == 0xffffffffu && n_work <= 9 * lanes_resident) use_queue = false;
    // CHAINS form (round 5): job arrays of a few walks per resident lane -- a shard of a multi-GPU run, RMAT-18..20 sized calls --
    // in ONE launch whose wavefronts run the float chains themselves (from the pool, PW_LANES_CHAIN_TH steps at a time) instead
    // of parking the walks for lanes_chain_kernel and a next round: every round of such a call lasts as long as its slowest
    // walks.  Its chain code costs a wavefront per SIMD and its chain passes run at 20 of 64 lanes, so it pays up to ~32 jobs per
    // resident lane (RMAT-22 graph, 1.3 / 2.6 / 5.2 M jobs: 13.3 -> 7.3, 16.8 -> 11.7, 23.9 -> 21.4 ms per call; 10.5 M: 37.9 vs
    // 39.6; 41.9 M: 118 vs 146); lists of less than a job per lane keep the plain in-place launch.
    // Round 6: with the LATE rounds in the CHAINS form (below) the rounds win from ~16 jobs per resident lane on -- 2.6 / 5.2 /
    // 10.5 M jobs: CHAINS from the first round 10.6 / 19.6 / 36.5 ms per call, rounds + late chains 10.8 / 18.3 / 32.4 (1.3 M: 6.5
    // against 9.4, one in-place launch) -- so the form takes job arrays of up to 16 jobs per resident lane (rounds 5-6: 32).
    // PECANPY_AMD_LANE_CHAINS=0/1 overrides.
    int occ_c = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_c, (const void *)pw::walk_lanes_kernel<false, false, false, false, false, true>, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    if (occ_c < 1) occ_c = 1;
    const uint64_t lanes_resident_c = (uint64_t)g->n_cu * (uint64_t)occ_c * pw::WAVES_PER_BLOCK * pw::WAVE;
    bool chains_form = !weighted && !getenv("PECANPY_AMD_VERIFY_TIGHT") && !tail_env && !getenv("PECANPY_AMD_NO_CHAIN_QUEUE") &&
                       g->list_max_len == 0xffffffffu && n_work >= lanes_resident_c && n_work <= 16 * lanes_resident_c;
    if (const char *ce = getenv("PECANPY_AMD_LANE_CHAINS")) chains_form = atoi(ce) != 0 && !weighted && !getenv("PECANPY_AMD_VERIFY_TIGHT");
    if (chains_form) use_queue = true;      // (a step that finds the pool full is still parked: the round loop below takes care of it)
    // Queue capacities (round 6: from what CAN be parked, not from the job array -- device memory a fresh box has not handed out
    // before costs ~23 ms per GB, and two queues of n_jobs records were 5.4 GB at RMAT-22): only walks whose start has neighbours
    // are ever parked (g->call_runnable of them in a whole-array call: the stream's draws / walk_length), and the second queue
    // only ever holds walks that were in the first -- it is allocated after round 0, for the walks that round parked.
    // + the void slots of every wavefront's LAST reservation (< 128 each; leftovers of earlier ones are used up)
    const uint64_t can_park = (!wa.job_list && g->call_runnable && g->call_runnable < n_work) ? g->call_runnable : n_work;
    const size_t q_cap = (size_t)can_park + 2 * (size_t)lanes_resident;
    if (use_queue && g->susp[0].ensure(q_cap)) {
        (void)hipGetLastError();
        use_queue = false;              // no room for the queues: chains run in place
        chains_form = false;
    }
    HIP_TRY(hipMemsetAsync(g->counters.p + 6, 0, sizeof(unsigned long long), g->stream));
    // VERIFICATION of the interval decision (lane_tight: its inflation constants are argued, not proved -- DESIGN.md section 3).
    //  * test mode (PECANPY_AMD_VERIFY_TIGHT=1): EVERY step it settles is recorded and decided again by the float chain.
    //  * production (round 6, default): a SAMPLE of them -- the steps with ((job + 7919 j) & (N - 1)) == 0, N = 1024, or
    //    PECANPY_AMD_VERIFY_SAMPLE=N (a power of two; 0: off) -- is re-decided the same way, in the same call, counted in
    //    pw_stats.verify_checked / verify_mismatch; a walk with a mismatching record is walked AGAIN by the complete kernel
    //    (walk_kernel: membership searched, the float chain itself) and the host layer warns.  ~1e-4 of the steps: no cost.
    const char *ver_env = getenv("PECANPY_AMD_VERIFY_TIGHT");
    const bool verify_full = ver_env != nullptr && !weighted;
    uint32_t sample_n = 1024;
    if (const char *se = getenv("PECANPY_AMD_VERIFY_SAMPLE")) sample_n = (uint32_t)strtoul(se, nullptr, 10);
    if (sample_n & (sample_n - 1)) { uint32_t pw2 = 1; while (pw2 * 2 <= sample_n) pw2 *= 2; sample_n = pw2; }
    const bool verify_sample = !verify_full && !weighted && !tails && sample_n > 0;
    const bool verify = verify_full || verify_sample;
    la.ver_poison = ((verify_full && strcmp(ver_env, "poison") == 0) || (verify_sample && getenv("PECANPY_AMD_VERIFY_SAMPLE_POISON"))) ? 1u : 0u;
    la.ver_mask = verify_full ? 0u : sample_n - 1u;
    la.ver = nullptr;
    la.ver_count = g->counters.p + 40;
    la.ver_cap = 0;
    const uint32_t VER_JOBS_CAP = 65536;
    if (verify_full) {
        const char *cap_env = getenv("PECANPY_AMD_VERIFY_CAP");
        // room for every step of the launch (on hub-heavy graphs most steps are ambiguous), within half of the free memory
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        uint64_t cap = n_work * (uint64_t)wa.L + 4096;
        if (cap > g->ver.cap) {   // (a larger buffer than the handle holds: only as far as memory allows)
            const uint64_t lim = free_b / 2 / sizeof(pw::VerRec);
            if (cap > lim) cap = lim > g->ver.cap ? lim : g->ver.cap;
        }
        if (cap_env) cap = (uint64_t)strtoull(cap_env, nullptr, 10);
        if (g->ver.ensure(cap) || g->ver_bad.ensure(16)) return PW_ERR_NOMEM;
        la.ver = g->ver.p;
        la.ver_cap = cap;
    } else if (verify_sample) {
        // (a round records at most its settled steps / N; a full buffer drops the excess and counts it: verify_dropped)
        const uint64_t cap = n_work * (uint64_t)wa.L / sample_n / 2 + 65536;
        if (g->ver.ensure(cap) || g->ver_bad.ensure(16) || g->ver_jobs.ensure(VER_JOBS_CAP)) return PW_ERR_NOMEM;
        la.ver = g->ver.p;
        la.ver_cap = cap;
        HIP_TRY(hipMemsetAsync(g->counters.p + 44, 0, 4 * sizeof(unsigned long long), g->stream));
    }
    unsigned long long nr = 0, parked = 0;
    uint64_t todo = n_work;
    int n_rounds = 0;
    const uint64_t late_chains = getenv("PECANPY_AMD_LATE_CHAINS") ? (uint64_t)strtoull(getenv("PECANPY_AMD_LATE_CHAINS"), nullptr, 10) : 16ull;
    for (int round = 0;; round++) {
        // LATE rounds in the CHAINS form (round 6): once a round resumes few enough walks -- the third round of an RMAT-22 pass on --
        // the rest of the pass is a sequence of short launches that each last as long as their slowest walks (lane round, chain
        // launch, lane round, ...); the CHAINS form runs the chains of such a round inside it, so the sequence ends one or two
        // launches earlier.  PECANPY_AMD_LATE_CHAINS = resumed walks per resident lane up to which a round takes it (0: never;
        // default 16).  RMAT-22 pass, same box, same walks: the rounds behind the second, 1.43 M + 0.32 M walks, 2.96 + 2.09 ms ->
        // one round of 3.7 ms; the pass 103.4 -> 102.3 ms.  32 (the second round too: 5.2 M walks): 12.9 ms against 7.7 + 3.0 + 2.1;
        // 64 (every round behind the first): 44.6 against 36.1 (profiles/r06_late_chains.txt).
        const bool chains_late = late_chains > 0 && round >= 1 && !weighted && !tails && use_queue && !verify_full && !tail_env &&
                                 g->list_max_len == 0xffffffffu && todo >= lanes_resident_c && todo <= late_chains * lanes_resident_c;
        const bool chains_now = (chains_form && round == 0) || chains_late;
        bool queue_out = chains_now || (use_queue && todo > tail && round < (weighted ? 512 : 64));
        if (queue_out && round >= 1 && g->susp[round & 1].ensure((size_t)todo + 2 * (size_t)lanes_resident)) {
            (void)hipGetLastError();
            queue_out = false;          // (no room for the other queue: this round runs its chains in place and is the last)
        }
        la.susp = queue_out ? g->susp[round & 1].p : nullptr;
        la.susp_count = g->counters.p + 32;
        la.susp_chunk = todo > 32 * lanes_resident ? 128u : 1u;   // (void slots: < 128 per wavefront)
        la.resume = round ? g->susp[(round - 1) & 1].p : nullptr;
        la.n_resume = round ? todo : 0;
        uint64_t want = (todo + pw::WAVES_PER_BLOCK * pw::WAVE - 1) / (pw::WAVES_PER_BLOCK * pw::WAVE);
        uint64_t grid = (uint64_t)g->n_cu * (uint64_t)(chains_now ? occ_c : queue_out ? occ : occ_in);
        if (grid > want) grid = want;
        if (grid < 1) grid = 1;
        {   // consecutive jobs share pages of the draw stream and of the output: a wavefront reserves runs of them, at
            // most a quarter of its share of the work (balance), a power of two in [64, PW_LANES_CHUNK]
            const uint64_t share = todo / (grid * pw::WAVES_PER_BLOCK * 4);
            uint32_t c = 64;
            while (c * 2 <= PW_LANES_CHUNK && (uint64_t)c * 2 <= share) c *= 2;
            la.job_chunk = c;
        }
        HIP_TRY(hipMemsetAsync(g->counters.p, 0, sizeof(unsigned long long), g->stream));
        HIP_TRY(hipMemsetAsync(g->counters.p + 32, 0, sizeof(unsigned long long), g->stream));
        while (g->round_ev.size() < 2 * (size_t)round + 2) {   // (this round's pair: read behind the last round)
            hipEvent_t ne = nullptr;
            HIP_TRY(hipEventCreate(&ne));
            g->round_ev.push_back(ne);
        }
        HIP_TRY(hipEventRecord(g->round_ev[2 * (size_t)round], g->stream));
        if (verify_full || (verify_sample && round == 0)) HIP_TRY(hipMemsetAsync(g->counters.p + 40, 0, 4 * sizeof(unsigned long long), g->stream));
        const dim3 lgrid((unsigned)grid), lblock(pw::WAVES_PER_BLOCK * pw::WAVE);
        if (chains_now && verify) hipLaunchKernelGGL((pw::walk_lanes_kernel<false, true, false, false, false, true>), lgrid, lblock, 0, g->stream, la);
        else if (chains_now) hipLaunchKernelGGL((pw::walk_lanes_kernel<false, false, false, false, false, true>), lgrid, lblock, 0, g->stream, la);
        else if (weighted && queue_out) hipLaunchKernelGGL((pw::walk_lanes_kernel<false, false, false, false, true>), lgrid, lblock, 0, g->stream, la);
        else if (weighted) hipLaunchKernelGGL((pw::walk_lanes_kernel<true, false, false, false, true>), lgrid, lblock, 0, g->stream, la);
        else if (queue_out && verify) hipLaunchKernelGGL((pw::walk_lanes_kernel<false, true>), lgrid, lblock, 0, g->stream, la);
        else if (queue_out && tails) hipLaunchKernelGGL((pw::walk_lanes_kernel<false, false, false, true>), lgrid, lblock, 0, g->stream, la);
        else if (queue_out) hipLaunchKernelGGL((pw::walk_lanes_kernel<false, false>), lgrid, lblock, 0, g->stream, la);
        else if (verify) hipLaunchKernelGGL((pw::walk_lanes_kernel<true, true>), lgrid, lblock, 0, g->stream, la);
        else if (tails) hipLaunchKernelGGL((pw::walk_lanes_kernel<true, false, false, true>), lgrid, lblock, 0, g->stream, la);
        else hipLaunchKernelGGL((pw::walk_lanes_kernel<true, false>), lgrid, lblock, 0, g->stream, la);
        HIP_TRY(hipGetLastError());
        if (verify_full) {   // the chain decides this round's recorded steps again
            unsigned long long n_rec = 0;
            HIP_TRY(hipMemcpyAsync(&n_rec, g->counters.p + 40, sizeof(n_rec), hipMemcpyDeviceToHost, g->stream));
            HIP_TRY(hipStreamSynchronize(g->stream));
            const unsigned long long n_chk = n_rec < la.ver_cap ? n_rec : la.ver_cap;
            g->ver_dropped += n_rec - n_chk;
            if (n_chk) {
                HIP_TRY(hipMemsetAsync(g->counters.p + 40, 0, 4 * sizeof(unsigned long long), g->stream));
                hipLaunchKernelGGL(pw::lanes_verify_kernel, dim3((unsigned)((n_chk + 255) / 256)), dim3(256), 0, g->stream, g->ver.p,
                                   (uint64_t)n_chk, g->d_lines, g->d_clist, wa.w_prev, g->counters.p + 40, g->ver_bad.p, 16u);
                HIP_TRY(hipGetLastError());
                unsigned long long vc[4] = {0, 0, 0, 0};
                HIP_TRY(hipMemcpyAsync(vc, g->counters.p + 40, sizeof(vc), hipMemcpyDeviceToHost, g->stream));
                HIP_TRY(hipStreamSynchronize(g->stream));
                g->ver_checked += vc[0];
                g->ver_mismatch += vc[1];
                g->ver_ties += vc[2];
                if (vc[1]) {
                    pw::VerRec bad[16];
                    const unsigned nb = vc[3] < 16 ? (unsigned)vc[3] : 16u;
                    HIP_TRY(hipMemcpy(bad, g->ver_bad.p, sizeof(pw::VerRec) * nb, hipMemcpyDeviceToHost));
                    for (unsigned i = 0; i < nb; i++)
                        fprintf(stderr, "[verify] MISMATCH d=%u n_in=%u pp=%u kmax=%u tot=%.9g wo=%g r=%.17g: interval decision %u, float chain %u\n",
                                bad[i].d, bad[i].n_in, bad[i].pp, bad[i].kmax, (double)bad[i].tot, (double)bad[i].wo, bad[i].r, bad[i].choice, bad[i].job);
                }
            }
        }
        HIP_TRY(hipMemcpyAsync(&parked, g->counters.p + 32, sizeof(parked), hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (parked && weighted) {   // the parked steps of the weighted form: one wavefront each, the wave-per-walk scan
            HIP_TRY(hipMemsetAsync(g->counters.p + 13, 0, sizeof(unsigned long long), g->stream));   // (record counter of the persistent grid)
            const uint64_t want_e = (parked + pw::WAVES_PER_BLOCK - 1) / pw::WAVES_PER_BLOCK;
            static const int eager_env = getenv("PECANPY_AMD_EAGER_WGS") ? atoi(getenv("PECANPY_AMD_EAGER_WGS")) : 0;   // (workgroups per CU; experiments)
            const int eager_wgs = eager_env > 0 ? eager_env : (g->twin_active ? 6 : 8);
            const unsigned egrid_e = (unsigned)std::min<uint64_t>(want_e, (uint64_t)g->n_cu * (uint64_t)eager_wgs);
            if (extend) hipLaunchKernelGGL(pw::lanes_eager_weighted_kernel<true>, dim3(egrid_e), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, wa,
                                           g->susp[round & 1].p, (uint64_t)parked, g->counters.p + 12, (const uint32_t *)g->d_wedge_row,
                                           (const unsigned long long *)g->d_wck_off, (const float *)(getenv("PECANPY_AMD_NO_WCKPT") ? nullptr : g->d_wck));
            else hipLaunchKernelGGL(pw::lanes_eager_weighted_kernel<false>, dim3(egrid_e), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, wa,
                                    g->susp[round & 1].p, (uint64_t)parked, g->counters.p + 12, (const uint32_t *)g->d_wedge_row,
                                    (const unsigned long long *)g->d_wck_off, (const float *)(getenv("PECANPY_AMD_NO_WCKPT") ? nullptr : g->d_wck));
            HIP_TRY(hipGetLastError());
        } else if (parked) {   // settle the queue just filled
            hipLaunchKernelGGL(pw::lanes_chain_kernel, dim3((unsigned)((parked + 255) / 256)), dim3(256), 0, g->stream,
                               g->susp[round & 1].p, (uint64_t)parked, g->d_lines, g->d_clist, wa.w_prev, g->counters.p + 1);
            HIP_TRY(hipGetLastError());
            if (g->list_max_len != 0xffffffffu) {   // partial index: the steps whose entry's list was left out (one wavefront each)
                HIP_TRY(hipMemsetAsync(g->counters.p + 13, 0, sizeof(unsigned long long), g->stream));
                const uint64_t want_e = (parked + pw::WAVES_PER_BLOCK - 1) / pw::WAVES_PER_BLOCK;
                hipLaunchKernelGGL(pw::lanes_eager_kernel, dim3((unsigned)std::min<uint64_t>(want_e, (uint64_t)g->n_cu * 8)),
                                   dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, wa, g->susp[round & 1].p, (uint64_t)parked, g->counters.p + 12,
                                   (const uint32_t *)g->d_wedge_row);
                HIP_TRY(hipGetLastError());
            }
        }
        HIP_TRY(hipEventRecord(g->round_ev[2 * (size_t)round + 1], g->stream));
        g->lane_rounds++;
        n_rounds = round + 1;
        if (getenv("PW_DEBUG_ROUNDS")) {
            HIP_TRY(hipStreamSynchronize(g->stream));
            float rms = 0;
            HIP_TRY(hipEventElapsedTime(&rms, g->round_ev[2 * (size_t)round], g->round_ev[2 * (size_t)round + 1]));
            fprintf(stderr, "[lanes] round %d: %llu walks, %llu parked, %.2f ms\n", round, (unsigned long long)todo, parked, rms);
        }
        // (no wait here: the next round's launch -- or the read-back behind the loop -- follows the chain launch in stream order;
        //  the rounds' kernel times are read from their event pairs behind the last round)
        if (!parked) break;
        todo = parked;
    }
    unsigned long long vs[4] = {0, 0, 0, 0}, n_rec_s = 0;
    if (verify_sample) HIP_TRY(hipMemcpyAsync(&n_rec_s, g->counters.p + 40, sizeof(n_rec_s), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipMemcpyAsync(&nr, g->counters.p + 6, sizeof(nr), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    for (int r = 0; r < n_rounds; r++) {
        float rms = 0;
        HIP_TRY(hipEventElapsedTime(&rms, g->round_ev[2 * (size_t)r], g->round_ev[2 * (size_t)r + 1]));
        g->lane_ms += rms;
    }
    if (verify_sample && n_rec_s) {   // the sample of ALL rounds (the records accumulate across them): re-decided by the chain, one launch
        const unsigned long long n_chk = n_rec_s < la.ver_cap ? n_rec_s : la.ver_cap;
        g->ver_dropped += n_rec_s - n_chk;
        HIP_TRY(hipEventRecord(g->ev[4], g->stream));
        hipLaunchKernelGGL(pw::lanes_verify_kernel, dim3((unsigned)((n_chk + 255) / 256)), dim3(256), 0, g->stream, g->ver.p,
                           (uint64_t)n_chk, g->d_lines, g->d_clist, wa.w_prev, g->counters.p + 44, g->ver_bad.p, 16u, g->ver_jobs.p, VER_JOBS_CAP);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(g->ev[5], g->stream));
        HIP_TRY(hipMemcpyAsync(vs, g->counters.p + 44, sizeof(vs), hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        float vms = 0;
        HIP_TRY(hipEventElapsedTime(&vms, g->ev[4], g->ev[5]));
        g->lane_ms += vms;              // (part of the lane path's time: the roofline's kernel time includes it)
    }
    if (verify_full) g->ver.release();   // (test mode: up to half of the free memory -- not kept in the handle)
    if (verify_sample) {
        g->ver_checked += vs[0];
        g->ver_mismatch += vs[1];
        g->ver_ties += vs[2];
        if (vs[1]) {
            // A sampled interval decision disagrees with the float chain.  Never seen (1e9 decisions re-decided
            // on the device, 6e9 fuzzed on the host) -- but the bound is argued, not proved, so: say so, and walk the affected
            // walks again with the complete kernel, whose steps are the reference's float chain over a searched membership mask.
            pw::VerRec bad[16];
            const unsigned nb = vs[3] < 16 ? (unsigned)vs[3] : 16u;
            HIP_TRY(hipMemcpy(bad, g->ver_bad.p, sizeof(pw::VerRec) * nb, hipMemcpyDeviceToHost));
            for (unsigned i = 0; i < nb && getenv("PW_DEBUG_ROUNDS"); i++)
                fprintf(stderr, "[verify] MISMATCH d=%u n_in=%u pp=%u kmax=%u tot=%.9g wo=%g r=%.17g: interval decision %u, float chain %u\n",
                        bad[i].d, bad[i].n_in, bad[i].pp, bad[i].kmax, (double)bad[i].tot, (double)bad[i].wo, bad[i].r, bad[i].choice, bad[i].job);
            pw::WalkArgs wr = wa;
            wr.job_list = g->ver_jobs.p;
            wr.n_list = vs[3] < VER_JOBS_CAP ? vs[3] : VER_JOBS_CAP;
            wr.resume = 0;
            wr.stats = g->counters.p + 20;      // (their transitions were counted by the lane kernel already)
            int rcw = launch_wave_walks(g, wr, false);
            if (rcw) return rcw;
            HIP_TRY(hipStreamSynchronize(g->stream));
        }
    }
#ifdef PW_LANES_WATCHDOG
    {
        unsigned long long wd[32], zero[32] = {0};
        HIP_TRY(hipMemcpyFromSymbol(wd, HIP_SYMBOL(pw::g_wd), sizeof(wd)));
        if (wd[0]) {
            fprintf(stderr, "[watchdog] %llu wavefronts ran away; first: loop %llu block %llu active %016llx waiting %016llx exhausted %016llx "
                    "ambiguous %016llx pool [%llu, %llu) of %llu; lane 0: flags %llu j %llu d %llu n_in %llu job %llu\n", wd[0], wd[1], wd[9], wd[2], wd[3], wd[4],
                    wd[5], wd[6], wd[7], wd[8], wd[10], wd[11], wd[12], wd[13], wd[14]);
        }
        if (wd[16])
            fprintf(stderr, "[watchdog] %llu float chains ran away; first: k %llu kend %llu n_in %llu pp %llu i0 %llu next_in %llu c %08llx r %016llx "
                    "x_in %08llx x_out %08llx\n", wd[16], wd[17], wd[18], wd[19], wd[20], wd[21], wd[22], wd[23], wd[24], wd[25], wd[26]);
        if (wd[0] || wd[16]) HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(pw::g_wd), zero, sizeof(zero)));
    }
#endif
#ifdef PW_PROF_LANES
    {
        unsigned long long hp[16], zero[16] = {0};
        HIP_TRY(hipMemcpyFromSymbol(hp, HIP_SYMBOL(pw::g_lprof), sizeof(hp)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(pw::g_lprof), zero, sizeof(zero)));
        static const char *names[5] = {"refill", "draw + exact decision", "refined decisions", "edge record + store", "float chains"};
        double tot = 0;
        for (int i = 0; i < 5; i++) tot += (double)hp[i];
        for (int i = 0; i < 5; i++)
            fprintf(stderr, "[lane_prof] %-22s %6.2f%%  %.0f cycles per iteration\n", names[i], 100.0 * hp[i] / tot, (double)hp[i] / (double)hp[8]);
        fprintf(stderr, "[lane_prof] iterations %llu  refinement passes %llu (%.1f lanes each, %.0f cycles)  chain passes %llu (%.1f lanes each, %.0f cycles)\n",
                hp[8], hp[9], hp[9] ? (double)hp[10] / (double)hp[9] : 0.0, hp[9] ? (double)hp[2] / (double)hp[9] : 0.0, hp[5],
                hp[5] ? (double)hp[6] / (double)hp[5] : 0.0, hp[5] ? (double)hp[4] / (double)hp[5] : 0.0);
        fprintf(stderr, "[lane_prof] per iteration: deepest search %.2f probes, all probes %.1f, runnable lanes %.1f, lanes on overflow lists %.1f\n",
                (double)hp[11] / (double)hp[8], (double)hp[14] / (double)hp[8], (double)hp[13] / (double)hp[8], (double)hp[12] / (double)hp[8]);
    }
#endif
    *n_redo = nr;
    return 0;
}

// Unit-weight graphs whose 1/p or 1/q is not a power of two: the lane kernel in its FLOATS form -- every step is the
// reference's two float32 chains, evaluated per lane (walk_lanes.hip.h); one launch, nothing is parked.
static bool lanes_float_eligible(const pw_graph *g, const pw::WalkArgs &wa) {
    return g->kind == 0 && g->unit && g->d_lines && !g->lanes_off && !wa.lazy_ok && !getenv("PECANPY_AMD_NO_LANES");
}

// FLOATS form: the row totals of all arriving lines, once per (1/q, 1/p) (walk_lanes.hip.h: unit_tot_kernel; cached in the
// handle, reported as param_index_ms): the step then is ONE bounded decision instead of two float chains.  A call that
// samples fewer steps than the graph has lines runs the two-chain step (an existing table is used whatever the call's size).
static int ensure_unit_tot(pw_graph *g, const pw::WalkArgs &wa) {
    if (g->utot_failed || getenv("PECANPY_AMD_NO_UTOT")) return 0;
    const uint64_t n_lines = (uint64_t)g->nnz + (g->vlines ? g->n_nodes : 0);
    const bool fresh = g->d_utot && g->utot_wo == wa.w_out && g->utot_wp == wa.w_prev;
    if (fresh || !(wa.n_jobs * (uint64_t)wa.L >= n_lines || getenv("PECANPY_AMD_FORCE_TOT"))) return 0;
    if (!g->d_utot && hipMalloc((void **)&g->d_utot, sizeof(float) * (size_t)(n_lines ? n_lines : 1)) != hipSuccess) {
        (void)hipGetLastError();
        g->d_utot = nullptr;
        g->utot_failed = true;
        return 0;
    }
    g->utot_wo = g->utot_wp = 0;
    HIP_TRY(hipEventRecord(g->ev[4], g->stream));
    hipLaunchKernelGGL(pw::unit_tot_kernel, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, g->stream, g->d_lines, g->d_clist, n_lines,
                       wa.w_out, wa.w_prev, g->d_utot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(g->ev[5], g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    float tms = 0;
    HIP_TRY(hipEventElapsedTime(&tms, g->ev[4], g->ev[5]));
    g->param_ms_call += tms;
    g->utot_wo = wa.w_out;
    g->utot_wp = wa.w_prev;
    return 0;
}

static int launch_lane_float_walks(pw_graph *g, pw::WalkArgs &wa, uint64_t *n_redo) {
    const uint64_t n_work = wa.job_list ? wa.n_list : wa.n_jobs;
    if (g->redo.ensure(n_work ? n_work : 1)) return PW_ERR_NOMEM;
    pw::LanesArgs la;
    memset(&la, 0, sizeof(la));
    la.lines = g->d_lines;
    la.clist = g->d_clist;
    la.vlines = g->vlines ? 1u : 0u;
    la.vrec = g->d_vrec;
    la.nnz = g->nnz;
    la.L = wa.L;
    la.n_jobs = wa.n_jobs;
    la.starts = wa.starts;
    la.stream_off = wa.stream_off;
    la.job_list = wa.job_list;
    la.n_list = wa.n_list;
    la.rng = wa.rng;
    la.rng_base = wa.rng_base;
    la.out = wa.out;
    la.job_counter = g->counters.p;
    la.stats = g->counters.p + 1;
    la.redo_list = g->redo.p;
    la.redo_count = g->counters.p + 6;
    la.w_out = wa.w_out;
    la.w_prev = wa.w_prev;
    la.susp_count = g->counters.p + 32;
    la.ver_count = g->counters.p + 40;
    la.susp_chunk = 1;
    if (g->d_utot && g->utot_wo == wa.w_out && g->utot_wp == wa.w_prev && !getenv("PECANPY_AMD_NO_UTOT")) la.tot_e = g->d_utot;   // (ensure_unit_tot)
    // verification of the interval decision (round 6: lane_tight_values), as in launch_lane_walks: every settled step in test
    // mode (PECANPY_AMD_VERIFY_TIGHT=1), a sample of them in production (PECANPY_AMD_VERIFY_SAMPLE=N, default 1024, 0 = off)
    const char *ver_env = getenv("PECANPY_AMD_VERIFY_TIGHT");
    const bool verify_full = ver_env != nullptr && la.tot_e != nullptr;
    uint32_t sample_n = 1024;
    if (const char *se = getenv("PECANPY_AMD_VERIFY_SAMPLE")) sample_n = (uint32_t)strtoul(se, nullptr, 10);
    if (sample_n & (sample_n - 1)) { uint32_t pw2 = 1; while (pw2 * 2 <= sample_n) pw2 *= 2; sample_n = pw2; }
    const bool verify_sample = !verify_full && la.tot_e != nullptr && sample_n > 0;
    const bool verify = verify_full || verify_sample;
    const uint32_t VER_JOBS_CAP = 65536;
    if (verify) {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        uint64_t cap = verify_full ? n_work * (uint64_t)wa.L + 4096 : n_work * (uint64_t)wa.L / sample_n / 2 + 65536;
        const uint64_t lim = free_b / 2 / sizeof(pw::VerRec);
        if (cap > g->ver.cap && cap > lim) cap = lim > g->ver.cap ? lim : g->ver.cap;
        if (const char *cap_env = getenv("PECANPY_AMD_VERIFY_CAP")) cap = (uint64_t)strtoull(cap_env, nullptr, 10);
        if (g->ver.ensure(cap) || g->ver_bad.ensure(16) || g->ver_jobs.ensure(VER_JOBS_CAP)) return PW_ERR_NOMEM;
        la.ver = g->ver.p;
        la.ver_cap = cap;
        la.ver_mask = verify_full ? 0u : sample_n - 1u;
        la.ver_poison = ((verify_full && strcmp(ver_env, "poison") == 0) || (verify_sample && getenv("PECANPY_AMD_VERIFY_SAMPLE_POISON"))) ? 1u : 0u;
        HIP_TRY(hipMemsetAsync(g->counters.p + 40, 0, 8 * sizeof(unsigned long long), g->stream));
    }
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)pw::walk_lanes_kernel<true, false, true>, pw::WAVES_PER_BLOCK * pw::WAVE, 0));
    if (occ < 1) occ = 1;
    uint64_t want = (n_work + pw::WAVES_PER_BLOCK * pw::WAVE - 1) / (pw::WAVES_PER_BLOCK * pw::WAVE);
    uint64_t grid = (uint64_t)g->n_cu * (uint64_t)occ;
    if (grid > want) grid = want;
    if (grid < 1) grid = 1;
    {   // (job chunks as in launch_lane_walks; smaller: a lane's step is ~10x longer here and the tail matters more)
        const uint64_t share = n_work / (grid * pw::WAVES_PER_BLOCK * 8);
        uint32_t c = 64;
        while (c * 2 <= 256 && (uint64_t)c * 2 <= share) c *= 2;
        la.job_chunk = c;
    }
    HIP_TRY(hipMemsetAsync(g->counters.p, 0, sizeof(unsigned long long), g->stream));
    HIP_TRY(hipMemsetAsync(g->counters.p + 6, 0, sizeof(unsigned long long), g->stream));
    HIP_TRY(hipEventRecord(g->ev[4], g->stream));
    if (verify) hipLaunchKernelGGL((pw::walk_lanes_kernel<true, true, true>), dim3((unsigned)grid), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, la);
    else hipLaunchKernelGGL((pw::walk_lanes_kernel<true, false, true>), dim3((unsigned)grid), dim3(pw::WAVES_PER_BLOCK * pw::WAVE), 0, g->stream, la);
    HIP_TRY(hipGetLastError());
    unsigned long long nr = 0, n_rec = 0;
    if (verify) HIP_TRY(hipMemcpyAsync(&n_rec, g->counters.p + 40, sizeof(n_rec), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipMemcpyAsync(&nr, g->counters.p + 6, sizeof(nr), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    unsigned long long vs[4] = {0, 0, 0, 0};
    if (verify && n_rec) {   // the recorded steps, decided again by the float chain over the whole row
        const unsigned long long n_chk = n_rec < la.ver_cap ? n_rec : la.ver_cap;
        g->ver_dropped += n_rec - n_chk;
        hipLaunchKernelGGL(pw::lanes_verify_kernel, dim3((unsigned)((n_chk + 255) / 256)), dim3(256), 0, g->stream, g->ver.p, (uint64_t)n_chk,
                           g->d_lines, g->d_clist, wa.w_prev, g->counters.p + 44, g->ver_bad.p, 16u, g->ver_jobs.p, VER_JOBS_CAP, 1u);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(vs, g->counters.p + 44, sizeof(vs), hipMemcpyDeviceToHost, g->stream));
    }
    HIP_TRY(hipEventRecord(g->ev[5], g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev[4], g->ev[5]));
    g->lane_ms += ms;
    g->lane_rounds++;
    if (verify_full) g->ver.release();
    if (verify) {
        g->ver_checked += vs[0];
        g->ver_mismatch += vs[1];
        g->ver_ties += vs[2];
        if (vs[1]) {
            pw::VerRec bad[16];
            const unsigned nb = vs[3] < 16 ? (unsigned)vs[3] : 16u;
            HIP_TRY(hipMemcpy(bad, g->ver_bad.p, sizeof(pw::VerRec) * nb, hipMemcpyDeviceToHost));
            for (unsigned i = 0; i < nb && (verify_full || getenv("PW_DEBUG_ROUNDS")); i++)
                fprintf(stderr, "[verify] MISMATCH (FLOATS) d=%u n_in=%u pp=%u tot=%.9g wo=%g r=%.17g: interval decision %u, float chain %u\n",
                        bad[i].d, bad[i].n_in, bad[i].pp, (double)bad[i].tot, (double)bad[i].wo, bad[i].r, bad[i].choice, bad[i].job);
            if (verify_sample) {   // the affected walks again, by the complete kernel (as in launch_lane_walks)
                pw::WalkArgs wr = wa;
                wr.job_list = g->ver_jobs.p;
                wr.n_list = vs[3] < VER_JOBS_CAP ? vs[3] : VER_JOBS_CAP;
                wr.resume = 0;
                wr.stats = g->counters.p + 20;
                int rcw = launch_wave_walks(g, wr, false);
                if (rcw) return rcw;
                HIP_TRY(hipStreamSynchronize(g->stream));
            }
        }
    }
    *n_redo = nr;
    return 0;
}

static int launch_walks(pw_graph *g, pw::WalkArgs &wa, bool extend, uint64_t *redo_total) {
    const bool floats = lanes_float_eligible(g, wa);
    // weighted lane form: whole job arrays only (repair passes of a directed graph run on job lists: the wave kernel), and
    // only when the rounds have a queue -- the in-place form can do nothing with a step it cannot decide (first steps
    // included) but hand the walk to walk_kernel
    bool wlanes = g->wl_active && !wa.job_list;
    if (wlanes) {
        const char *tail_env = getenv("PECANPY_AMD_CHAIN_TAIL");
        const uint64_t wtail = tail_env ? (uint64_t)strtoull(tail_env, nullptr, 10) : 8192ull;
        if (wa.n_jobs <= wtail || getenv("PECANPY_AMD_NO_CHAIN_QUEUE")) wlanes = false;
    }
    if (wlanes) g->wl_used = true;
    if (!floats && !wlanes && !lanes_eligible(g, wa)) return launch_wave_walks(g, wa, extend, redo_total);
    uint64_t n_redo = 0;
    int rc = wlanes ? launch_lane_walks(g, wa, &n_redo, true, extend)
                    : (floats ? launch_lane_float_walks(g, wa, &n_redo) : launch_lane_walks(g, wa, &n_redo));
    // Walks the lane kernel cannot step (overflow reads, rows outside the exact range, tie budget) go to walk_kernel,
    // which resumes them at that step and finishes them.  Measured and rejected: handing them BACK to the lane kernel
    // once they are on a CSR entry again -- as extra cycles after the rounds (198 vs 189 ms per RMAT-22 pass, round 2)
    // or after ONE eager step, every round followed by a walk_kernel launch over its redo list (200 vs 181 ms, round
    // 3): such walks overflow again and again, an eager step on a 97 k-entry hub row takes milliseconds, and each
    // launch waits for the slowest of them.  Also measured: finishing the first round's redo walks on a second stream
    // BESIDE the later rounds (179.5 vs 178.0 ms): six 80-register wavefronts of the lane kernel fill the SIMDs' register
    // files, walk_kernel's wavefronts only get in as lane wavefronts retire -- nothing overlaps.
    if (rc || !n_redo) return rc;
    if (redo_total) *redo_total += n_redo;
    pw::WalkArgs wr = wa;
    wr.job_list = g->redo.p;
    wr.n_list = n_redo;
    wr.resume = 1u;   // from the step the lane kernel stopped at (its rows hold the walks so far)
    return launch_wave_walks(g, wr, extend);
}

// MT19937 doubles covering [stream_skip, stream_skip + total) of the stream seeded with `seed`, expanded into g->rng
// (double #(first_block * 312 + k) at g->rng.p[k]): n_gen generators, each expanding `per_gen` (a power of two)
// consecutive blocks of 312 doubles; generator states come from the seed state by polynomial jump-ahead on the device
// (binary tree of x^(624*2^m) jumps).  `cacheable`: the generator states may be taken from / kept in the handle's
// cache (a repeated call with the same seed and shape skips the jump launches).  Events ev[0] / ev[1] bracket the
// kernels on g->stream.
static int expand_stream(pw_graph *g, uint32_t seed, bool cacheable, uint64_t stream_skip, uint64_t total, uint64_t *rng_base) {
    const uint64_t first_block = stream_skip / 312;
    const uint64_t end_block = (stream_skip + total + 311) / 312;
    const uint64_t n_blocks = end_block > first_block ? end_block - first_block : 1;
    if (g->rng_hold.valid && g->rng_hold.seed == seed && first_block >= g->rng_hold.first_block &&
        first_block + n_blocks <= g->rng_hold.first_block + g->rng_hold.n_blocks) {
        HIP_TRY(hipEventRecord(g->ev[0], g->stream));   // (nothing to do: the callers still read the pair of events)
        HIP_TRY(hipEventRecord(g->ev[1], g->stream));
        *rng_base = g->rng_hold.first_block * 312;
        return 0;
    }
    g->rng_hold.valid = false;      // (g->rng is about to be overwritten: whatever was held is gone)
    g->rng_hold.user = false;
    if (g->rng.ensure(n_blocks * 312)) return PW_ERR_NOMEM;
    uint64_t per_gen = 1;
    int per_gen_log = 0;
    // generators: each level of the jump tree is a launch or two, each generator expands its blocks one after the other --
    // short streams want fewer generators (RMAT-18, 381 k blocks: 256 / 512 / 1024 / 2048 / 4096 -> 1.72 / 1.62 / 2.06 /
    // 2.66 / 3.33 ms), long ones more (RMAT-22, 5.2 M blocks: 1024 / 2048 / 4096 -> 7.8 / 7.2 / 8.9 ms)
    static const uint64_t gens_env = getenv("PECANPY_AMD_MT_GENS") ? (uint64_t)strtoull(getenv("PECANPY_AMD_MT_GENS"), nullptr, 10) : 0ull;
    const uint64_t gens_target = gens_env ? gens_env : (n_blocks < (1ull << 21) ? 512ull : 2048ull);
    while (per_gen * gens_target < n_blocks) { per_gen <<= 1; per_gen_log++; }
    const uint32_t n_gen = (uint32_t)((n_blocks + per_gen - 1) / per_gen);
    if (!g->jump_table_ready) {
        const size_t words = (size_t)(pw::MtJump::MAX_POW2 + 1) * pw::MT_PW;
        if (g->jump_table.ensure(words)) return PW_ERR_NOMEM;
        HIP_TRY(hipMemcpy(g->jump_table.p, pw::MtJump::instance().pow2_table(), words * sizeof(uint64_t),
                          hipMemcpyHostToDevice));
        g->jump_table_ready = true;
    }
    if ((first_block >> (pw::MtJump::MAX_POW2 + 1)) != 0) return fail(PW_ERR_INVALID, "stream offset too large");
    {   // every jump the tree below needs must be in the table -- checked before anything is launched
        uint32_t top = 1;
        int top_log = 0;
        while (top < n_gen) { top <<= 1; top_log++; }
        if (top_log > 0 && top_log - 1 + per_gen_log > pw::MtJump::MAX_POW2) return fail(PW_ERR_INVALID, "stream too long");
    }
    // generator states: from the cache when this (seed, offset, shape) was expanded before
    pw_graph::MtCache *mc = nullptr;
    for (auto &c : g->mt_cache)
        if (c.valid && c.seed == seed && c.first_block == first_block && c.per_gen_log == per_gen_log && c.n_gen == n_gen) mc = &c;
    const bool mt_hit = mc != nullptr && cacheable;
    if (!mt_hit) {
        mc = &g->mt_cache[0];
        for (auto &c : g->mt_cache)
            if (!c.valid) { mc = &c; break; } else if (c.stamp < mc->stamp) mc = &c;
        mc->valid = false;
        if (mc->states.ensure((size_t)pw::MT_N * n_gen)) return PW_ERR_NOMEM;
        if (!g->jump_tmp.p) {   // (a jump's polynomial taps are split over several workgroups: partial results, kept zeroed)
            if (g->jump_tmp.ensure((size_t)256 * pw::MT_N)) return PW_ERR_NOMEM;
            HIP_TRY(hipMemsetAsync(g->jump_tmp.p, 0, sizeof(uint32_t) * 256 * pw::MT_N, g->stream));
        }
        if (!g->seed_state) HIP_TRY(hipHostMalloc((void **)&g->seed_state, sizeof(uint32_t) * pw::MT_N, hipHostMallocDefault));
        HIP_TRY(hipStreamSynchronize(g->stream));   // (the pinned seed state of an earlier call has been consumed)
        pw::mt_seed_state(g->seed_state, seed);
        HIP_TRY(hipMemcpyAsync(mc->states.p, g->seed_state, sizeof(uint32_t) * pw::MT_N, hipMemcpyHostToDevice, g->stream));
    }
    mc->stamp = ++g->mt_stamp;
    uint32_t *const mt_states = mc->states.p;
    HIP_TRY(hipEventRecord(g->ev[0], g->stream));
    if (!mt_hit) {
        // a jump's polynomial taps are split over several workgroups while a level has fewer jumps than CUs
        auto jump = [&](uint32_t jumps, const uint64_t *poly, uint32_t src_stride, uint32_t dst_offset) {
            uint32_t parts = jumps >= 128 ? 1u : (uint32_t)(g->n_cu > 0 ? g->n_cu : 256) / jumps;
            if (parts > 39) parts = 39;
            if (parts < 1) parts = 1;
            static const bool v1 = env_on("PECANPY_AMD_MT_JUMP_V1");   // (the tap loop of rounds 1-5, for same-box comparisons)
            if (v1) hipLaunchKernelGGL(pw::mt_jump_v1_kernel, dim3(jumps * parts), dim3(640), 0, g->stream, mt_states, poly, src_stride,
                                       dst_offset, parts, g->jump_tmp.p);
            else hipLaunchKernelGGL(pw::mt_jump_kernel, dim3(jumps * parts), dim3(640), 0, g->stream, mt_states, poly, src_stride,
                                    dst_offset, parts, g->jump_tmp.p);
            if (parts > 1)
                hipLaunchKernelGGL(pw::mt_jump_store_kernel, dim3(jumps), dim3(640), 0, g->stream, mt_states, g->jump_tmp.p,
                                   src_stride, dst_offset);
        };
        for (int m = 0; m <= pw::MtJump::MAX_POW2; m++)  // generator 0 -> first_block
            if ((first_block >> m) & 1) jump(1u, g->jump_table.p + (size_t)m * pw::MT_PW, 0u, 0u);
        uint32_t top = 1;
        int top_log = 0;
        while (top < n_gen) { top <<= 1; top_log++; }
        for (int lvl = top_log - 1; lvl >= 0; lvl--) {  // generator i -> i + 2^lvl, for i % 2^(lvl+1) == 0
            uint32_t s = 1u << lvl;
            if (s >= n_gen) continue;
            uint32_t pairs = (n_gen - s + 2 * s - 1) / (2 * s);
            int m = lvl + per_gen_log;
            jump(pairs, g->jump_table.p + (size_t)m * pw::MT_PW, 2 * s, s);
        }
        mc->valid = true;
        mc->seed = seed; mc->first_block = first_block; mc->per_gen_log = per_gen_log; mc->n_gen = n_gen;
    }
    // (measured and not kept, round 6: the block in two LDS copies -- three barriers per block instead of six -- with 64 / 128 /
    //  256 lanes per generator: 8.4 / 6.9 / 6.06 ms of jump + expansion per RMAT-22 pass against 6.08-6.16: the expansion is bound
    //  neither by its barriers nor by its stores (3 TB/s; a plain fill of the 12.9 GB: 4.8 TB/s) but by the per-generator chain of LDS
    //  round trips at eight workgroups per CU)
    hipLaunchKernelGGL(pw::mt_expand_kernel, dim3(n_gen), dim3(256), 0, g->stream, mt_states,
                       (uint32_t *)nullptr, g->rng.p, per_gen, n_blocks);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(g->ev[1], g->stream));
    *rng_base = first_block * 312;
    return 0;
}

// The draws [stream_skip, stream_skip + n_draws) of `seed`'s stream expanded ONCE and kept: pw_simulate_device calls with that
// seed whose range lies inside find their draws in place (no jump-ahead tree, no expansion) until pw_stream_release or the
// next pw_stream_hold -- for a shard walked in chunks (every chunk would pay ~3 ms of sequential jump launches otherwise).
PW_EXPORT int pw_stream_hold(pw_graph *g, uint32_t seed, uint64_t stream_skip, uint64_t n_draws) {
    if (!g) return fail(PW_ERR_INVALID, "null pointer");
    if (set_device(g)) return PW_ERR_HIP;
    if (g->counters.ensure(N_COUNTERS)) return PW_ERR_NOMEM;
    g->rng_hold.valid = false;
    g->rng_hold.user = false;
    if (!n_draws) return PW_OK;
    uint64_t base = 0;
    int rc = expand_stream(g, seed, true, stream_skip, n_draws, &base);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(g->stream));
    g->rng_hold.valid = true;
    g->rng_hold.user = true;
    g->rng_hold.seed = seed;
    g->rng_hold.first_block = base / 312;
    g->rng_hold.n_blocks = (stream_skip + n_draws + 311) / 312 > base / 312 ? (stream_skip + n_draws + 311) / 312 - base / 312 : 1;
    return PW_OK;
}
PW_EXPORT int pw_stream_release(pw_graph *g) {
    if (!g) return fail(PW_ERR_INVALID, "null pointer");
    g->rng_hold.valid = false;
    g->rng_hold.user = false;
    return PW_OK;
}

// Test hook: doubles #offset .. #offset + n of RandomState(seed).random_sample as the DEVICE produces them (jump tree +
// expansion kernels of a walk call), copied to the host.
PW_EXPORT int pw_stream_sample_device(pw_graph *g, uint32_t seed, uint64_t offset, uint64_t n, double *out) {
    if (!g || (n && !out)) return fail(PW_ERR_INVALID, "null pointer");
    if (set_device(g)) return PW_ERR_HIP;
    if (!n) return PW_OK;
    uint64_t base = 0;
    int rc = expand_stream(g, seed, false, offset, n, &base);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, g->rng.p + (offset - base), sizeof(double) * n, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return PW_OK;
}

static int simulate_device_impl(pw_graph *g, int mode, double p, double q, int extend,
                                const uint32_t *d_starts, uint64_t n_jobs, uint32_t walk_length,
                                int has_seed, uint32_t seed, uint64_t stream_skip, uint32_t *d_out,
                                pw_stats *stats) {
    if (!g || (n_jobs && (!d_starts || !d_out))) return fail(PW_ERR_INVALID, "null pointer");
    if (mode < PW_MODE_SPARSE_OTF || mode > PW_MODE_PRECOMP_FIRST_ORDER) return fail(PW_ERR_INVALID, "unknown mode");
    if (mode == PW_MODE_SPARSE_OTF && g->kind != 0) return fail(PW_ERR_UNSUPPORTED, "SparseOTF needs a CSR graph handle");
    if (mode == PW_MODE_DENSE_OTF && g->kind != 1) return fail(PW_ERR_UNSUPPORTED, "DenseOTF needs a dense graph handle");
    if (extend && !g->unit && !g->d_thr)
        return fail(PW_ERR_INVALID, "extend: call pw_graph_set_thresholds() first");
    if (!(p > 0) || !(q > 0)) return fail(PW_ERR_INVALID, "p and q must be positive");
    if (walk_length < 1) return fail(PW_ERR_INVALID, "walk_length must be >= 1");
    if (n_jobs >= 0xffffffffull) return fail(PW_ERR_INVALID, "n_jobs must fit uint32");
    if (set_device(g)) return PW_ERR_HIP;
    pw_stats st;
    memset(&st, 0, sizeof(st));
    if (n_jobs == 0) { if (stats) *stats = st; return PW_OK; }
    if (!has_seed) seed = os_seed();
    if (g->counters.ensure(N_COUNTERS)) return PW_ERR_NOMEM;
    { int rcs = check_starts(g, d_starts, n_jobs); if (rcs) return rcs; }
    if (mode >= PW_MODE_PRECOMP) {
        if (stream_skip) return fail(PW_ERR_UNSUPPORTED, "alias / first-order modes consume a variable number of "
                                                         "words per step: the stream cannot be sharded");
        int rcs = simulate_sequential(g, mode, p, q, extend, d_starts, n_jobs, walk_length, has_seed, seed, d_out, &st);
        if (!rcs && stats) *stats = st;
        return rcs;
    }

    // 1. stream offsets (nominal: every walk from a start with neighbours runs L steps)
    uint64_t total = 0;
    int rc = compute_offsets(g, d_starts, nullptr, walk_length, n_jobs, stream_skip, false, &total, nullptr);
    if (rc) return rc;

    g->call_runnable = total / walk_length;
    const bool lanes_pre = g->kind == 0 && g->unit && g->d_lines && !g->lanes_off && mode == PW_MODE_SPARSE_OTF;
    // (zero-fill of the walk matrix for the lane kernel: on the side stream, overlapped with the stream expansion.  No
    //  return path may leave that write to the CALLER's buffer in flight: the guard waits for the side stream.)
    struct SideGuard {
        hipStream_t s;
        bool armed;
        ~SideGuard() { if (armed) (void)hipStreamSynchronize(s); }
    } side_guard{g->stream2, false};
    if (lanes_pre) {
        side_guard.armed = true;
        HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(uint32_t) * (size_t)n_jobs * ((size_t)walk_length + 2), g->stream2));
        HIP_TRY(hipEventRecord(g->ev_side, g->stream2));
    }
    // 2. the doubles [stream_skip, stream_skip + total) of the seed's MT19937 stream, expanded on the device
    uint64_t rng_base = 0;
    rc = expand_stream(g, seed, has_seed != 0, stream_skip, total, &rng_base);
    if (rc) return rc;

    // 3. walks
    HIP_TRY(hipMemsetAsync(g->counters.p, 0, N_COUNTERS * sizeof(unsigned long long), g->stream));
    pw::WalkArgs wa;
    wa.g = csr_dev(g);
    wa.p = p;
    wa.q = q;
    wa.L = walk_length;
    wa.n_jobs = n_jobs;
    wa.starts = d_starts;
    wa.stream_off = g->stream_off.p;
    wa.job_list = nullptr;
    wa.n_list = 0;
    wa.rng = g->rng.p;
    wa.rng_base = rng_base;
    wa.out = d_out;
    wa.job_counter = g->counters.p;
    wa.stats = g->counters.p + 1;
    wa.tot_e = nullptr;
    wa.tot_v = nullptr;
    wa.resume = 0;
    {
        // unit-weight biases exactly as the kernels form them: fl32(f64(1.0f) / q) (sparse_rw.py:59-62)
        wa.w_out = (float)(1.0 / q);
        wa.w_prev = (float)(1.0 / p);
        auto pow2_ok = [](float w) {
            int e = 0;
            return std::frexp(w, &e) == 0.5f && e > -60 && e < 60;
        };
        wa.lazy_ok = (pow2_ok(wa.w_out) && pow2_ok(wa.w_prev)) ? 1u : 0u;
    }
    g->param_ms_call = 0;
    rc = ensure_tot_table(g, wa, extend != 0);   // (before the timed walk region: a per-(p, q) index, reported apart)
    if (rc) return rc;
    if (mode == PW_MODE_SPARSE_OTF && lanes_float_eligible(g, wa)) { rc = ensure_unit_tot(g, wa); if (rc) return rc; }
    g->wl_active = false;
    g->wl_used = false;
    const char *wtail_env = getenv("PECANPY_AMD_CHAIN_TAIL");
    const uint64_t wtail = wtail_env ? (uint64_t)strtoull(wtail_env, nullptr, 10) : 8192ull;
    if (mode == PW_MODE_SPARSE_OTF && n_jobs > wtail && !getenv("PECANPY_AMD_NO_CHAIN_QUEUE")) {   // (the rule of launch_walks)
        rc = ensure_wlane_tables(g, wa, extend != 0, &g->wl_active);   // (likewise; the weighted lane form needs both)
        if (rc) return rc;
        if (g->wl_active) {   // the lane kernel only writes the cells a walk fills (unit graphs: zero-filled on the side stream above)
            HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(uint32_t) * (size_t)n_jobs * ((size_t)walk_length + 2), g->stream));
        }
    }
    if (g->on_tables_ready) { auto cb = g->on_tables_ready; g->on_tables_ready = nullptr; cb(); }
    uint64_t redo_total = 0;
    const bool lanes = lanes_eligible(g, wa) || lanes_float_eligible(g, wa);
    g->lane_ms = 0;
    g->lane_rounds = 0;
    g->ver_checked = g->ver_mismatch = g->ver_dropped = g->ver_ties = 0;
    HIP_TRY(hipEventRecord(g->ev[2], g->stream));
    // the lane kernel only writes the cells a walk fills: the matrix starts zeroed (pecanpy.py:182-187) -- by the side
    // stream, under the stream expansion above
    if (lanes_pre) HIP_TRY(hipStreamWaitEvent(g->stream, g->ev_side, 0));
    rc = launch_walks(g, wa, extend != 0, &redo_total);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(g->ev[3], g->stream));
    unsigned long long h[N_COUNTERS];
    HIP_TRY(hipMemcpyAsync(h, g->counters.p, sizeof(h), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
#ifdef PW_PROF
    {
        unsigned long long hp[16], zero[16] = {0};
        HIP_TRY(hipMemcpyFromSymbol(hp, HIP_SYMBOL(pw::g_prof), sizeof(hp)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(pw::g_prof), zero, sizeof(zero)));
        static const char *names[16] = {"step pro/epilogue", "lazy setup", "membership chunks", "build_rank", "seq_head",
                                        "unit_chain", "eager fallback", "exact search", "#chunks", "#searches", "#eager steps",
                                        "#ambiguous", "#chain tie fallbacks", "#search rounds", "-", "-"};
        double tot = 0;
        for (int i = 0; i < 8; i++) tot += (double)hp[i];
        fprintf(stderr, "[pw_prof] steps=%llu\n", h[1]);
        for (int i = 0; i < 8; i++)
            fprintf(stderr, "[pw_prof] %-20s %6.2f%%  %.1f cycles/step\n", names[i], 100.0 * hp[i] / tot, (double)hp[i] / (double)h[1]);
        for (int i = 8; i < 16; i++) fprintf(stderr, "[pw_prof] %-20s %.3f /step\n", names[i], (double)hp[i] / (double)h[1]);
    }
#endif
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev[0], g->ev[1]));
    st.rng_kernel_ms = ms;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev[2], g->ev[3]));
    st.walk_kernel_ms = ms;
    st.walk_kernel_launches = 1;

    // 4. dead ends shift the stream addresses of every later walk: re-address and re-run the
    //    affected walks until the addressing is self-consistent (directed graphs only).
    //    Each round makes a longer prefix of the job array final.  With few dead ends that converges in a handful of
    //    rounds over the whole array; on sink-heavy directed graphs the single-stream semantics (pecanpy.py:198-206)
    //    is inherently sequential -- a re-addressed walk draws new numbers, may end elsewhere and shifts everything
    //    behind it again -- so after max_rounds the repair goes on BLOCK-WISE and stays exact: only the jobs of a
    //    window of `blk` jobs behind the first inconsistent one are re-addressed and re-walked per round (the jobs
    //    behind the window keep the offsets they were last walked with and are not touched until the window reaches
    //    them), O(dead-end walks) small launches instead of O(dead-end walks) passes over the whole array.  Slow but
    //    exact; pw_stats.repair_rounds counts the rounds.  NOMINAL addressing -- every walk owns a fixed slot of
    //    walk_length draws: reproducible under the seed and statistically equivalent, but not the reference's
    //    draw-for-draw assignment -- is an explicit opt-in (PECANPY_AMD_NOMINAL_STREAM=1), reported through
    //    pw_stats.stream_addressing = 1.
    uint64_t dead = h[4];
    const uint64_t max_rounds = 32;
    const bool nominal_ok = env_on("PECANPY_AMD_NOMINAL_STREAM");
    // (a) whole-array rounds
    uint64_t first = ~0ull;          // smallest job whose offset changed in the last whole-array round (re-walked since: final)
    bool consistent = dead == 0;
    while (!consistent && st.repair_rounds < max_rounds) {
        uint64_t tot2 = 0, n_changed = 0;
        rc = compute_offsets(g, d_starts, d_out, walk_length, n_jobs, stream_skip, true, &tot2, &n_changed, 0, n_jobs, &first);
        if (rc) return rc;
        if (getenv("PW_DEBUG_ROUNDS")) fprintf(stderr, "[repair] round %llu: %llu jobs re-addressed\n", (unsigned long long)st.repair_rounds, (unsigned long long)n_changed);
        if (n_changed == 0) { consistent = true; break; }
        st.repair_rounds++;
        wa.job_list = g->changed.p;
        wa.n_list = n_changed;
        HIP_TRY(hipEventRecord(g->ev[2], g->stream));
        rc = launch_walks(g, wa, extend != 0, &redo_total);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(g->ev[3], g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        HIP_TRY(hipEventElapsedTime(&ms, g->ev[2], g->ev[3]));
        st.walk_kernel_ms += ms;
        st.walk_kernel_launches++;
    }
    if (!consistent && nominal_ok) {
        // (b) opt-in: every walk owns a fixed slot of walk_length draws
        uint64_t tot2 = 0, n_changed = 0;
        rc = compute_offsets(g, d_starts, nullptr, walk_length, n_jobs, stream_skip, true, &tot2, &n_changed);
        if (rc) return rc;
        st.stream_addressing = 1;
        if (getenv("PW_DEBUG_ROUNDS")) fprintf(stderr, "[repair] %llu jobs re-addressed (nominal slots)\n", (unsigned long long)n_changed);
        if (n_changed) {
            st.repair_rounds++;
            wa.job_list = g->changed.p;
            wa.n_list = n_changed;
            HIP_TRY(hipEventRecord(g->ev[2], g->stream));
            rc = launch_walks(g, wa, extend != 0, &redo_total);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(g->ev[3], g->stream));
            HIP_TRY(hipStreamSynchronize(g->stream));
            HIP_TRY(hipEventElapsedTime(&ms, g->ev[2], g->ev[3]));
            st.walk_kernel_ms += ms;
            st.walk_kernel_launches++;
        }
    } else if (!consistent) {
        // (c) BLOCK-WISE, exact.  Invariant: every job below `first` is final and off_first is the stream offset of job `first`.
        // A round addresses the WINDOW [first, first + blk) from off_first (three small launches over the window, not the
        // array), re-walks the jobs of the window whose offset differs from the one they were last walked with, and moves
        // `first` to the first of them -- that job has now been walked with its final offset, everything before it was
        // consistent -- or to the end of a window in which nothing differed.  At least one job per round, O(window) work per
        // round; the window grows while rounds clear most of it and shrinks while they do not.  A time budget bounds the
        // whole (PECANPY_AMD_REPAIR_SECONDS, default 600; 0: none): the error names the alternatives.
        uint64_t blk = 4096;
        const char *blk_env = getenv("PECANPY_AMD_REPAIR_BLOCK");
        if (blk_env) blk = (uint64_t)strtoull(blk_env, nullptr, 10);
        if (blk < 1) blk = 1;
        const double budget_s = getenv("PECANPY_AMD_REPAIR_SECONDS") ? atof(getenv("PECANPY_AMD_REPAIR_SECONDS")) : 600.0;
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_begin = now();
        if (first == ~0ull || first >= n_jobs) first = 0;
        uint64_t off_first = stream_skip;
        if (first) HIP_TRY(hipMemcpy(&off_first, g->stream_off.p + first, sizeof(uint64_t), hipMemcpyDeviceToHost));   // (last whole-array scan)
        while (first < n_jobs) {
            const uint64_t n_win = first + blk < n_jobs ? blk : n_jobs - first;
            uint64_t tot2 = 0, n_changed = 0, fm = ~0ull;
            rc = compute_offsets(g, d_starts, d_out, walk_length, n_win, off_first, true, &tot2, &n_changed, first, n_jobs, &fm);
            if (rc) return rc;
            if (getenv("PW_DEBUG_ROUNDS")) fprintf(stderr, "[repair] round %llu (block-wise): window [%llu, +%llu): %llu jobs re-addressed\n",
                                                   (unsigned long long)st.repair_rounds, (unsigned long long)first, (unsigned long long)n_win, (unsigned long long)n_changed);
            if (n_changed == 0) {           // the whole window is consistent: final
                first += n_win;
                off_first += tot2;
                if (!blk_env && blk < (1ull << 20)) blk *= 2;
                continue;
            }
            st.repair_rounds++;
            wa.job_list = g->changed.p;
            wa.n_list = n_changed;
            HIP_TRY(hipEventRecord(g->ev[2], g->stream));
            rc = launch_walks(g, wa, extend != 0, &redo_total);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(g->ev[3], g->stream));
            HIP_TRY(hipMemcpyAsync(&off_first, g->stream_off.p + fm, sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
            HIP_TRY(hipStreamSynchronize(g->stream));
            HIP_TRY(hipEventElapsedTime(&ms, g->ev[2], g->ev[3]));
            st.walk_kernel_ms += ms;
            st.walk_kernel_launches++;
            if (!blk_env) {
                const uint64_t adv = fm - first;
                if (adv >= n_win / 2 && blk < (1ull << 20)) blk *= 2;
                else if (adv < n_win / 16 && blk > 256) blk /= 2;
            }
            first = fm;                     // (walked with its final offset just now; the next window starts on it)
            if (budget_s > 0 && now() - t_begin > budget_s)
                return fail(PW_ERR_UNSUPPORTED, "dead-end repair: " + std::to_string(first) + " of " + std::to_string(n_jobs) + " walks final after " +
                                                std::to_string((long long)budget_s) + " s of block-wise re-addressing (a directed graph full of sinks makes the "
                                                "reference's single random stream sequential, pecanpy.py:198-206).  PECANPY_AMD_REPAIR_SECONDS=0 lifts the "
                                                "limit; PECANPY_AMD_NOMINAL_STREAM=1 gives every walk a fixed slot of the stream instead (seeded, not the "
                                                "reference's assignment)");
        }
    }
    if (st.repair_rounds) {
        // statistics of the final, self-consistent matrix
        HIP_TRY(hipMemcpy(h, g->counters.p, sizeof(h), hipMemcpyDeviceToHost));
        uint64_t tot2 = 0;
        if (!st.stream_addressing) {
            rc = compute_offsets(g, d_starts, d_out, walk_length, n_jobs, stream_skip, false, &tot2, nullptr);
            if (rc) return rc;
        } else {  // offsets stay nominal; count the steps from the lengths without touching them
            std::vector<uint32_t> lens(n_jobs);
            HIP_TRY(hipMemcpy2D(lens.data(), sizeof(uint32_t), d_out + walk_length + 1,
                                sizeof(uint32_t) * ((size_t)walk_length + 2), sizeof(uint32_t), n_jobs,
                                hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < n_jobs; i++) tot2 += lens[i] - 1;
        }
        st.total_steps = tot2;
    } else {
        st.total_steps = h[1];
    }
    st.overflow_reads = h[2];
    st.clamped_reads = h[3];
    st.dead_end_walks = dead;
    st.lane_kernel = g->wl_used ? 3u : (lanes ? (lanes_float_eligible(g, wa) ? 2u : 1u) : 0u);
    st.redo_walks = redo_total;
    st.list_entries_read = h[7];
    st.ambiguous_steps = h[8];
    st.wave_chain_steps = h[10];
    st.param_index_ms = g->param_ms_call;
    st.lane_kernel_ms = g->lane_ms;
    st.lane_rounds = g->lane_rounds;
    st.verify_checked = g->ver_checked;
    st.verify_mismatch = g->ver_mismatch;
    st.verify_dropped = g->ver_dropped;
    st.verify_ties = g->ver_ties;
    st.eager_steps = h[12];
    st.index_max_list = g->kind == 0 && g->d_lines ? g->list_max_len : 0u;
    if (stats) *stats = st;
    return PW_OK;
}

// ---- TWIN contexts: the two halves of a weighted job array side by side on one GPU (round 6) --------------------------------
// The weighted lane form alternates two kernels that leave most of the GPU idle in turn: a lane round (one lane per walk) and
// the eager kernel that decides what the round parked (one wavefront per record, bound by the latency of its row scans) --
// 51 % + 48 % of a C5 pass.  Walked as two independent halves on two call contexts (own streams, counters and queues; graph,
// index and per-(p, q) tables shared), one half's eager kernel runs beside the other half's lane round: C5 417 -> ~330 ms per
// pass with two plain replicas (tools/replica_bench.py), without a second copy of anything here.  The walks are those of one
// call: the second half is addressed into the stream by the draws of the first (walked again if dead ends made them fewer).
static pw_graph *make_twin(pw_graph *g) {
    pw_graph *t = new pw_graph();
    if (graph_common_init(t, g->device)) { pw_graph_destroy(t); return nullptr; }
    t->alias = true;
    return t;
}
// everything the twin reads of the graph: pointers and the keys that say which (p, q) the tables were built for
static void twin_share(pw_graph *t, const pw_graph *g) {
    t->kind = g->kind; t->n_nodes = g->n_nodes; t->nnz = g->nnz; t->unit = g->unit; t->max_degree = g->max_degree;
    t->d_indptr = g->d_indptr; t->d_indices = g->d_indices; t->d_hasnbr = g->d_hasnbr; t->d_data = g->d_data; t->d_thr = g->d_thr;
    t->d_adjbits = g->d_adjbits; t->d_deg = g->d_deg; t->bits_only = g->bits_only; t->d_foff = g->d_foff; t->d_fbits = g->d_fbits;
    t->d_kf = g->d_kf; t->d_tab_off = g->d_tab_off; t->d_slots = g->d_slots; t->d_vrec = g->d_vrec; t->d_lines = g->d_lines;
    t->d_clist = g->d_clist; t->clist_bytes = g->clist_bytes; t->line_bytes = g->line_bytes; t->vlines = g->vlines;
    t->has_loop = g->has_loop; t->lanes_off = g->lanes_off; t->n_clist = g->n_clist; t->list_max_len = g->list_max_len;
    t->words_per_row = g->words_per_row;
    t->d_tot_e = g->d_tot_e; t->d_tot_v = g->d_tot_v; t->tot_p = g->tot_p; t->tot_q = g->tot_q; t->tot_extend = g->tot_extend;
    t->tot_thr_version = g->tot_thr_version; t->thr_version = g->thr_version; t->tot_failed = g->tot_failed;
    t->d_utot = g->d_utot; t->utot_wo = g->utot_wo; t->utot_wp = g->utot_wp; t->utot_failed = g->utot_failed;
    t->d_wb = g->d_wb; t->d_wpq = g->d_wpq; t->d_wdl = g->d_wdl; t->d_wl_dprev = g->d_wl_dprev; t->d_wl_off = g->d_wl_off;
    t->d_wedge_row = g->d_wedge_row; t->d_wp1 = g->d_wp1; t->d_wck_off = g->d_wck_off; t->d_wck = g->d_wck; t->wdl_cap = g->wdl_cap;
    t->wck_cap = g->wck_cap; t->wl_p = g->wl_p; t->wl_q = g->wl_q; t->wl_extend = g->wl_extend; t->wl_thr_version = g->wl_thr_version;
    t->wl_failed = g->wl_failed;
}

static void add_stats(pw_stats &total, const pw_stats &st, bool side_by_side) {
    total.total_steps += st.total_steps; total.overflow_reads += st.overflow_reads; total.clamped_reads += st.clamped_reads;
    total.dead_end_walks += st.dead_end_walks; total.repair_rounds += st.repair_rounds;
    if (side_by_side) {
        total.walk_kernel_ms = std::max(total.walk_kernel_ms, st.walk_kernel_ms); total.rng_kernel_ms = std::max(total.rng_kernel_ms, st.rng_kernel_ms);
        total.lane_kernel_ms = std::max(total.lane_kernel_ms, st.lane_kernel_ms); total.lane_rounds = std::max(total.lane_rounds, st.lane_rounds);
        total.param_index_ms = std::max(total.param_index_ms, st.param_index_ms);
    } else {
        total.walk_kernel_ms += st.walk_kernel_ms; total.rng_kernel_ms += st.rng_kernel_ms; total.lane_kernel_ms += st.lane_kernel_ms;
        total.lane_rounds += st.lane_rounds; total.param_index_ms += st.param_index_ms;
    }
    total.walk_kernel_launches += st.walk_kernel_launches; total.stream_addressing |= st.stream_addressing;
    total.redo_walks += st.redo_walks; total.list_entries_read += st.list_entries_read; total.ambiguous_steps += st.ambiguous_steps;
    total.wave_chain_steps += st.wave_chain_steps; total.verify_checked += st.verify_checked; total.verify_mismatch += st.verify_mismatch;
    total.verify_dropped += st.verify_dropped; total.verify_ties += st.verify_ties; total.eager_steps += st.eager_steps;
}

static int simulate_twin(pw_graph *g, int mode, double p, double q, int extend, const uint32_t *d_starts, uint64_t n_jobs,
                         uint32_t walk_length, uint32_t seed, uint64_t stream_skip, uint32_t *d_out, pw_stats *stats) {
    if (set_device(g)) return PW_ERR_HIP;
    if (!g->twin) g->twin = make_twin(g);
    pw_graph *t = g->twin;
    if (!t) return simulate_device_impl(g, mode, p, q, extend, d_starts, n_jobs, walk_length, 1, seed, stream_skip, d_out, stats);
    if (g->counters.ensure(N_COUNTERS)) return PW_ERR_NOMEM;
    { int rcs = check_starts(g, d_starts, n_jobs); if (rcs) return rcs; }
    const uint64_t half = n_jobs / 2;
    const size_t W = (size_t)walk_length + 2;
    uint64_t nominal_a = 0;
    int rc = compute_offsets(g, d_starts, nullptr, walk_length, half, stream_skip, false, &nominal_a, nullptr);
    if (rc) return rc;
    pw_stats sa, sb;
    memset(&sa, 0, sizeof(sa));
    memset(&sb, 0, sizeof(sb));
    std::promise<bool> ready;
    std::future<bool> ready_f = ready.get_future();
    bool signalled = false;
    g->on_tables_ready = [&]() { twin_share(t, g); signalled = true; ready.set_value(true); };
    int rc_b = 0;
    std::string err_b;
    std::thread tb;
    auto run_b = [&](uint64_t skip_b) {
        (void)hipSetDevice(t->device);
        rc_b = simulate_device_impl(t, mode, p, q, extend, d_starts + half, n_jobs - half, walk_length, 1, seed, skip_b, d_out + half * W, &sb);
        if (rc_b) err_b = g_err;
    };
    bool threaded = true;
    try {
        tb = std::thread([&]() { if (ready_f.get()) run_b(stream_skip + nominal_a); });
    } catch (const std::system_error &) { threaded = false; }
    g->twin_active = t->twin_active = threaded;
    struct ActiveGuard { pw_graph *a, *b; ~ActiveGuard() { a->twin_active = b->twin_active = false; } } active_guard{g, t};
    rc = simulate_device_impl(g, mode, p, q, extend, d_starts, half, walk_length, 1, seed, stream_skip, d_out, &sa);
    g->on_tables_ready = nullptr;
    if (!signalled) ready.set_value(false);     // (the first half failed before its tables were in place: the second is not walked)
    if (threaded) tb.join();
    else if (!rc) { twin_share(t, g); run_b(stream_skip + nominal_a); }   // (no thread to be had: one half after the other)
    if (rc) return rc;
    if (rc_b) return fail(rc_b, err_b);
    if (sa.total_steps != nominal_a && !sa.stream_addressing) {   // dead ends in the first half: the second starts earlier in the stream
        twin_share(t, g);
        run_b(stream_skip + sa.total_steps);
        if (rc_b) return fail(rc_b, err_b);
    }
    if (stats) { *stats = sa; add_stats(*stats, sb, true); }
    return PW_OK;
}

PW_EXPORT int pw_simulate_device(pw_graph *g, int mode, double p, double q, int extend,
                                 const uint32_t *d_starts, uint64_t n_jobs, uint32_t walk_length,
                                 int has_seed, uint32_t seed, uint64_t stream_skip, uint32_t *d_out,
                                 pw_stats *stats) {
    // weighted CSR graphs on the lane index, whole job arrays of a million walks or more: two halves side by side (above)
    if (g && !g->alias && g->kind == 0 && !g->unit && g->d_lines && !g->lanes_off && mode == PW_MODE_SPARSE_OTF && n_jobs >= (1ull << 20) &&
        d_starts && d_out && walk_length >= 1 && p > 0 && q > 0 && (!extend || g->d_thr) && !getenv("PECANPY_AMD_NO_TWIN") &&
        !getenv("PECANPY_AMD_NO_LANES") && !getenv("PECANPY_AMD_NO_WLANES") && !getenv("PECANPY_AMD_NO_CHAIN_QUEUE")) {
        if (!has_seed) seed = os_seed();
        return simulate_twin(g, mode, p, q, extend, d_starts, n_jobs, walk_length, seed, stream_skip, d_out, stats);
    }
    return simulate_device_impl(g, mode, p, q, extend, d_starts, n_jobs, walk_length, has_seed, seed, stream_skip, d_out, stats);
}

// Device -> pageable host copy through a ring of pinned staging buffers (a plain hipMemcpy to pageable memory runs at
// ~1/5 of the PCIe rate; the walk matrix is 13.8 GB at RMAT-22): the calling thread issues the chunk DMAs as buffers
// come free, a few worker threads each take the next chunk that has landed and copy it to its place (the destination
// is usually fresh pageable memory -- first-touch page faults set the pace of a thread, so several work side by side,
// each on a chunk of its own, with no barrier between chunks).  The buffers live in the handle.  `feed` (pw_simulate's
// parts): the bytes [0, feed->ready) of the source are final -- a chunk's DMA is issued once the producer has
// announced it, so ONE copy runs through all parts and the link never idles between them.
struct CopyFeed {
    std::mutex m;
    std::condition_variable cv;
    size_t ready = 0;
    bool abort = false;
    void announce(size_t r) { { std::lock_guard<std::mutex> l(m); ready = r; } cv.notify_all(); }
    void stop() { { std::lock_guard<std::mutex> l(m); abort = true; } cv.notify_all(); }
    // true once [0, upto) is final (block: wait for it); false: not yet / the producer gave up
    bool have(size_t upto, bool block) {
        std::unique_lock<std::mutex> l(m);
        if (block) cv.wait(l, [&]() { return abort || ready >= upto; });
        return !abort && ready >= upto;
    }
    bool stopped() { std::lock_guard<std::mutex> l(m); return abort; }
};

static int copy_out_staged(pw_graph *g, void *dst, const void *d_src, size_t bytes, CopyFeed *feed) {
    const size_t CH = (size_t)8 << 20;
    constexpr size_t NBUF = pw_graph::N_STAGE;
    bool pinned = bytes >= 2 * CH;
    if (pinned)
        for (auto &b : g->stage)
            if (!b && hipHostMalloc(&b, CH, hipHostMallocDefault) != hipSuccess) {
                b = nullptr;
                (void)hipGetLastError();
                pinned = false;   // no pinned memory: plain copy
            }
    if (!pinned) {
        if (feed && !feed->have(bytes, true)) return 0;   // (the producer failed: its error is the call's)
        HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, g->copy_stream));
        HIP_TRY(hipStreamSynchronize(g->copy_stream));
        return 0;
    }
    static const int T_env = getenv("PECANPY_AMD_COPY_THREADS") ? atoi(getenv("PECANPY_AMD_COPY_THREADS")) : 0;
    const int T_use = T_env > 0 ? (T_env > 32 ? 32 : T_env) : 8;
    const size_t n_ch = (bytes + CH - 1) / CH;
    const bool dbg = getenv("PECANPY_AMD_COPY_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto chunk_len = [&](size_t c) { return c * CH + CH < bytes ? CH : bytes - c * CH; };
    struct Ring {
        std::mutex m;
        std::condition_variable cv;
        size_t issued = 0, next = 0;        // chunks whose DMA + event are on the copy stream / handed to a worker
        size_t drained[NBUF] = {};          // buffer b: 1 + the last chunk copied out of it
        bool stop = false;
        hipError_t err = hipSuccess;
    } ring;
    auto worker = [&]() {
        (void)hipSetDevice(g->device);
        for (;;) {
            size_t c;
            {
                std::unique_lock<std::mutex> l(ring.m);
                c = ring.next;
                if (c >= n_ch || ring.stop) return;
                ring.next++;
                ring.cv.wait(l, [&]() { return ring.stop || ring.issued > c; });
                if (ring.stop) return;
            }
            const hipError_t e = hipEventSynchronize(g->ev_copy[c % NBUF]);
            if (e == hipSuccess) memcpy((char *)dst + c * CH, g->stage[c % NBUF], chunk_len(c));
            {
                std::lock_guard<std::mutex> l(ring.m);
                if (e != hipSuccess) { ring.err = e; ring.stop = true; }
                ring.drained[c % NBUF] = c + 1;
            }
            ring.cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 0; t < T_use; t++) pool.emplace_back(worker);
    } catch (const std::exception &) {   // (thread limit of the process: go on with the workers there are, or give up)
        if (pool.empty()) return fail(PW_ERR_NOMEM, "no thread for the copy out");
    }
    double t_feed = 0, t_ring = 0;
    hipError_t issue_err = hipSuccess;
    bool gave_up = false;
    for (size_t c = 0; c < n_ch; c++) {
        const double t0 = now();
        if (feed && !feed->have(c * CH + chunk_len(c), true)) { gave_up = true; break; }   // (the producer failed: its error is the call's)
        const double t1 = now();
        {   // the buffer's previous chunk has been copied out
            std::unique_lock<std::mutex> l(ring.m);
            ring.cv.wait(l, [&]() { return ring.stop || c < NBUF || ring.drained[c % NBUF] == c - NBUF + 1; });
            if (ring.stop) break;
        }
        t_feed += t1 - t0;
        t_ring += now() - t1;
        issue_err = hipMemcpyAsync(g->stage[c % NBUF], (const char *)d_src + c * CH, chunk_len(c), hipMemcpyDeviceToHost, g->copy_stream);
        if (issue_err == hipSuccess) issue_err = hipEventRecord(g->ev_copy[c % NBUF], g->copy_stream);
        if (issue_err != hipSuccess) break;
        { std::lock_guard<std::mutex> l(ring.m); ring.issued = c + 1; }
        ring.cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> l(ring.m);
        if (gave_up || issue_err != hipSuccess) ring.stop = true;   // (workers still waiting for a chunk that will not come)
    }
    ring.cv.notify_all();
    for (auto &th : pool) th.join();
    if (issue_err != hipSuccess) return fail(PW_ERR_HIP, hipGetErrorString(issue_err));
    if (ring.err != hipSuccess) return fail(PW_ERR_HIP, hipGetErrorString(ring.err));
    if (dbg) fprintf(stderr, "[copy_out] %.2f GB in %zu chunks: issuer waited %.1f ms for the walks, %.1f ms for a free buffer (%d copy threads)\n",
                     bytes / 1e9, n_ch, t_feed * 1e3, t_ring * 1e3, T_use);
    return 0;
}

PW_EXPORT int pw_simulate(pw_graph *g, int mode, double p, double q, int extend, const uint32_t *starts,
                          uint64_t n_jobs, uint32_t walk_length, int has_seed, uint32_t seed,
                          uint64_t stream_skip, uint32_t *out, pw_stats *stats) {
    if (!g || (n_jobs && (!starts || !out))) return fail(PW_ERR_INVALID, "null pointer");
    if (set_device(g)) return PW_ERR_HIP;
    uint32_t *d_starts = nullptr, *d_out = nullptr;
    const bool dbg = getenv("PECANPY_AMD_COPY_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // The matrix leaves the device at the PCIe rate (RMAT-18: 17 ms for 0.86 GB against 13 ms of walks): large job arrays
    // are walked in PARTS and the copy of part k (helper thread, copy stream, pinned staging) runs under the kernels of
    // part k + 1 (the copy is the longer of the two: the link stays busy throughout).  Part k + 1's
    // stream address = draws the earlier parts ACTUALLY consumed; every part's addressing is exact (the block-wise repair of
    // pw_simulate_device), so the walks are those of one call whatever the split (dead ends included).  Alias modes consume a variable number of words per step: one part.
    const size_t W = (size_t)walk_length + 2;
    const size_t row_bytes = sizeof(uint32_t) * W;
    int n_parts = 1;
    // (the first part's walks are the only ones the copy does not hide: more parts while a part stays a launch worth making --
    //  RMAT-18, 0.86 GB: 4 / 8 / 16 parts -> 22.7 / 21.5 / 29.0 ms per call)
    if (mode < PW_MODE_PRECOMP && (size_t)n_jobs * row_bytes >= ((size_t)128 << 20) && !getenv("PECANPY_AMD_NO_PARTS"))
        n_parts = (size_t)n_jobs * row_bytes >= ((size_t)800 << 20) ? 8 : 4;
    if (const char *pe = getenv("PECANPY_AMD_PARTS")) { n_parts = atoi(pe); if (n_parts < 1 || mode >= PW_MODE_PRECOMP) n_parts = 1; }
    // (nominal addressing -- the opt-in fallback of the dead-end repair -- must be decided ONCE for the whole array: a part that
    //  fell back would own slots up to skip + nominal while the next part started at skip + actual; exact addressing, the
    //  default, makes the walks those of one call whatever the split)
    if (env_on("PECANPY_AMD_NOMINAL_STREAM")) n_parts = 1;
    if ((uint64_t)n_parts > n_jobs) n_parts = 1;
    if (!has_seed && n_parts > 1) { seed = os_seed(); has_seed = 1; }   // (every part walks the same stream)
    // RING (round 6): from 8 parts on the device never holds the whole matrix -- three part-sized buffers take the parts in turn
    // (part k walks into buffer k % 3 once part k - 3 has left it) and every part expands its own stretch of the stream: an RMAT-22
    // call touches 5.2 + 1.6 GB of device memory instead of 13.8 + 12.9 GB.  On a fresh box, where memory nobody has used costs
    // ~23 ms per GB of hipMalloc, that is the difference between 690 and ~450 ms for a call whose floor is 250 ms of PCIe; the
    // per-part jump-ahead trees (~3 ms each) run under the copy, which is the longer leg.  PECANPY_AMD_NO_RING=1: whole matrix.
    const bool ring = n_parts >= 8 && !getenv("PECANPY_AMD_NO_RING");
    constexpr int R = 3;
    const uint64_t part_rows = (n_jobs + (uint64_t)n_parts - 1) / (uint64_t)n_parts + 1;
    const size_t out_elems = ring ? (size_t)R * part_rows * W : (size_t)n_jobs * W;
    HIP_TRY(hipMalloc((void **)&d_starts, sizeof(uint32_t) * (n_jobs ? n_jobs : 1)));
    hipError_t e = hipMalloc((void **)&d_out, sizeof(uint32_t) * (out_elems ? out_elems : 1));
    if (e != hipSuccess) { (void)hipFree(d_starts); return fail(PW_ERR_NOMEM, hipGetErrorString(e)); }
    int rc = 0;
    const double t1 = now();
    e = hipMemcpy(d_starts, starts, sizeof(uint32_t) * n_jobs, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(PW_ERR_HIP, hipGetErrorString(e));
    const double t2 = now();
    pw_stats total;
    memset(&total, 0, sizeof(total));
    CopyFeed feed, drained;            // feed: parts walked (ring) / bytes final (whole matrix); drained: parts copied out (ring)
    std::thread copier;
    bool copy_inline = false;
    int copy_rc = 0;
    std::string copy_err;
    uint64_t skip = stream_skip;
    auto part_lo = [&](int part) { return (uint64_t)part * n_jobs / (uint64_t)n_parts; };
    // whole-matrix form: the draws of the whole array, expanded once (a part's range -- offset by what the earlier parts
    // consumed, never more than their nominal share -- lies inside): one jump tree instead of one per part
    struct HoldGuard {
        pw_graph *g;
        bool mine;
        ~HoldGuard() { if (mine) g->rng_hold.valid = false; }
    } hold_guard{g, false};
    if (!rc && n_parts > 1 && !ring) {
        g->rng_hold.valid = false;        // (a caller's pw_stream_hold ends here: this call expands the stream of its own array)
        g->rng_hold.user = false;
        hold_guard.mine = true;
        if (g->counters.ensure(N_COUNTERS)) rc = PW_ERR_NOMEM;
        if (!rc) rc = check_starts(g, d_starts, n_jobs);
        uint64_t nominal = 0, base = 0;
        if (!rc) rc = compute_offsets(g, d_starts, nullptr, walk_length, n_jobs, stream_skip, false, &nominal, nullptr);
        if (!rc) (void)hipEventRecord(g->ev[7], g->stream);
        if (!rc) rc = expand_stream(g, seed, true, stream_skip, nominal, &base);
        if (!rc) {
            (void)hipEventRecord(g->ev[6], g->stream);   // (ev[0] / ev[1] are recorded again by every part)
            g->rng_hold.valid = true;
            g->rng_hold.seed = seed;
            g->rng_hold.first_block = base / 312;
            g->rng_hold.n_blocks = (stream_skip + nominal + 311) / 312 > base / 312 ? (stream_skip + nominal + 311) / 312 - base / 312 : 1;
        }
    }
    const bool held = g->rng_hold.valid && hold_guard.mine;
    auto copy_job = [&]() {
        (void)hipSetDevice(g->device);
        if (!ring) {
            copy_rc = copy_out_staged(g, out, d_out, (size_t)n_jobs * row_bytes, &feed);
        } else {
            for (int part = 0; part < n_parts && !copy_rc; part++) {
                if (!feed.have((size_t)part + 1, true)) break;             // (the walker gave up: its error is the call's)
                const uint64_t lo = part_lo(part), hi = part_lo(part + 1);
                copy_rc = copy_out_staged(g, out + lo * W, d_out + (size_t)(part % R) * part_rows * W, (size_t)(hi - lo) * row_bytes, nullptr);
                if (!copy_rc) drained.announce((size_t)part + 1);
            }
        }
        if (copy_rc) { copy_err = g_err; drained.stop(); }     // (g_err is thread local: carried over below)
    };
    for (int part = 0; part < n_parts && !rc; part++) {
        const uint64_t lo = part_lo(part), hi = part_lo(part + 1);
        if (ring && part >= R && !copy_inline && !drained.have((size_t)(part - R) + 1, true)) break;   // (the copy failed: its error is the call's)
        pw_stats st;
        memset(&st, 0, sizeof(st));
        uint32_t *dst = ring ? d_out + (size_t)(part % R) * part_rows * W : d_out + lo * W;
        rc = pw_simulate_device(g, mode, p, q, extend, d_starts + lo, hi - lo, walk_length, has_seed, seed, skip, dst, &st);
        if (rc) break;
        skip += st.total_steps;
        if (part == 0) total = st;
        else add_stats(total, st, false);
        if (!copier.joinable() && !copy_inline) {      // (ONE copy thread for all parts: it follows what is announced below)
            try {
                copier = std::thread(copy_job);
            } catch (const std::system_error &) {      // (thread limit of the process: the copy runs on this thread)
                copy_inline = true;
            }
        }
        feed.announce(ring ? (size_t)part + 1 : (size_t)hi * row_bytes);
        if (ring && copy_inline) {                     // (no thread: every part is copied out before the next one is walked)
            const int c = copy_out_staged(g, out + lo * W, dst, (size_t)(hi - lo) * row_bytes, nullptr);
            if (c) rc = c;
        }
    }
    if (rc) feed.stop();
    const double t3 = now();
    if (copier.joinable()) copier.join();
    else if (!rc && copy_inline && !ring) {
        copy_rc = copy_out_staged(g, out, d_out, (size_t)n_jobs * row_bytes, &feed);
        if (copy_rc) copy_err = g_err;
    }
    if (!rc && copy_rc) rc = fail(copy_rc, copy_err);
    if (!rc && held) {   // (the parts found their draws in place: the one expansion is the call's generator time)
        float ms = 0;
        if (hipEventElapsedTime(&ms, g->ev[7], g->ev[6]) == hipSuccess) total.rng_kernel_ms += ms;
    }
    if (!rc && stats) *stats = total;
    const double t4 = now();
    (void)hipFree(d_starts);
    (void)hipFree(d_out);
    if (dbg) fprintf(stderr, "[pw_simulate] alloc %.1f ms, starts in %.1f, walks %.1f, matrix out %.1f, free %.1f (%d parts%s)\n", (t1 - t0) * 1e3,
                     (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (now() - t4) * 1e3, n_parts, ring ? ", ring of 3 buffers" : "");
    return rc;
}

// ---- the walk operator over SEVERAL handles (replicas of one graph, pw_csr_create_multi), one host thread per handle -----------
// The job array is split into contiguous shards (the walks are independent, the graph is replicated);
// shard i is addressed into the ONE random stream by the draws the earlier shards consume -- announced by
// pw_count_stream_draws' rule (exact on undirected graphs) and, when dead ends made a shard consume fewer (directed graphs),
// corrected by walking the later shards again with the draws actually consumed: the matrix is that of one pw_simulate call
// whatever the number of handles.  out: host memory (out_on_device = 0: every device copies its rows out itself, one PCIe
// link each) or device memory of handles[0]'s GPU (1: the other devices' rows land there by peer copies over xGMI).
PW_EXPORT int pw_simulate_multi(pw_graph *const *handles, int n_handles, int mode, double p, double q, int extend,
                                const uint32_t *starts, uint64_t n_jobs, uint32_t walk_length, int has_seed, uint32_t seed,
                                uint64_t stream_skip, uint32_t *out, int out_on_device, pw_stats *stats) {
    if (!handles || n_handles < 1 || (n_jobs && (!starts || !out))) return fail(PW_ERR_INVALID, "null pointer / no handle");
    for (int i = 0; i < n_handles; i++)
        if (!handles[i]) return fail(PW_ERR_INVALID, "null handle");
    for (int i = 1; i < n_handles; i++)
        if (handles[i]->n_nodes != handles[0]->n_nodes || handles[i]->nnz != handles[0]->nnz || handles[i]->kind != handles[0]->kind)
            return fail(PW_ERR_INVALID, "pw_simulate_multi: the handles are not replicas of one graph");
    if (!has_seed) { seed = os_seed(); has_seed = 1; }   // (every shard walks the same stream)
    const size_t W = (size_t)walk_length + 2;
    int n_sh = n_handles;
    if (mode >= PW_MODE_PRECOMP || n_jobs < (uint64_t)n_handles * 64) n_sh = 1;   // (alias modes: a sequential stream; tiny arrays: one device)
    if (n_sh == 1 && !out_on_device) return pw_simulate(handles[0], mode, p, q, extend, starts, n_jobs, walk_length, has_seed, seed, stream_skip, out, stats);
    struct Shard {
        uint64_t lo = 0, hi = 0, skip = 0, nominal = 0;
        uint32_t *d_starts = nullptr, *d_tmp = nullptr;
        pw_stats st;
        int rc = 0;
        std::string err;
        bool final_ = false;
    };
    std::vector<Shard> sh((size_t)n_sh);
    for (int i = 0; i < n_sh; i++) {
        sh[i].lo = (uint64_t)i * n_jobs / n_sh;
        sh[i].hi = (uint64_t)(i + 1) * n_jobs / n_sh;
        memset(&sh[i].st, 0, sizeof(pw_stats));
    }
    auto for_shards = [&](const std::function<void(int)> &fn, int first) {
        std::vector<std::thread> th;
        for (int i = first + 1; i < n_sh; i++) {
            try { th.emplace_back(fn, i); } catch (const std::system_error &) { fn(i); }
        }
        fn(first);
        for (auto &t : th) t.join();
    };
    // 1. the draws every shard announces (device resident copies of the starts for the device-output form)
    for_shards([&](int i) {
        Shard &S = sh[i];
        pw_graph *g = handles[i];
        if (S.hi == S.lo) return;
        if (mode >= PW_MODE_PRECOMP) return;
        S.rc = pw_count_stream_draws(g, starts + S.lo, S.hi - S.lo, walk_length, &S.nominal);
        if (S.rc) S.err = g_err;
    }, 0);
    for (auto &S : sh) if (S.rc) return fail(S.rc, S.err);
    { uint64_t run = stream_skip; for (auto &S : sh) { S.skip = run; run += S.nominal; } }
    // 2. walk; shards whose address turns out wrong (an earlier shard consumed fewer draws: dead ends) walk again
    int first_open = 0;
    for (int round = 0; round <= n_sh && first_open < n_sh; round++) {
        for_shards([&](int i) {
            Shard &S = sh[i];
            pw_graph *g = handles[i];
            if (S.hi == S.lo) return;
            const uint64_t n = S.hi - S.lo;
            if (!out_on_device) {
                S.rc = pw_simulate(g, mode, p, q, extend, starts + S.lo, n, walk_length, 1, seed, S.skip, out + S.lo * W, &S.st);
                if (S.rc) S.err = g_err;
                return;
            }
            // device output on handles[0]'s GPU: walk into local memory, then one peer copy (in place when it IS that GPU)
            hipError_t e = hipSetDevice(g->device);
            // (PECANPY_AMD_MULTI_FORCE_PEER=1, tests on a one-GPU box: replicas on the first device take the peer-copy path too)
            const bool local = g->device == handles[0]->device && !(i > 0 && env_on("PECANPY_AMD_MULTI_FORCE_PEER"));
            if (e == hipSuccess && !S.d_starts) {
                e = hipMalloc((void **)&S.d_starts, sizeof(uint32_t) * n);
                if (e == hipSuccess) e = hipMemcpy(S.d_starts, starts + S.lo, sizeof(uint32_t) * n, hipMemcpyHostToDevice);
                if (e == hipSuccess && !local) e = hipMalloc((void **)&S.d_tmp, sizeof(uint32_t) * n * W);
            }
            if (e != hipSuccess) { S.rc = PW_ERR_HIP; S.err = std::string("pw_simulate_multi: ") + hipGetErrorString(e); return; }
            uint32_t *dst = local ? out + S.lo * W : S.d_tmp;
            // A shard on another GPU is walked in three chunks of decreasing size (3 : 2 : 1) and chunk c travels -- peer copy on
            // the handle's copy stream, one xGMI link -- while chunk c + 1 is walked: what stays exposed at the end is the
            // smallest chunk's transfer (the tapered chunks of bench.py's RCCL gather).  Chunk c + 1 is
            // addressed by the draws chunk c actually consumed.  PECANPY_AMD_MULTI_CHUNKS overrides (1: one piece).
            int n_ch = (!local && n >= (1ull << 20)) ? 3 : 1;
            if (const char *ce = getenv("PECANPY_AMD_MULTI_CHUNKS")) { n_ch = atoi(ce); if (n_ch < 1) n_ch = 1; if (n_ch > 16) n_ch = 16; }
            if ((uint64_t)n_ch > n) n_ch = 1;
            const uint64_t wsum = (uint64_t)n_ch * (n_ch + 1) / 2;
            uint64_t a = 0, acc = 0, skip_c = S.skip;
            pw_stats tot;
            memset(&tot, 0, sizeof(tot));
            if (n_ch > 1 && mode < PW_MODE_PRECOMP) {   // one jump-ahead tree for the shard, not one per chunk
                S.rc = pw_stream_hold(g, seed, S.skip, S.nominal);
                if (S.rc) { S.err = g_err; return; }
            }
            for (int c = 0; c < n_ch && !S.rc; c++) {
                acc += (uint64_t)(n_ch - c);
                const uint64_t b = c + 1 == n_ch ? n : acc * n / wsum;
                if (b == a) continue;
                pw_stats st;
                memset(&st, 0, sizeof(st));
                S.rc = pw_simulate_device(g, mode, p, q, extend, S.d_starts + a, b - a, walk_length, 1, seed, skip_c, dst + a * W, &st);
                if (S.rc) { S.err = g_err; break; }
                skip_c += st.total_steps;
                if (c == 0) tot = st;
                else {
                    tot.total_steps += st.total_steps; tot.overflow_reads += st.overflow_reads; tot.clamped_reads += st.clamped_reads;
                    tot.dead_end_walks += st.dead_end_walks; tot.repair_rounds += st.repair_rounds; tot.walk_kernel_ms += st.walk_kernel_ms;
                    tot.rng_kernel_ms += st.rng_kernel_ms; tot.walk_kernel_launches += st.walk_kernel_launches;
                    tot.stream_addressing |= st.stream_addressing; tot.lane_rounds += st.lane_rounds; tot.redo_walks += st.redo_walks;
                    tot.list_entries_read += st.list_entries_read; tot.ambiguous_steps += st.ambiguous_steps; tot.lane_kernel_ms += st.lane_kernel_ms;
                    tot.wave_chain_steps += st.wave_chain_steps; tot.param_index_ms += st.param_index_ms; tot.verify_checked += st.verify_checked;
                    tot.verify_mismatch += st.verify_mismatch; tot.verify_dropped += st.verify_dropped; tot.verify_ties += st.verify_ties;
                    tot.eager_steps += st.eager_steps;
                }
                if (!local) {   // (the walks of this chunk are complete: pw_simulate_device returns after its stream has drained)
                    e = hipMemcpyPeerAsync(out + (S.lo + a) * W, handles[0]->device, S.d_tmp + a * W, g->device, sizeof(uint32_t) * (b - a) * W, g->copy_stream);
                    if (e != hipSuccess) { S.rc = PW_ERR_HIP; S.err = std::string("pw_simulate_multi (peer copy): ") + hipGetErrorString(e); }
                }
                a = b;
            }
            S.st = tot;
            (void)pw_stream_release(g);
            if (!local) {
                e = hipStreamSynchronize(g->copy_stream);
                if (e != hipSuccess && !S.rc) { S.rc = PW_ERR_HIP; S.err = std::string("pw_simulate_multi (peer copy): ") + hipGetErrorString(e); }
            }
        }, first_open);
        int bad = -1;
        for (int i = first_open; i < n_sh; i++) if (sh[i].rc) { bad = i; break; }
        if (bad >= 0) break;
        // shard i is final once every earlier shard is and its address equals what they consumed
        uint64_t run = sh[first_open].skip;
        int i = first_open;
        for (; i < n_sh; i++) {
            if (sh[i].skip != run) break;
            sh[i].final_ = true;
            run += sh[i].st.total_steps;
        }
        first_open = i;
        for (int j = i; j < n_sh; j++) { sh[j].skip = run; run += sh[j].st.total_steps; }   // (their own counts as the next estimate)
    }
    int rc = 0;
    for (auto &S : sh) {
        if (S.rc && !rc) rc = fail(S.rc, S.err);
    }
    for (int i = 0; i < n_sh; i++) {
        if (sh[i].d_starts || sh[i].d_tmp) {
            (void)hipSetDevice(handles[i]->device);
            if (sh[i].d_starts) (void)hipFree(sh[i].d_starts);
            if (sh[i].d_tmp) (void)hipFree(sh[i].d_tmp);
        }
    }
    if (!rc && first_open < n_sh) rc = fail(PW_ERR_HIP, "pw_simulate_multi: shard addressing did not settle");
    if (rc) return rc;
    if (stats) {
        pw_stats total = sh[0].st;
        for (int i = 1; i < n_sh; i++) {
            const pw_stats &st = sh[i].st;
            total.total_steps += st.total_steps; total.overflow_reads += st.overflow_reads; total.clamped_reads += st.clamped_reads;
            total.dead_end_walks += st.dead_end_walks; total.repair_rounds += st.repair_rounds;
            total.walk_kernel_ms = std::max(total.walk_kernel_ms, st.walk_kernel_ms);     // (the shards run side by side)
            total.rng_kernel_ms = std::max(total.rng_kernel_ms, st.rng_kernel_ms);
            total.lane_kernel_ms = std::max(total.lane_kernel_ms, st.lane_kernel_ms);
            total.walk_kernel_launches += st.walk_kernel_launches;
            total.stream_addressing |= st.stream_addressing; total.lane_rounds = std::max(total.lane_rounds, st.lane_rounds);
            total.redo_walks += st.redo_walks; total.list_entries_read += st.list_entries_read; total.ambiguous_steps += st.ambiguous_steps;
            total.wave_chain_steps += st.wave_chain_steps; total.param_index_ms = std::max(total.param_index_ms, st.param_index_ms);
            total.verify_checked += st.verify_checked; total.verify_mismatch += st.verify_mismatch; total.verify_dropped += st.verify_dropped;
            total.verify_ties += st.verify_ties; total.eager_steps += st.eager_steps;
        }
        *stats = total;
    }
    return PW_OK;
}

PW_EXPORT int pw_mt_random_sample(uint32_t seed, uint64_t offset, uint64_t n, double *out) {
    if (n && !out) return fail(PW_ERR_INVALID, "null pointer");
    pw::mt_random_sample_host(seed, offset, n, out);
    return PW_OK;
}

// ---- single-step probe (Base.get_move_forward of the drop-in API; probability vectors for the parity tests) --------
typedef void (*probe_kernel_fn)(pw::WalkArgs, const pw::ProbeArgs *);

static int run_probe(pw_graph *g, int mode, double p, double q, int extend, uint32_t cur, int has_prev, uint32_t prev, double r,
                     void *probs_host, uint32_t *out_host) {
    if (!g) return fail(PW_ERR_INVALID, "null pointer");
    if (mode == PW_MODE_SPARSE_OTF && g->kind != 0) return fail(PW_ERR_UNSUPPORTED, "SparseOTF needs a CSR graph handle");
    if (mode == PW_MODE_DENSE_OTF && g->kind != 1) return fail(PW_ERR_UNSUPPORTED, "DenseOTF needs a dense graph handle");
    if (mode != PW_MODE_SPARSE_OTF && mode != PW_MODE_DENSE_OTF) return fail(PW_ERR_UNSUPPORTED, "single steps are provided for the on-the-fly modes");
    if (g->bits_only) return fail(PW_ERR_UNSUPPORTED, "dense graph created from packed bits has no compressed rows");
    if (!(p > 0) || !(q > 0)) return fail(PW_ERR_INVALID, "p and q must be positive");
    if (cur >= g->n_nodes || (has_prev && prev >= g->n_nodes)) return fail(PW_ERR_INVALID, "vertex out of range");
    if (extend && !g->unit && !g->d_thr) return fail(PW_ERR_INVALID, "extend: call pw_graph_set_thresholds() first");
    if (set_device(g)) return PW_ERR_HIP;
    const bool ext = extend && !g->unit;
    const size_t elem = g->kind == 0 ? sizeof(float) : sizeof(double);
    pw::ProbeArgs *d_pa = nullptr;
    uint32_t *d_out = nullptr;
    void *d_probs = nullptr;
    auto cleanup = [&]() {
        if (d_pa) (void)hipFree(d_pa);
        if (d_out) (void)hipFree(d_out);
        if (d_probs) (void)hipFree(d_probs);
    };
    hipError_t e = hipMalloc((void **)&d_pa, sizeof(pw::ProbeArgs));
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, 4 * sizeof(uint32_t));
    if (e == hipSuccess && probs_host) e = hipMalloc(&d_probs, elem * ((size_t)g->max_degree + 1));
    if (e != hipSuccess) { cleanup(); return fail(PW_ERR_NOMEM, hipGetErrorString(e)); }
    pw::ProbeArgs pa;
    pa.cur = cur; pa.has_prev = has_prev ? 1u : 0u; pa.prev = has_prev ? prev : 0u; pa.want_probs = probs_host ? 1u : 0u;
    pa.r = r; pa.probs = d_probs; pa.out = d_out;
    pw::WalkArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.g = csr_dev(g);
    wa.p = p;
    wa.q = q;
    wa.L = 1;
    wa.w_out = (float)(1.0 / q);
    wa.w_prev = (float)(1.0 / p);
    probe_kernel_fn fn;
    if (g->kind == 0) fn = g->unit ? pw::step_probe_kernel<float, false, true, false>
                                    : (ext ? pw::step_probe_kernel<float, false, false, true> : pw::step_probe_kernel<float, false, false, false>);
    else fn = g->unit ? pw::step_probe_kernel<double, true, true, false>
                      : (ext ? pw::step_probe_kernel<double, true, false, true> : pw::step_probe_kernel<double, true, false, false>);
    uint32_t zero[4] = {0, 0, 0, 0};
    e = hipMemcpyAsync(d_pa, &pa, sizeof(pa), hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_out, zero, sizeof(zero), hipMemcpyHostToDevice, g->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fn, dim3(1), dim3(pw::WAVE), 0, g->stream, wa, (const pw::ProbeArgs *)d_pa);
        e = hipGetLastError();
    }
    uint32_t out[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    if (e == hipSuccess && probs_host && out[2]) e = hipMemcpy(probs_host, d_probs, elem * out[2], hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(PW_ERR_HIP, std::string("single-step probe: ") + hipGetErrorString(e));
    out_host[0] = out[0]; out_host[1] = out[1]; out_host[2] = out[2];
    return PW_OK;
}

PW_EXPORT int pw_step(pw_graph *g, int mode, double p, double q, int extend, uint32_t cur, int has_prev, uint32_t prev, double r,
                      uint32_t *next, uint32_t *position) {
    if (!next) return fail(PW_ERR_INVALID, "null pointer");
    uint32_t out[3];
    int rc = run_probe(g, mode, p, q, extend, cur, has_prev, prev, r, nullptr, out);
    if (rc) return rc;
    if (out[2] == 0) return fail(PW_ERR_INVALID, "vertex has no neighbours");
    *next = out[1];
    if (position) *position = out[0];
    return PW_OK;
}

PW_EXPORT int pw_probs(pw_graph *g, int mode, double p, double q, int extend, uint32_t cur, int has_prev, uint32_t prev, void *probs,
                       uint32_t *n) {
    if (!probs || !n) return fail(PW_ERR_INVALID, "null pointer");
    uint32_t out[3];
    int rc = run_probe(g, mode, p, q, extend, cur, has_prev, prev, 0.5, probs, out);
    if (rc) return rc;
    *n = out[2];
    return PW_OK;
}

// ---- skip-gram with negative sampling over a walk matrix (sgns.hip.h) -------------------------
PW_EXPORT int pw_sgns_train(int device, const uint32_t *walks, uint64_t n_walks, uint32_t walk_length, uint32_t n_nodes,
                            uint32_t dim, uint32_t window, uint32_t negative, uint32_t epochs, float alpha, float min_alpha,
                            float sample, uint32_t seed, uint32_t workers, float *vectors) {
    if (!walks || !vectors || !n_walks || !n_nodes) return fail(PW_ERR_INVALID, "null pointer / empty corpus");
    if (dim == 0 || dim > 64 * pw::SGNS_MAX_PER_LANE) return fail(PW_ERR_INVALID, "dim must be in 1..512");
    if (window == 0 || epochs == 0) return fail(PW_ERR_INVALID, "window and epochs must be positive");
    int ndev = pw_device_count();
    if (ndev <= 0) return fail(PW_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(PW_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    const uint32_t L = walk_length;
    const size_t wbytes = sizeof(uint32_t) * (size_t)n_walks * ((size_t)L + 2), vbytes = sizeof(float) * (size_t)n_nodes * dim;
    uint32_t *d_walks = nullptr, *d_table = nullptr;
    unsigned long long *d_cnt = nullptr;
    float *d_syn0 = nullptr, *d_syn1 = nullptr, *d_keep = nullptr;
    auto cleanup = [&]() {
        for (void *q : {(void *)d_walks, (void *)d_table, (void *)d_cnt, (void *)d_syn0, (void *)d_syn1, (void *)d_keep})
            if (q) (void)hipFree(q);
    };
    auto bail = [&](int code, const std::string &msg) { cleanup(); return fail(code, msg); };
    hipError_t e = hipMalloc((void **)&d_walks, wbytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, sizeof(unsigned long long) * ((size_t)n_nodes + 1));   // [n_nodes]: first bad walk
    if (e == hipSuccess) e = hipMalloc((void **)&d_syn0, vbytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_syn1, vbytes);
    if (e == hipSuccess) e = hipMemcpy(d_walks, walks, wbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, sizeof(unsigned long long) * (size_t)n_nodes);
    if (e == hipSuccess) e = hipMemset(d_cnt + n_nodes, 0xff, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(d_syn1, 0, vbytes);
    if (e != hipSuccess) return bail(PW_ERR_HIP, std::string("pw_sgns_train: ") + hipGetErrorString(e));
    // vocabulary statistics (and the check that the matrix only names nodes of the vocabulary)
    const uint64_t n_items = n_walks * (uint64_t)(L + 1);
    hipLaunchKernelGGL(pw::sgns_count_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, 0, d_walks, n_walks, L, n_nodes, d_cnt,
                       d_cnt + n_nodes);
    std::vector<unsigned long long> cnt((size_t)n_nodes + 1);
    e = hipMemcpy(cnt.data(), d_cnt, sizeof(unsigned long long) * ((size_t)n_nodes + 1), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return bail(PW_ERR_HIP, std::string("pw_sgns_train: ") + hipGetErrorString(e));
    if (cnt[n_nodes] != ~0ull)
        return bail(PW_ERR_INVALID, "walk " + std::to_string(cnt[n_nodes]) + ": node id >= n_nodes or length cell > walk_length + 1");
    double total = 0, pow_total = 0;
    for (uint32_t i = 0; i < n_nodes; i++) { total += (double)cnt[i]; pow_total += std::pow((double)cnt[i], 0.75); }
    if (!(total > 0)) return bail(PW_ERR_INVALID, "the walk matrix holds no nodes");
    // word2vec's unigram^0.75 table (a word that never occurs owns no slot) and subsampling probabilities
    const uint32_t table_size = (uint32_t)std::min<uint64_t>(1ull << 26, std::max<uint64_t>(1ull << 16, 16ull * n_nodes));
    std::vector<uint32_t> table(table_size);
    {
        uint32_t t = 0, last = 0;
        double cum = 0;
        for (uint32_t w = 0; w < n_nodes; w++) {
            if (!cnt[w]) continue;
            last = w;
            cum += std::pow((double)cnt[w], 0.75) / pow_total;
            while (t < table_size && (double)(t + 1) / table_size <= cum) table[t++] = w;
        }
        while (t < table_size) table[t++] = last;   // (rounding of the last share)
    }
    std::vector<float> keep;
    if (sample > 0) {
        keep.resize(n_nodes);
        const double thr = (double)sample * total;
        for (uint32_t i = 0; i < n_nodes; i++)
            keep[i] = cnt[i] ? (float)std::min(1.0, (std::sqrt((double)cnt[i] / thr) + 1.0) * thr / (double)cnt[i]) : 1.0f;
    }
    // syn0 ~ U(-0.5, 0.5) / dim (word2vec.c), seeded
    std::vector<float> init((size_t)n_nodes * dim);
    {
        uint64_t x = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed << 17);
        for (auto &f : init) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            f = (((float)((x >> 40) & 0xffffff) / 16777216.0f) - 0.5f) / (float)dim;
        }
    }
    e = hipMalloc((void **)&d_table, sizeof(uint32_t) * (size_t)table_size);
    if (e == hipSuccess) e = hipMemcpy(d_table, table.data(), sizeof(uint32_t) * (size_t)table_size, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_syn0, init.data(), vbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && sample > 0) {
        e = hipMalloc((void **)&d_keep, sizeof(float) * (size_t)n_nodes);
        if (e == hipSuccess) e = hipMemcpy(d_keep, keep.data(), sizeof(float) * (size_t)n_nodes, hipMemcpyHostToDevice);
    }
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return bail(PW_ERR_HIP, std::string("pw_sgns_train: ") + hipGetErrorString(e));
    pw::SgnsArgs a;
    a.walks = d_walks; a.n_walks = n_walks; a.L = L; a.dim = dim; a.window = window; a.negative = negative;
    a.syn0 = d_syn0; a.syn1 = d_syn1; a.table = d_table; a.table_size = table_size; a.keep = d_keep;
    a.alpha = alpha; a.min_alpha = min_alpha; a.item_total = n_items * epochs; a.seed = seed;
    // concurrency.  workers == 1: ONE wavefront walks the corpus in sentence order (deterministic: the run gensim's
    // workers=1 corresponds to; compared with oracle/sgns_ref.c).  workers == 0: as many wavefronts as keep hogwild
    // collisions rare -- far more wavefronts than vocabulary rows in flight would overwrite most updates of a small
    // graph: at least ~256 items per wavefront.  workers > 1: that many wavefronts.
    unsigned blocks, threads = 256;
    if (workers == 1) { blocks = 1; threads = 64; }
    else if (workers > 1) blocks = (workers + 3) / 4;
    else {
        const uint64_t want_blocks = (n_items + 1023) / 1024;
        blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want_blocks, (uint64_t)prop.multiProcessorCount * 8));
    }
    for (uint32_t ep = 0; ep < epochs; ep++) {
        a.item_base = n_items * ep;
        hipLaunchKernelGGL(pw::sgns_kernel, dim3(blocks), dim3(threads), 0, 0, a);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(vectors, d_syn0, vbytes, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(PW_ERR_HIP, std::string("pw_sgns_train: ") + hipGetErrorString(e));
    return PW_OK;
}

// ---- host self test of the exact-arithmetic decision ---------------------------------------------------
PW_EXPORT int pw_selftest_exact_decision(const uint8_t *cls, uint32_t n, float w_out, float w_prev, const double *r,
                                         uint32_t n_r, uint32_t *chain, uint32_t *exact) {
    if (!cls || !r || !chain || !exact || n == 0) return fail(PW_ERR_INVALID, "bad argument");
    auto pow2 = [](float w) { int e = 0; return std::frexp(w, &e) == 0.5f; };
    if (!pow2(w_out) || !pow2(w_prev)) return fail(PW_ERR_UNSUPPORTED, "biases must be powers of two");
    uint32_t cnt[3] = {0, 0, 0};
    for (uint32_t k = 0; k < n; k++) {
        if (cls[k] > 2) return fail(PW_ERR_INVALID, "class must be 0 (out), 1 (common) or 2 (prev)");
        cnt[cls[k]]++;
    }
    if (cnt[2] > 1) return fail(PW_ERR_INVALID, "at most one prev");
    // same set-up as sample_step_unit_lazy (walk_sparse.hip.h)
    float u = 1.0f;
    if (cnt[0] && w_out < u) u = w_out;
    if (cnt[2] && w_prev < u) u = w_prev;
    const double td = (double)cnt[1] + (double)cnt[0] * (double)w_out + (double)cnt[2] * (double)w_prev;
    if (!(td <= 16777216.0 * (double)u)) return fail(PW_ERR_UNSUPPORTED, "row total not exact in float32");
    const float tot = (float)td;
    const float x_in = 1.0f / tot, x_out = x_in * w_out, x_prev = x_in * w_prev;
    const double units = td / (double)u;
    const uint32_t w_units[3] = {cnt[0] ? (uint32_t)(w_out / u) : 1u, (uint32_t)(1.0f / u), cnt[2] ? (uint32_t)(w_prev / u) : 1u};
    const uint32_t wmax = std::max(w_units[0], std::max(w_units[1], w_units[2]));
    std::vector<uint32_t> E(n);
    uint32_t acc = 0;
    for (uint32_t k = 0; k < n; k++) { acc += w_units[cls[k]]; E[k] = acc; }
    for (uint32_t i = 0; i < n_r; i++) {
        // the reference: np.searchsorted(np.cumsum(float32 values), r), sequential float32 additions
        float c = 0.0f;
        uint32_t kc = n;
        for (uint32_t k = 0; k < n; k++) {
            c = c + (cls[k] == 1 ? x_in : (cls[k] == 0 ? x_out : x_prev));
            if ((double)c >= r[i]) { kc = k; break; }
        }
        chain[i] = kc;
        const pw::ExactThresholds th = pw::exact_thresholds_f32(r[i] * units, n, wmax);
        const uint32_t k1 = (uint32_t)(std::lower_bound(E.begin(), E.end(), th.lo) - E.begin());
        exact[i] = (k1 < n && E[k1] >= th.hi) ? k1 : 0xffffffffu;
    }
    return PW_OK;
}

PW_EXPORT int pw_selftest_exact_decision_f64(const uint8_t *cls, uint32_t n, double w_out, double w_prev,
                                             const double *r, uint32_t n_r, uint32_t *chain, uint32_t *exact) {
    if (!cls || !r || !chain || !exact || n == 0) return fail(PW_ERR_INVALID, "bad argument");
    auto pow2 = [](double w) { int e = 0; return std::frexp(w, &e) == 0.5; };
    if (!pow2(w_out) || !pow2(w_prev)) return fail(PW_ERR_UNSUPPORTED, "biases must be powers of two");
    uint64_t cnt[3] = {0, 0, 0};
    for (uint32_t k = 0; k < n; k++) {
        if (cls[k] > 2) return fail(PW_ERR_INVALID, "class must be 0 (out), 1 (common) or 2 (prev)");
        cnt[cls[k]]++;
    }
    if (cnt[2] > 1) return fail(PW_ERR_INVALID, "at most one prev");
    // same set-up as walk_dense_bits_kernel (walk_dense.hip.h)
    double u = 1.0;
    if (cnt[0] && w_out < u) u = w_out;
    if (cnt[2] && w_prev < u) u = w_prev;
    const double tot = (double)cnt[1] + (double)cnt[0] * w_out + (double)cnt[2] * w_prev;
    const double S = tot / u, wi = 1.0 / u, wo = w_out / u, wp = w_prev / u;
    const double wmax = std::max(wi, std::max(wo, wp));
    if (!(S <= 1099511627776.0 && wmax + 2.0 <= 1048576.0)) return fail(PW_ERR_UNSUPPORTED, "row outside the exact range");
    const double x[3] = {w_out / tot, 1.0 / tot, w_prev / tot};
    const uint64_t w_units[3] = {(uint64_t)wo, (uint64_t)wi, (uint64_t)wp};
    std::vector<uint64_t> E(n);
    uint64_t acc = 0;
    for (uint32_t k = 0; k < n; k++) { acc += w_units[cls[k]]; E[k] = acc; }
    for (uint32_t i = 0; i < n_r; i++) {
        double c = 0.0;
        uint32_t kc = n;
        for (uint32_t k = 0; k < n; k++) {
            c = c + x[cls[k]];
            if (c >= r[i]) { kc = k; break; }
        }
        chain[i] = kc;
        const pw::ExactThresholds64 th = pw::exact_thresholds_f64(r[i] * S, (double)n, wmax);
        const uint32_t k1 = (uint32_t)(std::lower_bound(E.begin(), E.end(), th.lo) - E.begin());
        exact[i] = (k1 < n && E[k1] >= th.hi) ? k1 : 0xffffffffu;
    }
    return PW_OK;
}

// ---- self tests of the lane kernel's per-thread routines (seqscan.h: lane_decide, lane_tight, lane_chain) ----------
// One row given by its classes (0 out, 1 common, 2 prev), many targets.  The same routine runs on the host and -- in
// lane_selftest_kernel, one thread per target -- on the DEVICE, where the compiler's code generation and the hardware's
// arithmetic (v_rcp_f32 in lane_tight, -ffp-contract=off adds) are what the walk kernels actually execute.
namespace {

struct LaneRow {
    std::vector<uint16_t> cl16;
    std::vector<uint32_t> cl32;
    uint32_t n_cl = 0, pp = 0xffffffffu, wide = 0;
    uint32_t piv_off = 0;   // element offset of the list's pivots inside cl16 / cl32 (0: none) -- as the index build lays them out
    float tot = 0, x_in = 0, x_out = 0, x_prev = 0;
    pw::ListView view() const { return pw::list_view_of(wide ? (const void *)cl32.data() : (const void *)cl16.data(), wide, n_cl, piv_off); }
};

int lane_row_setup(const uint8_t *cls, uint32_t n, float w_out, float w_prev, LaneRow &row, bool need_pow2 = true) {
    auto pow2 = [](float w) { int e = 0; return std::frexp(w, &e) == 0.5f; };
    if (need_pow2 && (!pow2(w_out) || !pow2(w_prev))) return fail(PW_ERR_UNSUPPORTED, "biases must be powers of two");
    if (!(w_out > 0.0f) || !(w_prev > 0.0f)) return fail(PW_ERR_INVALID, "biases must be positive");
    uint32_t cnt[3] = {0, 0, 0};
    row.wide = n > 65536u ? 1u : 0u;   // positions of rows up to 65536 entries are uint16 (walk_lanes.hip.h)
    for (uint32_t k = 0; k < n; k++) {
        if (cls[k] > 2) return fail(PW_ERR_INVALID, "class must be 0 (out), 1 (common) or 2 (prev)");
        cnt[cls[k]]++;
        if (cls[k] == 1) { if (row.wide) row.cl32.push_back(k); else row.cl16.push_back((uint16_t)k); }
        if (cls[k] == 2) row.pp = k;
    }
    if (cnt[2] > 1) return fail(PW_ERR_INVALID, "at most one prev");
    row.n_cl = cnt[1];
    row.cl16.resize(row.cl16.size() + 8, 0xffffu);       // a list window may read past the end of the list
    row.cl32.resize(row.cl32.size() + 8, 0xffffffffu);
    if (pw::list_has_pivots(row.wide, row.n_cl) && !getenv("PW_SELFTEST_NO_PIVOTS")) {   // pivots, as eline_pivots_kernel writes them
        const uint32_t np = pw::list_pivot_count(row.wide), step = pw::list_pivot_step(row.wide, row.n_cl);
        row.piv_off = (uint32_t)(row.wide ? row.cl32.size() : row.cl16.size());
        for (uint32_t k = 0; k < np; k++) {
            if (row.wide) row.cl32.push_back(row.cl32[(size_t)(k + 1) * step]);
            else row.cl16.push_back(row.cl16[(size_t)(k + 1) * step]);
        }
    }
    row.tot = (float)((double)cnt[1] + (double)cnt[0] * (double)w_out + (double)cnt[2] * (double)w_prev);
    row.x_in = 1.0f / row.tot;
    row.x_out = row.x_in * w_out;
    row.x_prev = row.x_in * w_prev;
    return 0;
}

}  // namespace

namespace pw {
// lane_decide -> lane_tight -> lane_chain for one target; out = { lane, kmax, tight, chain_lane }
PW_HD void lane_selftest_one(uint32_t n, const ListView &cl, uint32_t n_cl, uint32_t pp, float w_out, float w_prev, double r,
                             uint32_t *out) {
    LaneStep ls{0.0f, 0u, 0u, 0u, 0u, 0u, 0u};
    const uint32_t lane = lane_decide(n, n_cl, pp, r, w_out, w_prev, cl, ls);
    out[0] = lane;
    out[1] = lane == LANE_AMBIGUOUS ? ls.kmax : 0u;
    out[2] = lane == LANE_AMBIGUOUS ? lane_tight(n, pp, r, w_out, w_prev, ls) : lane;
    // the per-thread float chain: over the ambiguous prefix, or the whole row when decided
    if (lane == LANE_REDO) { out[3] = LANE_REDO; return; }
    const uint32_t kend = lane == LANE_AMBIGUOUS ? ls.kmax : n;
    const float x_in = 1.0f / ls.tot;
    uint32_t reads = 0;
    out[3] = lane_chain(kend, n_cl, pp, r, x_in, x_in * w_out, x_in * w_prev, cl, reads);
}

__global__ void __launch_bounds__(256)
lane_selftest_kernel(const uint8_t *cls, uint32_t n, const void *cl, uint32_t wide, uint32_t piv_off, uint32_t n_cl, uint32_t pp, float w_out,
                     float w_prev, const double *r, uint32_t n_r, uint32_t *chain, uint32_t *out4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_r) return;
    // the reference: sequential float32 cumsum + searchsorted (pecanpy.py:556-557), by this thread
    uint32_t cnt_in = 0, cnt_out = 0, cnt_pv = 0;
    for (uint32_t k = 0; k < n; k++) { const uint8_t c = cls[k]; cnt_in += c == 1; cnt_out += c == 0; cnt_pv += c == 2; }
    const float tot = (float)((double)cnt_in + (double)cnt_out * (double)w_out + (double)cnt_pv * (double)w_prev);
    const float x_in = 1.0f / tot, x_out = x_in * w_out, x_prev = x_in * w_prev;
    float c = 0.0f;
    uint32_t kc = n;
    for (uint32_t k = 0; k < n; k++) {
        c = c + (cls[k] == 1 ? x_in : (cls[k] == 0 ? x_out : x_prev));
        if ((double)c >= r[i]) { kc = k; break; }
    }
    chain[i] = kc;
    uint32_t o[4];
    lane_selftest_one(n, list_view_of(cl, wide, n_cl, piv_off), n_cl, pp, w_out, w_prev, r[i], o);
    out4[4 * i] = o[0]; out4[4 * i + 1] = o[1]; out4[4 * i + 2] = o[2]; out4[4 * i + 3] = o[3];
}
}  // namespace pw

namespace pw {
// the FLOATS form of the lane kernel for one target: row total by lane_chain(r = +inf), then the search over w / tot
PW_HD void lane_floats_one(uint32_t n, const ListView &cl, uint32_t n_cl, uint32_t pp, float w_out, float w_prev, double r,
                           uint32_t *choice, float *tot) {
    uint32_t reads = 0;
    float rowsum = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
#else
    const double inf = INFINITY;
#endif
    uint32_t res = lane_chain<true>(n, n_cl, pp, inf, 1.0f, w_out, w_prev, cl, reads, &rowsum);
    *tot = rowsum;
    if (res != LANE_CHAIN_END) { *choice = res; return; }   // (LANE_TIE)
    // round 5: the float64-bounded decision first (what the kernel does with the total from its table), the chain over
    // w / tot only where that leaves the step open
    uint32_t probes = 0, ks = 0;
    const uint32_t b = lane_decide_unit_bounded(n, n_cl, pp, r, rowsum, w_out, w_prev, cl, probes, ks);
    if (b <= n) { *choice = b == n ? LANE_CHAIN_END : b; return; }
    *choice = lane_chain<true>(n, n_cl, pp, r, 1.0f / rowsum, w_out / rowsum, w_prev / rowsum, cl, reads);
}

__global__ void __launch_bounds__(256)
lane_floats_selftest_kernel(const uint8_t *cls, uint32_t n, const void *cl, uint32_t wide, uint32_t piv_off, uint32_t n_cl, uint32_t pp, float w_out,
                            float w_prev, const double *r, uint32_t n_r, uint32_t *chain, uint32_t *lane, float *tots) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_r) return;
    float tot = 0.0f;   // the reference: sequential float32 w.sum(), w / tot, cumsum, searchsorted -- by this thread
    for (uint32_t k = 0; k < n; k++) tot = tot + (cls[k] == 1 ? 1.0f : (cls[k] == 0 ? w_out : w_prev));
    const float x_in = 1.0f / tot, x_out = w_out / tot, x_prev = w_prev / tot;
    float c = 0.0f;
    uint32_t kc = n;
    for (uint32_t k = 0; k < n; k++) {
        c = c + (cls[k] == 1 ? x_in : (cls[k] == 0 ? x_out : x_prev));
        if ((double)c >= r[i]) { kc = k; break; }
    }
    chain[i] = kc;
    float tl = 0.0f;
    uint32_t ch = 0;
    lane_floats_one(n, list_view_of(cl, wide, n_cl, piv_off), n_cl, pp, w_out, w_prev, r[i], &ch, &tl);
    lane[i] = ch;
    tots[2 * i] = tot;
    tots[2 * i + 1] = tl;
}
}  // namespace pw

// The FLOATS form of a lane-kernel step (1/p or 1/q not a power of two: arbitrary float32 row values): chain[i] = the
// reference's position for draw r[i] (sequential float32 w.sum(), w / tot, cumsum, searchsorted), lane[i] = the same by
// two closed-form chains of one thread (0xfffffffb: never reached, 0xfffffffa: tie budget), tots[2 i], tots[2 i + 1] =
// the sequential row total and the thread's.  on_device: one GPU thread per target.
PW_EXPORT int pw_selftest_lane_floats(int on_device, int device, const uint8_t *cls, uint32_t n, float w_out, float w_prev,
                                      const double *r, uint32_t n_r, uint32_t *chain, uint32_t *lane, float *tots) {
    if (!cls || !r || !chain || !lane || !tots || n == 0) return fail(PW_ERR_INVALID, "bad argument");
    LaneRow row;
    int rc = lane_row_setup(cls, n, w_out, w_prev, row, false);
    if (rc) return rc;
    if (!on_device) {
        float tot = 0.0f;
        for (uint32_t k = 0; k < n; k++) tot = tot + (cls[k] == 1 ? 1.0f : (cls[k] == 0 ? w_out : w_prev));
        const float x_in = 1.0f / tot, x_out = w_out / tot, x_prev = w_prev / tot;
        std::vector<float> c(n);
        float acc = 0.0f;
        for (uint32_t k = 0; k < n; k++) { acc = acc + (cls[k] == 1 ? x_in : (cls[k] == 0 ? x_out : x_prev)); c[k] = acc; }
        for (uint32_t i = 0; i < n_r; i++) {
            uint32_t lo = 0, hi = n;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((double)c[mid] >= r[i]) hi = mid; else lo = mid + 1; }
            chain[i] = lo;
            float tl = 0.0f;
            pw::lane_floats_one(n, row.view(), row.n_cl, row.pp, w_out, w_prev, r[i], &lane[i], &tl);
            tots[2 * (size_t)i] = tot;
            tots[2 * (size_t)i + 1] = tl;
        }
        return PW_OK;
    }
    int ndev = pw_device_count();
    if (ndev <= 0) return fail(PW_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(PW_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    uint8_t *d_cls = nullptr;
    void *d_cl = nullptr;
    double *d_r = nullptr;
    uint32_t *d_chain = nullptr, *d_lane = nullptr;
    float *d_tots = nullptr;
    auto cleanup = [&]() {
        for (void *q : {(void *)d_cls, d_cl, (void *)d_r, (void *)d_chain, (void *)d_lane, (void *)d_tots})
            if (q) (void)hipFree(q);
    };
    const size_t cl_bytes = row.wide ? row.cl32.size() * sizeof(uint32_t) : row.cl16.size() * sizeof(uint16_t);
    const void *cl_host = row.wide ? (const void *)row.cl32.data() : (const void *)row.cl16.data();
    const size_t nr = n_r ? n_r : 1;
    hipError_t e = hipMalloc((void **)&d_cls, n);
    if (e == hipSuccess) e = hipMalloc(&d_cl, cl_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_r, sizeof(double) * nr);
    if (e == hipSuccess) e = hipMalloc((void **)&d_chain, sizeof(uint32_t) * nr);
    if (e == hipSuccess) e = hipMalloc((void **)&d_lane, sizeof(uint32_t) * nr);
    if (e == hipSuccess) e = hipMalloc((void **)&d_tots, sizeof(float) * 2 * nr);
    if (e == hipSuccess) e = hipMemcpy(d_cls, cls, n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cl, cl_host, cl_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && n_r) e = hipMemcpy(d_r, r, sizeof(double) * (size_t)n_r, hipMemcpyHostToDevice);
    if (e == hipSuccess && n_r) {
        hipLaunchKernelGGL(pw::lane_floats_selftest_kernel, dim3((n_r + 255) / 256), dim3(256), 0, 0, d_cls, n, d_cl, row.wide, row.piv_off, row.n_cl,
                           row.pp, w_out, w_prev, d_r, n_r, d_chain, d_lane, d_tots);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(chain, d_chain, sizeof(uint32_t) * (size_t)n_r, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(lane, d_lane, sizeof(uint32_t) * (size_t)n_r, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(tots, d_tots, sizeof(float) * 2 * (size_t)n_r, hipMemcpyDeviceToHost);
    }
    cleanup();
    if (e != hipSuccess) return fail(PW_ERR_HIP, std::string("pw_selftest_lane_floats: ") + hipGetErrorString(e));
    return PW_OK;
}

// The bounded decision of the FLOATS form alone (host): lane[i] = lane_decide_unit_bounded's verdict for draw r[i] given the
// row's sequential float32 total -- the position, n (never reached) or 0xfffffffd (left open) -- and chain[i] = the
// reference's position (sequential float32 w.sum(), w / tot, cumsum, searchsorted; n: never reached).
PW_EXPORT int pw_selftest_lane_unit_bounded(const uint8_t *cls, uint32_t n, float w_out, float w_prev, const double *r, uint32_t n_r,
                                            uint32_t *chain, uint32_t *lane) {
    if (!cls || !r || !chain || !lane || n == 0) return fail(PW_ERR_INVALID, "bad argument");
    LaneRow row;
    int rc = lane_row_setup(cls, n, w_out, w_prev, row, false);
    if (rc) return rc;
    float tot = 0.0f;
    for (uint32_t k = 0; k < n; k++) tot = tot + (cls[k] == 1 ? 1.0f : (cls[k] == 0 ? w_out : w_prev));
    const float x_in = 1.0f / tot, x_out = w_out / tot, x_prev = w_prev / tot;
    std::vector<float> c(n);
    float acc = 0.0f;
    for (uint32_t k = 0; k < n; k++) { acc = acc + (cls[k] == 1 ? x_in : (cls[k] == 0 ? x_out : x_prev)); c[k] = acc; }
    for (uint32_t i = 0; i < n_r; i++) {
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((double)c[mid] >= r[i]) hi = mid; else lo = mid + 1; }
        chain[i] = lo;
        uint32_t probes = 0, ks = 0;
        lane[i] = pw::lane_decide_unit_bounded(n, row.n_cl, row.pp, r[i], tot, w_out, w_prev, row.view(), probes, ks);
        if (lane[i] > n && lane[i] != pw::LANE_AMBIGUOUS) return fail(PW_ERR_INVALID, "unexpected verdict");
        if (ks > lo) return fail(PW_ERR_INVALID, "k_safe beyond the reference's position");
    }
    return PW_OK;
}

// The FLOATS step with the interval decision in front of the chain (round 6): lane[i] = lane_decide_unit_bounded's verdict,
// tight[i] = that verdict when it is one, else lane_tight_values' (position, or 0xfffffffd: left to the float chain).
PW_EXPORT int pw_selftest_lane_unit_tight(const uint8_t *cls, uint32_t n, float w_out, float w_prev, const double *r, uint32_t n_r,
                                          uint32_t *chain, uint32_t *lane, uint32_t *tight) {
    if (!cls || !r || !chain || !lane || !tight || n == 0) return fail(PW_ERR_INVALID, "bad argument");
    LaneRow row;
    int rc = lane_row_setup(cls, n, w_out, w_prev, row, false);
    if (rc) return rc;
    float tot = 0.0f;
    for (uint32_t k = 0; k < n; k++) tot = tot + (cls[k] == 1 ? 1.0f : (cls[k] == 0 ? w_out : w_prev));
    const float x_in = 1.0f / tot, x_out = w_out / tot, x_prev = w_prev / tot;
    std::vector<float> c(n);
    float acc = 0.0f;
    for (uint32_t k = 0; k < n; k++) { acc = acc + (cls[k] == 1 ? x_in : (cls[k] == 0 ? x_out : x_prev)); c[k] = acc; }
    for (uint32_t i = 0; i < n_r; i++) {
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((double)c[mid] >= r[i]) hi = mid; else lo = mid + 1; }
        chain[i] = lo;
        uint32_t probes = 0, ks = 0;
        pw::BoundedAmb amb;
        lane[i] = pw::lane_decide_unit_bounded(n, row.n_cl, row.pp, r[i], tot, w_out, w_prev, row.view(), probes, ks, &amb);
        if (lane[i] > n && lane[i] != pw::LANE_AMBIGUOUS) return fail(PW_ERR_INVALID, "unexpected verdict");
        if (ks > lo) return fail(PW_ERR_INVALID, "k_safe beyond the reference's position");
        tight[i] = lane[i];
        if (lane[i] == pw::LANE_AMBIGUOUS && ks > 0) tight[i] = pw::lane_tight_values(n, row.pp, r[i], x_in, x_out, x_prev, ks, amb.f, amb.p_next, amb.z_abs);
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    if (getenv("PW_TIGHT_STATS")) {
        fprintf(stderr, "tight bail reasons:");
        for (int k = 1; k < 24; k++) if (pw::g_tight_reason[k]) fprintf(stderr, " [%d]=%llu", k, (unsigned long long)pw::g_tight_reason[k]);
        fprintf(stderr, "\n");
    }
#endif
    return PW_OK;
}

PW_EXPORT int pw_selftest_lane(int on_device, int device, const uint8_t *cls, uint32_t n, float w_out, float w_prev, const double *r,
                               uint32_t n_r, uint32_t *chain, uint32_t *lane, uint32_t *kmax, uint32_t *tight,
                               uint32_t *chain_lane) {
    if (!cls || !r || !chain || !lane || !kmax || !tight || n == 0) return fail(PW_ERR_INVALID, "bad argument");
    LaneRow row;
    int rc = lane_row_setup(cls, n, w_out, w_prev, row);
    if (rc) return rc;
    if (!on_device) {
        // the chain once: prefix sums, then one binary search per draw (the sums are non-decreasing)
        std::vector<float> c(n);
        float acc = 0.0f;
        for (uint32_t k = 0; k < n; k++) {
            acc = acc + (cls[k] == 1 ? row.x_in : (cls[k] == 0 ? row.x_out : row.x_prev));
            c[k] = acc;
        }
        for (uint32_t i = 0; i < n_r; i++) {
            uint32_t lo = 0, hi = n;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((double)c[mid] >= r[i]) hi = mid; else lo = mid + 1; }
            chain[i] = lo;
            uint32_t o[4];
            pw::lane_selftest_one(n, row.view(), row.n_cl, row.pp, w_out, w_prev, r[i], o);
            lane[i] = o[0]; kmax[i] = o[1]; tight[i] = o[2];
            if (chain_lane) chain_lane[i] = o[3];
        }
#if !defined(__HIP_DEVICE_COMPILE__)
        if (getenv("PW_TIGHT_STATS")) {
            fprintf(stderr, "tight bail reasons:");
            for (int k = 1; k < 24; k++) if (pw::g_tight_reason[k]) fprintf(stderr, " [%d]=%llu", k, (unsigned long long)pw::g_tight_reason[k]);
            fprintf(stderr, "\n");
        }
#endif
        return PW_OK;
    }
    int ndev = pw_device_count();
    if (ndev <= 0) return fail(PW_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(PW_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    uint8_t *d_cls = nullptr;
    void *d_cl = nullptr;
    double *d_r = nullptr;
    uint32_t *d_chain = nullptr, *d_out = nullptr;
    auto cleanup = [&]() {
        for (void *q : {(void *)d_cls, d_cl, (void *)d_r, (void *)d_chain, (void *)d_out})
            if (q) (void)hipFree(q);
    };
    const size_t cl_bytes = row.wide ? row.cl32.size() * sizeof(uint32_t) : row.cl16.size() * sizeof(uint16_t);
    const void *cl_host = row.wide ? (const void *)row.cl32.data() : (const void *)row.cl16.data();
    hipError_t e = hipMalloc((void **)&d_cls, n);
    if (e == hipSuccess) e = hipMalloc(&d_cl, cl_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_r, sizeof(double) * (size_t)(n_r ? n_r : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_chain, sizeof(uint32_t) * (size_t)(n_r ? n_r : 1));
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, sizeof(uint32_t) * 4 * (size_t)(n_r ? n_r : 1));
    if (e == hipSuccess) e = hipMemcpy(d_cls, cls, n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cl, cl_host, cl_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && n_r) e = hipMemcpy(d_r, r, sizeof(double) * (size_t)n_r, hipMemcpyHostToDevice);
    std::vector<uint32_t> out4((size_t)4 * n_r);
    if (e == hipSuccess && n_r) {
        hipLaunchKernelGGL(pw::lane_selftest_kernel, dim3((n_r + 255) / 256), dim3(256), 0, 0, d_cls, n, d_cl, row.wide, row.piv_off, row.n_cl, row.pp,
                           w_out, w_prev, d_r, n_r, d_chain, d_out);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(chain, d_chain, sizeof(uint32_t) * (size_t)n_r, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(out4.data(), d_out, sizeof(uint32_t) * 4 * (size_t)n_r, hipMemcpyDeviceToHost);
    }
    cleanup();
    if (e != hipSuccess) return fail(PW_ERR_HIP, std::string("pw_selftest_lane: ") + hipGetErrorString(e));
    for (uint32_t i = 0; i < n_r; i++) {
        lane[i] = out4[4 * (size_t)i]; kmax[i] = out4[4 * (size_t)i + 1]; tight[i] = out4[4 * (size_t)i + 2];
        if (chain_lane) chain_lane[i] = out4[4 * (size_t)i + 3];
    }
    return PW_OK;
}

// The weighted-row decision of the lane kernel (seqscan.h: lane_decide_weighted) on a host row, beside the reference's
// sequential float32 chain.  vals[k] = the step's value of neighbour k before normalisation, base[k] = its base value
// (what it would be as a plain "out" neighbour), cls[k] = 0 other (vals == base) / 1 common neighbour / 2 prev.
PW_EXPORT int pw_selftest_lane_weighted(const float *vals, const float *base, const uint8_t *cls, uint32_t n, const double *r,
                                        uint32_t n_r, uint32_t *chain, uint32_t *lane) {
    if (!vals || !base || !cls || !r || !chain || !lane || n == 0) return fail(PW_ERR_INVALID, "bad argument");
    float tot = 0.0f;
    for (uint32_t k = 0; k < n; k++) tot = tot + vals[k];                     // sequential float32 w.sum()
    std::vector<float> c(n);
    float acc = 0.0f;
    for (uint32_t k = 0; k < n; k++) { acc = acc + vals[k] / tot; c[k] = acc; }   // cumsum of fl32(w / tot)
    std::vector<pw::PrefixPair> pq(n);
    std::vector<double> dl;
    double trun = 0.0;
    bool any_pos = false, any_neg = false;
    std::vector<uint16_t> cl16;
    std::vector<uint32_t> cl32;
    const uint32_t wide = n > 65536u ? 1u : 0u;
    double run = 0.0, drun = 0.0, dprev = 0.0;
    uint32_t pp = 0xffffffffu;
    for (uint32_t k = 0; k < n; k++) {
        run += (double)base[k];
        trun += run;
        pq[k] = pw::PrefixPair{run, trun};
        if (cls[k] == 1) {
            if (vals[k] > base[k]) any_pos = true;
            if (vals[k] < base[k]) any_neg = true;
            drun += (double)vals[k] - (double)base[k];
            dl.push_back(drun);
            if (wide) cl32.push_back(k); else cl16.push_back((uint16_t)k);
        } else if (cls[k] == 2) {
            if (pp != 0xffffffffu) return fail(PW_ERR_INVALID, "at most one prev");
            pp = k;
            dprev = (double)vals[k] - (double)base[k];
        } else if (cls[k] != 0 || vals[k] != base[k]) return fail(PW_ERR_INVALID, "class 0 elements carry their base value");
    }
    const uint32_t n_cl = (uint32_t)dl.size();
    cl16.resize(cl16.size() + 8, 0xffffu);
    cl32.resize(cl32.size() + 8, 0xffffffffu);
    uint32_t piv_off = 0;
    if (pw::list_has_pivots(wide, n_cl)) {
        const uint32_t np = pw::list_pivot_count(wide), step = pw::list_pivot_step(wide, n_cl);
        piv_off = (uint32_t)(wide ? cl32.size() : cl16.size());
        for (uint32_t k = 0; k < np; k++) { if (wide) cl32.push_back(cl32[(size_t)(k + 1) * step]); else cl16.push_back(cl16[(size_t)(k + 1) * step]); }
    }
    dl.push_back(0.0);
    const pw::ListView view = pw::list_view_of(wide ? (const void *)cl32.data() : (const void *)cl16.data(), wide, n_cl, piv_off);
    if (any_pos && any_neg) return fail(PW_ERR_INVALID, "the common neighbours' differences must share one sign (that of q - 1)");
    const pw::WeightedRow wr{pq.data(), dl.data(), dprev, !any_neg};
    for (uint32_t i = 0; i < n_r; i++) {
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((double)c[mid] >= r[i]) hi = mid; else lo = mid + 1; }
        chain[i] = lo;
        uint32_t probes = 0, k_safe = 0;
        lane[i] = pw::lane_decide_weighted(n, n_cl, pp, r[i], tot, wr, view, probes, k_safe);
        if (k_safe > chain[i]) return fail(PW_ERR_INVALID, "lane_decide_weighted: k_safe beyond the chain's position");
    }
    return PW_OK;
}

// ---- node2vec+ thresholds (host only) -----------------------------------------------------------------
PW_EXPORT int pw_noise_thresholds_csr(const uint32_t *indptr, const float *data, uint32_t n_nodes, double gamma, float *thr) {
    if (!indptr || !thr || (!data && indptr[n_nodes] != 0)) return fail(PW_ERR_INVALID, "null pointer");
    pw::noise_thresholds_csr(indptr, data, n_nodes, gamma, thr);
    return PW_OK;
}

PW_EXPORT int pw_noise_thresholds_csr_numpy1(const uint32_t *indptr, const float *data, uint32_t n_nodes, double gamma, float *thr) {
    if (!indptr || !thr || (!data && indptr[n_nodes] != 0)) return fail(PW_ERR_INVALID, "null pointer");
    pw::noise_thresholds_csr(indptr, data, n_nodes, gamma, thr, true);
    return PW_OK;
}

PW_EXPORT int pw_noise_thresholds_dense(const double *data, uint32_t n_nodes, double gamma, float *thr) {
    if (!data || !thr) return fail(PW_ERR_INVALID, "null pointer");
    pw::noise_thresholds_dense(data, n_nodes, gamma, thr);
    return PW_OK;
}

// ---- edge-list ingestion (host only) ----------------------------------------------------------------
struct pw_edgelist {
    pw::EdgeList el;
};

PW_EXPORT int pw_edgelist_read(const char *path, int weighted, int directed, const char *delimiter, pw_edgelist **out) {
    if (!path || !out) return fail(PW_ERR_INVALID, "null pointer");
    *out = nullptr;
    pw_edgelist *h = new (std::nothrow) pw_edgelist();
    if (!h) return fail(PW_ERR_NOMEM, "out of host memory");
    int st;
    try {
        st = pw::read_edgelist(path, weighted != 0, directed != 0, delimiter, h->el);
    } catch (const std::bad_alloc &) {
        delete h;
        return fail(PW_ERR_NOMEM, "out of host memory while reading the edge list");
    }
    if (st == pw::EL_OK) { *out = h; return PW_OK; }
    delete h;
    if (st == pw::EL_IO_ERROR) return fail(PW_ERR_INVALID, std::string("cannot read ") + path);
    return fail(PW_ERR_UNSUPPORTED, "edge list needs the statement-by-statement reader");
}

PW_EXPORT int pw_edgelist_shape(const pw_edgelist *e, uint64_t *n_nodes, uint64_t *nnz, uint64_t *insertions,
                                uint64_t *id_bytes) {
    if (!e) return fail(PW_ERR_INVALID, "null pointer");
    if (n_nodes) *n_nodes = e->el.indptr.size() - 1;
    if (nnz) *nnz = e->el.indices.size();
    if (insertions) *insertions = e->el.insertions;
    if (id_bytes) *id_bytes = e->el.id_chars.size();
    return PW_OK;
}

PW_EXPORT int pw_edgelist_export(const pw_edgelist *e, uint32_t *indptr, uint32_t *indices, float *data,
                                 double *data64, uint64_t *id_offsets, char *id_chars) {
    if (!e) return fail(PW_ERR_INVALID, "null pointer");
    const pw::EdgeList &el = e->el;
    if (indptr) memcpy(indptr, el.indptr.data(), sizeof(uint32_t) * el.indptr.size());
    if (indices && !el.indices.empty()) memcpy(indices, el.indices.data(), sizeof(uint32_t) * el.indices.size());
    if (data && !el.data.empty()) memcpy(data, el.data.data(), sizeof(float) * el.data.size());
    if (data64 && !el.data64.empty()) memcpy(data64, el.data64.data(), sizeof(double) * el.data64.size());
    if (id_offsets) memcpy(id_offsets, el.id_off.data(), sizeof(uint64_t) * el.id_off.size());
    if (id_chars && !el.id_chars.empty()) memcpy(id_chars, el.id_chars.data(), el.id_chars.size());
    return PW_OK;
}

PW_EXPORT void pw_edgelist_destroy(pw_edgelist *e) { delete e; }

PW_EXPORT int pw_selftest_seqscan_f32(const float *x, uint32_t n, double r, int use_target, uint32_t chunk,
                                      uint32_t *index, float *sum) {
    if (!x || !index || !sum || chunk == 0) return fail(PW_ERR_INVALID, "bad argument");
    *index = seqscan_emulate<float>(x, n, r, use_target != 0, chunk, sum);
    return PW_OK;
}

PW_EXPORT int pw_selftest_seqscan_f64(const double *x, uint32_t n, double r, int use_target, uint32_t chunk,
                                      uint32_t *index, double *sum) {
    if (!x || !index || !sum || chunk == 0) return fail(PW_ERR_INVALID, "bad argument");
    *index = seqscan_emulate<double>(x, n, r, use_target != 0, chunk, sum);
    return PW_OK;
}
