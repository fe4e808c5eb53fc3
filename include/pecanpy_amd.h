/*
 * pecanpy_amd.h -- C ABI of the MI355X-native node2vec walk engine (libpecanpy_amd.so).
 *
 * The reference (krishnanlab/PecanPy) has no FFI: its operator boundary for walk generation is
 * the njit function
 *     Base._random_walks(tot_num_jobs, walk_length, random_state, start_node_idx_ary,
 *                        has_nbrs, move_forward, progress_proxy) -> uint32[tot_num_jobs, L+2]
 * (reference src/pecanpy/pecanpy.py:164-210) whose callbacks are built from the graph arrays by
 * get_has_nbrs()/get_move_forward() (pecanpy.py:522-561, 576-614; rw/sparse_rw.py:12-20).
 * The entry points below are what a ctypes/cffi binding of that boundary binds to: the graph
 * arrays become a device-resident handle (pw_csr_create / pw_dense_create), and one call
 * (pw_simulate*) replaces _random_walks + the two callbacks.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns 0 on success, a negative pw_status on failure; pw_last_error()
 *     returns a thread-local human-readable message.  Nothing prints or aborts.
 *   - host buffers are borrowed for the duration of a call; device copies of the graph are owned
 *     by the handle until pw_graph_destroy().
 *   - one in-flight pw_simulate* per handle.
 *   - walk matrix layout = the reference's: row i = [start, n_1 .. n_L, len_i], unused cells 0,
 *     len_i = L+1 normally, 1 for a start without neighbours, j for a dead end before step j
 *     (pecanpy.py:182-206).
 *   - random stream = ONE MT19937 stream seeded like the reference (np.random.seed(random_state)
 *     inside _random_walks, pecanpy.py:177-178); walk i step j consumes double #(S_i + j),
 *     S_i = sum of (len-1) over earlier walks.  stream_skip shifts S_0 (multi-GPU shards).
 */
#ifndef PECANPY_AMD_H
#define PECANPY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pw_graph pw_graph; /* opaque, device resident */

typedef enum {
    PW_OK = 0,
    PW_ERR_INVALID = -1,     /* bad argument */
    PW_ERR_HIP = -2,         /* HIP runtime error (message has the hipError string) */
    PW_ERR_NO_DEVICE = -3,   /* no usable GPU */
    PW_ERR_UNSUPPORTED = -4, /* mode/option not available for this graph kind */
    PW_ERR_NOMEM = -5
} pw_status;

/* walk modes = the reference's mode classes (pecanpy.py:293-614) */
typedef enum {
    PW_MODE_SPARSE_OTF = 0,            /* SparseOTF.move_forward, pecanpy.py:543-559 */
    PW_MODE_DENSE_OTF = 1,             /* DenseOTF.move_forward,  pecanpy.py:597-612 */
    PW_MODE_PRECOMP = 2,               /* PreComp.move_forward,   pecanpy.py:409-438 */
    PW_MODE_FIRST_ORDER_UNWEIGHTED = 3,/* FirstOrderUnweighted,   pecanpy.py:299-309 */
    PW_MODE_PRECOMP_FIRST_ORDER = 4    /* PreCompFirstOrder,      pecanpy.py:319-334 */
} pw_mode;

typedef struct {
    uint64_t total_steps;     /* sampled transitions = sum_i (len_i - 1) */
    uint64_t overflow_reads;  /* steps where the float CDF never reached r (choice == degree,
                                 SURVEY.md App. D quirk 1; mirrored from the reference) */
    uint64_t clamped_reads;   /* of those, reads that would have left the index buffer */
    uint64_t dead_end_walks;  /* walks that stopped early at a vertex without out-edges */
    uint64_t repair_rounds;   /* extra passes needed to re-address the stream after dead ends */
    double walk_kernel_ms;    /* HIP-event time of the walk kernel launches of this call */
    double rng_kernel_ms;     /* HIP-event time of the MT19937 expansion kernels */
    uint32_t walk_kernel_launches;
    uint32_t stream_addressing; /* 0: the reference's exact draw assignment (also on sink-heavy directed graphs: block-wise
                                   repair); 1: nominal per-walk slots -- only with the explicit opt-in
                                   PECANPY_AMD_NOMINAL_STREAM=1 in the environment, after 32 re-addressing passes */
    /* lane kernel (one walk per lane, csrc/walk_lanes.hip.h): unit-weight CSR graphs, 1/p and 1/q powers of two */
    uint32_t lane_kernel;       /* 1: the call ran on the lane kernel; 2: on its float-chain form (1/p or 1/q not a power of two);
                                   3: on its weighted form (weighted CSR graphs: float64-bounded decision, eager_steps = the steps
                                   the wave-per-walk scan decided) */
    uint32_t lane_rounds;       /* lane kernel launches of the call: walks whose step needs the float32 chain are parked,
                                   the chains of a whole queue run in one launch, the next round resumes the walks */
    uint64_t redo_walks;        /* walks a fast kernel handed to the complete one: lane kernel -> wave-per-walk kernel
                                   (tie budget, rows outside the exact range, overflow reads without a line), register-only
                                   dense kernel -> column-space kernel with the float64 chain (undecided steps) */
    uint64_t list_entries_read; /* common-neighbour list entries the lane kernel read (uint16 positions: 2 bytes each; uint32 for
                                   rows of more than 65536 entries) */
    uint64_t ambiguous_steps;   /* steps the a-priori rounding bound left open (settled by the interval decision or the chain) */
    double lane_kernel_ms;      /* HIP-event time of the lane kernel launches alone */
    uint64_t wave_chain_steps;  /* of the ambiguous steps, those that needed the float32 chain itself */
    double param_index_ms;      /* device time spent in THIS call building an index that depends on (p, q, extend):
                                   per-edge normalisers of weighted graphs, hint tables; 0 when cached in the handle */
    /* PECANPY_AMD_VERIFY_TIGHT=1 (test mode of the lane kernel): every step the interval decision settled is decided
     * again by the sequential float32 chain on the device */
    uint64_t verify_checked;    /* steps re-decided */
    uint64_t verify_mismatch;   /* of those, steps where the chain picked a different neighbour (must be 0) */
    uint64_t verify_dropped;    /* steps not recorded because the record buffer was full */
    uint64_t verify_ties;       /* steps the chain declined (rounding-tie budget): not compared */
    /* partial lane index (the byte budget left the longest common-neighbour lists out, see pw_csr_create) */
    uint64_t eager_steps;       /* lane-kernel steps that arrived by an entry without a stored list: decided by one wavefront each
                                   (lanes_eager_kernel: membership searched), the walk resumed in the lane kernel */
    uint32_t index_max_list;    /* longest list the index stores beyond the edge lines (0xffffffff: all of them; 0: no lane index) */
    uint32_t reserved0;
} pw_stats;

/* ---- introspection ------------------------------------------------------------------- */
const char *pw_version(void);
const char *pw_last_error(void);
int pw_device_count(void);
/* One-time start-up of this library on `device` (the first stream a process creates through the library costs ~140 ms on an
 * MI355X box -- code objects, queues -- whatever torch or another library has initialised before): a caller with host work in front
 * of its first handle (reading an edge list: cli.py:328-337, graph.py:270-341) runs this on a helper thread beside that work and
 * pw_csr_create / pw_dense_create find the runtime warm.  *ms (optional) receives the wall clock of the call.  No reference
 * counterpart (Numba's JIT compilation at the first call is the reference's start-up cost, pecanpy.py:164). */
int pw_warmup(int device, double *ms);

/* ---- graph handles --------------------------------------------------------------------- */
/* CSR in the reference's SparseGraph layout (graph.py:409-413): indptr uint32[n_nodes+1],
 * indices uint32[nnz] ascending and duplicate-free per row, data float32[nnz].
 * data may be NULL: all weights 1.0 (the reference's unweighted graphs, graph.py:170,480).
 * Besides the three arrays the handle owns the membership index built here on the device (per-row
 * Bloom filters, adjacency hash index, key stream, vertex records and the lane index: one 64-byte edge line per CSR entry
 * with the common-neighbour list of the edge, longer lists in an overflow array; weighted graphs WITH self loops get
 * none): about 180 bytes per CSR entry at RMAT-22.
 * Limits: the 32-bit offsets of the index allow about 2^33 hash slots, i.e. graphs up to ~2 * 10^9 CSR
 * entries (PW_ERR_INVALID "graph too large" beyond that; the reference's own limit is nnz < 2^32).
 * Environment: PECANPY_AMD_NO_LAZY=1 skips the per-edge records (every step then takes the eager path;
 * used by the test-suite to cross-check the two step implementations).  PECANPY_AMD_INDEX_BUDGET=<bytes> bounds the
 * overflow array of the common-neighbour lists (default: half of the free device memory): beyond it the index is
 * PARTIAL -- edge lines always, the longest lists left out -- and steps that arrive by an entry without its list are
 * decided by one wavefront each (pw_stats.eager_steps) while the walk stays in the lane kernel. */
int pw_csr_create(const uint32_t *indptr, const uint32_t *indices, const float *data,
                  uint32_t n_nodes, uint32_t nnz, int device, pw_graph **out);

/* Device time (ms) of the index kernels pw_csr_create ran, device bytes of the index, and the number of entries
 * of the lane kernel's common-neighbour lists (0: lane index not built).  Any pointer may be NULL. */
int pw_graph_index_info(const pw_graph *g, double *build_ms, uint64_t *index_bytes, uint64_t *lane_list_entries);
/* Test hook: the lane index decoded to flat arrays (any pointer may be NULL).  For every CSR entry e = (u -> v):
 * n_in[e] = |N(u) & N(v)| (less u itself when u has a self loop: the reference takes prev out of the common neighbours,
 * sparse_rw.py:79-87), rev_pos[e] = position of u in row v (0xffffffff: v -> u is not an edge); entries = the
 * positions in row v of those common neighbours, ascending, concatenated in entry order (lane_list_entries values,
 * pw_graph_index_info).  PW_ERR_UNSUPPORTED when the graph has no lane index (weighted graphs with self loops). */
int pw_lane_index_export(pw_graph *g, uint32_t *n_in, uint32_t *rev_pos, uint32_t *entries);

/* Dense adjacency in the reference's DenseGraph layout (graph.py:576-580): float64[n, n]
 * row-major; nonzero mask = (data != 0). */
int pw_dense_create(const double *data, uint32_t n_nodes, int device, pw_graph **out);

/* Unweighted dense graph from its bit-packed adjacency: row u = ceil(n/64) uint64 words, bit x of the
 * row set <=> nonzero[u, x] (the reference's bool mask, graph.py:580, one bit per entry).  The
 * pointer may be a host pointer (on_device = 0) or a device pointer on `device` (on_device = 1);
 * lets a 100k-node dense graph be created without materialising the 80 GB float64 matrix. */
int pw_dense_create_bits(const uint64_t *adjbits, uint32_t n_nodes, int on_device, int device,
                         pw_graph **out);

/* node2vec+ noise thresholds, float32[n_nodes] (sparse_rw.py:22-35 / dense_rw.py:11-19);
 * required before a call with extend != 0. */
int pw_graph_set_thresholds(pw_graph *g, const float *thr);

void pw_graph_destroy(pw_graph *g);

/* ---- the walk operator ----------------------------------------------------------------- */
/*
 * Replaces Base._random_walks + has_nbrs + move_forward.
 *   starts      uint32[n_jobs]  (already shuffled by the caller, pecanpy.py:135-141)
 *   out         uint32[n_jobs * (walk_length + 2)]
 *   has_seed=0  -> seed taken from the OS (reference: random_state=None)
 *   stream_skip doubles of the stream consumed by earlier shards (0 for a whole job array)
 * pw_simulate takes host pointers (copies in/out); pw_simulate_device takes device pointers on
 * the handle's GPU (nothing crosses PCIe) and is what bench.py times.
 */
int pw_simulate(pw_graph *g, int mode, double p, double q, int extend, const uint32_t *starts,
                uint64_t n_jobs, uint32_t walk_length, int has_seed, uint32_t seed,
                uint64_t stream_skip, uint32_t *out, pw_stats *stats);

int pw_simulate_device(pw_graph *g, int mode, double p, double q, int extend,
                       const uint32_t *d_starts, uint64_t n_jobs, uint32_t walk_length,
                       int has_seed, uint32_t seed, uint64_t stream_skip, uint32_t *d_out,
                       pw_stats *stats);

/* ---- several GPUs from ONE process -------------------------------------------------------
 * The reference is one process: `pecanpy` reads a graph, calls simulate_walks() once and writes the result
 * (src/pecanpy/cli.py:328-351); its parallelism is Numba's thread pool inside _random_walks (pecanpy.py:165, 189).  The
 * counterpart here is one host thread per device inside one call -- no launcher, no torch.distributed:
 *   pw_csr_create_multi   builds the handle on devices[0] and REPLICATES it to the other devices (pw_graph_replicate:
 *                         CSR + the whole per-graph index copied device to device over xGMI, not rebuilt); a device may
 *                         be named more than once (every entry is a replica with its own streams and scratch).
 *                         out_handles[n_devices] receives the handles (destroy each with pw_graph_destroy).
 *   pw_device_mask_to_list  bit d of device_mask set <=> device d; returns how many devices the mask names (negative
 *                         pw_status when it names one that is not visible) and writes the first `cap` of them.
 *   pw_simulate_multi     the walk operator over replicas of one graph: the job array is split into contiguous shards
 *                         (the prange static chunking of pecanpy.py:189), shard i starts in the ONE random stream where
 *                         the earlier shards' draws end (announced as by pw_count_stream_draws; on directed graphs with
 *                         dead ends the later shards are walked again with the draws actually consumed), so the matrix
 *                         is bit-identical to pw_simulate on one handle.  starts: host pointer.  out_on_device = 0: out
 *                         is host memory, every device copies its rows out itself (its own PCIe link); 1: out is device
 *                         memory on handles[0]'s GPU, the other devices' rows arrive by peer copies (xGMI).
 *                         Alias / first-order modes (sequential stream) run on handles[0] alone.  pw_stats: sums; the
 *                         kernel times are the maximum over the shards (they run side by side).
 * Thresholds for extend != 0 are set per handle (pw_graph_set_thresholds; a replica inherits what its source had). */
int pw_graph_replicate(const pw_graph *src, int device, pw_graph **out);
int pw_device_mask_to_list(uint64_t device_mask, int *devices, int cap);
int pw_csr_create_multi(const uint32_t *indptr, const uint32_t *indices, const float *data, uint32_t n_nodes, uint32_t nnz,
                        const int *devices, int n_devices, pw_graph **out_handles);
int pw_simulate_multi(pw_graph *const *handles, int n_handles, int mode, double p, double q, int extend,
                      const uint32_t *starts, uint64_t n_jobs, uint32_t walk_length, int has_seed, uint32_t seed,
                      uint64_t stream_skip, uint32_t *out, int out_on_device, pw_stats *stats);

/* One transition of the on-the-fly modes for a given (cur, prev) -- the operator boundary of the reference's
 * callbacks: move_forward(cur, prev) (pecanpy.py:543-559 / 597-612) with the uniform draw r in [0, 1) supplied by the
 * caller (the reference draws it with np.random.random() inside), and get_normalized_probs(cur, prev)
 * (rw/sparse_rw.py:51-130, rw/dense_rw.py:34-118).  has_prev = 0: first step of a walk (prev ignored).
 *   pw_step : *next = sampled neighbour, *position (may be NULL) = its index in cur's row (== degree when the float
 *             CDF never reached r: the reference's overflow read, mirrored)
 *   pw_probs: probs = float32[degree(cur)] for CSR handles, float64[degree(cur)] for dense handles (room for the
 *             maximum degree of the graph), *n = degree(cur)
 * The same device code as the walk kernels' eager step; one launch per call (API compatibility and tests, not a
 * throughput path). */
int pw_step(pw_graph *g, int mode, double p, double q, int extend, uint32_t cur, int has_prev, uint32_t prev, double r,
            uint32_t *next, uint32_t *position);
int pw_probs(pw_graph *g, int mode, double p, double q, int extend, uint32_t cur, int has_prev, uint32_t prev, void *probs,
             uint32_t *n);

/* Alias tables of the PreComp modes, built on the device and kept in the handle.
 *   first_order = 0: PreComp.preprocess_transition_probs (pecanpy.py:442-507): sum(deg^2) entries,
 *                    table of (v, k-th neighbour as prev) at alias_indptr[v] + deg(v) * k
 *   first_order = 1: PreCompFirstOrder.preprocess_transition_probs (pecanpy.py:336-361): nnz entries
 * pw_simulate* with PW_MODE_PRECOMP / PW_MODE_PRECOMP_FIRST_ORDER builds them on demand.
 * pw_precomp_export copies them to host arrays (alias_indptr uint64[n_nodes+1] may be NULL for
 * first_order); *n_entries receives the table length (call with NULL arrays to query it). */
int pw_precomp_build(pw_graph *g, double p, double q, int extend, int first_order);
int pw_precomp_export(pw_graph *g, uint64_t *alias_indptr, uint32_t *alias_j, float *alias_q,
                      uint64_t *n_entries);

/* Number of stream doubles the jobs starts[0..n_jobs) consume when no walk dead-ends mid-way
 * (= walk_length x number of starts with at least one neighbour).  Lets a multi-GPU driver
 * compute each shard's stream_skip without running the walks. */
int pw_count_stream_draws(pw_graph *g, const uint32_t *starts, uint64_t n_jobs,
                          uint32_t walk_length, uint64_t *out_draws);

/* The draws [stream_skip, stream_skip + n_draws) of `seed`'s MT19937 stream, expanded once and KEPT in the handle:
 * pw_simulate_device calls with that seed whose draws lie inside use them in place (no jump-ahead, no expansion) until
 * pw_stream_release, the next pw_stream_hold or a pw_simulate call that walks in parts.  For a shard walked in several
 * calls (chunks that travel while the next is walked): one jump tree per shard instead of one per chunk. */
int pw_stream_hold(pw_graph *g, uint32_t seed, uint64_t stream_skip, uint64_t n_draws);
int pw_stream_release(pw_graph *g);

/* ---- skip-gram training over a walk matrix (the stage after the walks) ------------------- */
/* Word2Vec(walks, sg=1, negative, window, epochs) of Base.embed / cli.learn_embeddings (pecanpy.py:276-290,
 * cli.py:307-325) as a HIP kernel: skip-gram with negative sampling, word2vec.c's update rule, unigram^0.75 negative
 * table, walks thinned by the subsampling of frequent nodes (sample, 0 = off) before the windows are taken, window
 * shrunk at random, learning rate alpha -> min_alpha linearly over the run.  Every random choice is a hash of (seed,
 * epoch, walk, position, ...): the SET of updates depends on the seed only, their order on `workers`:
 *   workers = 1  one wavefront, sentence order: deterministic; equals the sequential restatement oracle/sgns_ref.c
 *                within float tolerance (tests/test_gpu_sgns.py) -- what gensim's workers=1 is to gensim;
 *   workers = 0  as many wavefronts as the corpus feeds (hogwild, unsynchronised updates, like gensim's threads);
 *   workers > 1  that many wavefronts.
 * Not bit-comparable with gensim itself (its own random streams; not installed here).
 *   walks   host uint32[n_walks, walk_length + 2] as pw_simulate returns it (indices into 0..n_nodes-1; a node id
 *           >= n_nodes or a length cell > walk_length + 1 is rejected with PW_ERR_INVALID)
 *   vectors host float32[n_nodes, dim] (out), dim <= 512 */
int pw_sgns_train(int device, const uint32_t *walks, uint64_t n_walks, uint32_t walk_length, uint32_t n_nodes,
                  uint32_t dim, uint32_t window, uint32_t negative, uint32_t epochs, float alpha, float min_alpha,
                  float sample, uint32_t seed, uint32_t workers, float *vectors);

/* ---- random stream service (host side; usable without a GPU) ---------------------------- */
/* doubles #offset.. of RandomState(seed).random_sample, produced with MT19937 jump-ahead. */
int pw_mt_random_sample(uint32_t seed, uint64_t offset, uint64_t n, double *out);

/* Test hook: the same doubles as the DEVICE produces them for a walk call (jump-ahead tree + expansion kernels on the
 * handle's GPU), copied to the host -- lets the tests compare the device stream with pw_mt_random_sample / NumPy at
 * the offsets the last shards of a multi-GPU run start from. */
int pw_stream_sample_device(pw_graph *g, uint32_t seed, uint64_t offset, uint64_t n, double *out);

/* ---- node2vec+ noisy-edge thresholds (host side; usable without a GPU) --------------------- */
/* thr[i] = max(mean(row i) + gamma * std(row i), 0) exactly as the reference's NumPy expression evaluates it
 * (rw/sparse_rw.py:22-35 on float32 CSR rows; rw/dense_rw.py:11-19 on the non-zero entries of float64 rows):
 * NumPy's pairwise summation, every step in the array's precision.  Rows without entries give NaN, as there.
 * The result is what pw_graph_set_thresholds() expects. */
int pw_noise_thresholds_csr(const uint32_t *indptr, const float *data, uint32_t n_nodes, double gamma, float *thr);
int pw_noise_thresholds_dense(const double *data, uint32_t n_nodes, double gamma, float *thr);
/* pw_noise_thresholds_csr evaluates mean + gamma * std as NumPy >= 2 does (float32 throughout); this variant as NumPy
 * 1.x does (the reference pins numpy==1.23.2: float32 scalar * Python float -> float64, one rounding on store).  They
 * differ by an ulp when gamma * std is not exact in float32 (gamma = 0.1, ...); the Python layer picks the one that
 * matches the installed NumPy, i.e. what the reference's expression would give in the same environment. */
int pw_noise_thresholds_csr_numpy1(const uint32_t *indptr, const float *data, uint32_t n_nodes, double gamma, float *thr);

/* ---- edge-list ingestion (host side; usable without a GPU) -------------------------------- */
/* Fast path of AdjlstGraph.read + to_csr (reference src/pecanpy/graph.py:270-341): parses a 2- or
 * 3-column edge list into the reference's CSR (vertices numbered by first appearance, rows ascending,
 * float32 weights, last duplicate wins) and the vertex-ID table.
 * Returns PW_OK, PW_ERR_INVALID (file cannot be read), or PW_ERR_UNSUPPORTED when the file needs the
 * statement-by-statement reader to reproduce Python-level behaviour (malformed line, non-positive or
 * exotic weight literal, conflicting duplicate edge -- the cases where the reference warns or raises --
 * or non-ASCII bytes); the caller then falls back to it.  The handle owns the arrays until destroyed. */
typedef struct pw_edgelist pw_edgelist;
int pw_edgelist_read(const char *path, int weighted, int directed, const char *delimiter, pw_edgelist **out);
/* n_nodes, nnz (distinct directed edges), insertions (the reference's num_edges counter), id_bytes */
int pw_edgelist_shape(const pw_edgelist *e, uint64_t *n_nodes, uint64_t *nnz, uint64_t *insertions,
                      uint64_t *id_bytes);
/* indptr uint32[n_nodes+1], indices uint32[nnz], data float32[nnz] (the CSR of to_csr), data64
 * float64[nnz] (the weights as parsed: AdjlstGraph.to_dense keeps float64), id_offsets uint64[n_nodes+1],
 * id_chars char[id_bytes] (vertex i's ID = id_chars[id_offsets[i] : id_offsets[i+1]]); NULL = skip */
int pw_edgelist_export(const pw_edgelist *e, uint32_t *indptr, uint32_t *indices, float *data,
                       double *data64, uint64_t *id_offsets, char *id_chars);
void pw_edgelist_destroy(pw_edgelist *e);

/* ---- self test hooks (host only, no GPU needed -- except pw_selftest_lane with on_device) -- */
/* Runs the binade-scan emulation of csrc/seqscan.h on a host array: returns through *index the
 * position np.searchsorted(np.cumsum(x), r) would return under sequential float32 semantics
 * (n if never reached) and through *sum the sequential float32 sum.  chunk = elements per pass. */
int pw_selftest_seqscan_f32(const float *x, uint32_t n, double r, int use_target, uint32_t chunk,
                            uint32_t *index, float *sum);
int pw_selftest_seqscan_f64(const double *x, uint32_t n, double r, int use_target, uint32_t chunk,
                            uint32_t *index, double *sum);

/* Exact-arithmetic decision of the float32 CDF search used by the unit-weight walk step (csrc/seqscan.h:
 * exact_thresholds_f32), on a host row: cls[k] = 0 (other neighbour, weight w_out), 1 (common neighbour,
 * weight 1) or 2 (prev, weight w_prev); w_out, w_prev powers of two.  For every r[i]: chain[i] = what the
 * reference's np.searchsorted(np.cumsum(w / w.sum()), r) returns under sequential float32 semantics
 * (pecanpy.py:556-557; n if never reached), exact[i] = the index decided in integer arithmetic, or
 * 0xffffffff when a partial sum lies inside the drift bound (the kernel then runs the float chain). */
int pw_selftest_exact_decision(const uint8_t *cls, uint32_t n, float w_out, float w_prev, const double *r,
                               uint32_t n_r, uint32_t *chain, uint32_t *exact);
/* The same decision as one thread of the lane kernel takes it, from the ascending positions of the common neighbours
 * (csrc/seqscan.h: lane_decide -> lane_tight -> lane_chain), beside the sequential float32 chain:
 *   chain[i]      = np.searchsorted(np.cumsum(float32 probs), r[i]) (n if never reached),
 *   lane[i]       = lane_decide: decided index, 0xfffffffd when a partial sum of the exact CDF lies inside the a-priori
 *                   drift bound (then kmax[i] = number of leading positions the chain may need: chain[i] < kmax[i] or
 *                   chain[i] == n), 0xfffffffc when the row is outside the exact range,
 *   tight[i]      = lane[i] when that is decided, else the list-free interval decision (lane_tight: the chain's
 *                   systematic drift bounded from the class counts lane_decide already knows): index or 0xfffffffd,
 *   chain_lane[i] = (may be NULL) the float32 chain as ONE thread evaluates it (lane_chain) over the first kmax[i]
 *                   positions (the whole row when lane[i] is decided): position, 0xfffffffb when that prefix never
 *                   reaches r[i], 0xfffffffa on a rounding tie the thread cannot afford.
 * on_device != 0: every target is decided by one thread of a kernel on `device` (the sequential chain included), i.e.
 * by the code the walk kernels run -- device code generation, v_rcp_f32 -- instead of the host build of the same
 * routines.  Rows of more than 65536 entries use uint32 positions, shorter ones uint16 (the lane index's formats). */
int pw_selftest_lane(int on_device, int device, const uint8_t *cls, uint32_t n, float w_out, float w_prev, const double *r,
                     uint32_t n_r, uint32_t *chain, uint32_t *lane, uint32_t *kmax, uint32_t *tight, uint32_t *chain_lane);
/* The step of the lane kernel's FLOATS form (unit weights, 1/p or 1/q NOT a power of two: w_out, w_prev arbitrary
 * positive float32): chain[i] = the reference's position (sequential float32 w.sum(), w / tot, cumsum, searchsorted),
 * lane[i] = the same from two closed-form chains of one thread (lane_chain with r = +inf for the total, then the
 * search; 0xfffffffb never reached, 0xfffffffa rounding-tie budget), tots[2 i] / tots[2 i + 1] = the sequential row
 * total / the thread's.  on_device: one GPU thread per target. */
int pw_selftest_lane_floats(int on_device, int device, const uint8_t *cls, uint32_t n, float w_out, float w_prev,
                            const double *r, uint32_t n_r, uint32_t *chain, uint32_t *lane, float *tots);
/* The bounded decision of that step alone (csrc/seqscan.h: lane_decide_unit_bounded, round 5 -- the real prefix sums of the
 * three row values in closed form + a rigorous bound on the float32 chain's drift), host only, given the row's sequential
 * float32 total: lane[i] = the position, n (never reached) or 0xfffffffd (left open: the chain decides); chain[i] = the
 * reference's position (n: never reached).  Fails when a k_safe lies beyond the reference's position. */
int pw_selftest_lane_unit_bounded(const uint8_t *cls, uint32_t n, float w_out, float w_prev, const double *r, uint32_t n_r,
                                  uint32_t *chain, uint32_t *lane);
/* ... with the interval decision of round 6 in front of the chain (csrc/seqscan.h: lane_tight_values -- the float32 chain's
 * systematic drift bounded from the class counts the bounded decision already has, for arbitrary float32 values):
 * tight[i] = lane[i] when that is a verdict, else the interval decision's position or 0xfffffffd (left to the float chain). */
int pw_selftest_lane_unit_tight(const uint8_t *cls, uint32_t n, float w_out, float w_prev, const double *r, uint32_t n_r,
                                uint32_t *chain, uint32_t *lane, uint32_t *tight);
/* The lane kernel's decision for WEIGHTED rows (csrc/seqscan.h: lane_decide_weighted: float64 prefix sums of the step's
 * values with a rigorous bound on the float32 chain's drift), host only.  vals[k] = the step's value of neighbour k
 * before normalisation (what get_normalized_probs holds at sparse_rw.py:87 / :126), base[k] = its value as a plain
 * "out" neighbour, cls[k] = 0 other (vals == base) / 1 common neighbour of prev and cur / 2 prev.  chain[i] = the
 * reference's position for draw r[i] (sequential float32 w.sum(), w / tot, cumsum, searchsorted; n: never reached),
 * lane[i] = the decision: the same position, or 0xfffffffd when a partial sum lies within the bound of the draw (the
 * step then takes the wave-per-walk scan). */
int pw_selftest_lane_weighted(const float *vals, const float *base, const uint8_t *cls, uint32_t n, const double *r,
                              uint32_t n_r, uint32_t *chain, uint32_t *lane);
/* float64 flavour (DenseOTF column-space kernel, dense_rw.py:34-72 semantics; exact_thresholds_f64). */
int pw_selftest_exact_decision_f64(const uint8_t *cls, uint32_t n, double w_out, double w_prev, const double *r,
                                   uint32_t n_r, uint32_t *chain, uint32_t *exact);

#ifdef __cplusplus
}
#endif
#endif
