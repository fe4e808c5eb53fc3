// aux_kernels.hip.h -- MT19937 stream expansion and stream-offset bookkeeping kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "walk_sparse.hip.h"

namespace pw {

// ---------------------------------------------------------------------------------------------
// MT19937 expansion: generator `gen` owns n_blocks consecutive 624-word blocks of the stream and
// writes them as 312 doubles each (np.random.random(): two words -> 53 bits; reference call
// site src/pecanpy/pecanpy.py:557; numba:cpython/randomimpl.py:133-146).
//   states_in : [n_gen][624] pre-twist state of each generator's first block
//   states_out: optional [n_gen][624] state after the last block (carry to the next batch)
//   out       : doubles, generator g writes out[(g * n_blocks + b) * 312 + t]
// One 256-thread workgroup per generator; the 624-word recurrence is evaluated in three
// dependency phases (i<227 | 227<=i<454 | 454<=i<624) straight from LDS.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mt_mix_dev(uint32_t hi, uint32_t lo, uint32_t far) {
    uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper_dev(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__global__ void __launch_bounds__(256)
mt_expand_kernel(const uint32_t *__restrict__ states_in, uint32_t *__restrict__ states_out,
                 double *__restrict__ out, uint64_t n_blocks, uint64_t total_blocks) {
    __shared__ uint32_t mt[624];
    const int t = threadIdx.x;
    const uint64_t gen = blockIdx.x;
    for (int i = t; i < 624; i += 256) mt[i] = states_in[gen * 624 + i];
    __syncthreads();
    uint64_t first = gen * n_blocks;
    uint64_t nb = first >= total_blocks ? 0 : (total_blocks - first < n_blocks ? total_blocks - first : n_blocks);
    double *dst = out + first * 312;
    for (uint64_t b = 0; b < nb; b++) {
        // phase 1: i in [0, 227): needs old[i], old[i+1], old[i+397]
        uint32_t v0 = 0, v1 = 0, v2 = 0;
        if (t < 227) v0 = mt_mix_dev(mt[t], mt[t + 1], mt[t + 397]);
        __syncthreads();
        if (t < 227) mt[t] = v0;
        __syncthreads();
        // phase 2: i in [227, 454): needs old[i], old[i+1], new[i-227]
        if (t < 227) v1 = mt_mix_dev(mt[t + 227], mt[t + 228], mt[t]);
        __syncthreads();
        if (t < 227) mt[t + 227] = v1;
        __syncthreads();
        // phase 3: i in [454, 624): needs old[i], old[i+1] (new[0] for i = 623), new[i-227]
        if (t < 170) v2 = mt_mix_dev(mt[t + 454], (t + 455 < 624) ? mt[t + 455] : mt[0], mt[t + 227]);
        __syncthreads();
        if (t < 170) mt[t + 454] = v2;
        __syncthreads();
        for (int i = t; i < 312; i += 256) {
            uint32_t a = mt_temper_dev(mt[2 * i]) >> 5;
            uint32_t c = mt_temper_dev(mt[2 * i + 1]) >> 6;
            dst[b * 312 + i] = ((double)a * 67108864.0 + (double)c) * (1.0 / 9007199254740992.0);
        }
        // the next phase-1 reads happen after its own barrier; writes above only read mt
    }
    __syncthreads();
    if (states_out)
        for (int i = t; i < 624; i += 256) states_out[gen * 624 + i] = mt[i];
}

// ---------------------------------------------------------------------------------------------
// Jump-ahead on the device: dst_state = g(A) src_state for a polynomial g of degree < 19937
// (host table MtJump::pow2_table(): g = x^(624*2^m) moves a generator 2^m blocks forward).
// With w_0.. the raw word sequence started by src_state,  dst[i] = XOR_{j : g_j = 1} w[i + j].
// One 640-thread workgroup per jump: 32 further blocks of the sequence are generated into LDS
// (33 * 624 words = 82 KB), then thread i folds the ~10^4 taps of g for output word i
// (consecutive threads read consecutive LDS words: conflict free).
// Block b jumps states[b * src_stride] -> states[b * src_stride + dst_offset] (in place allowed).
// ---------------------------------------------------------------------------------------------
constexpr int MT_SEQ_BLOCKS = 33;
constexpr int MT_POLY_WORDS = 312;

// `parts` > 1: the taps of g are split over `parts` workgroups per jump (grid = jumps * parts; the early levels of the
// jump tree have one, two, four ... jumps and would otherwise leave the GPU idle for ~0.4 ms each); every part folds
// its slice of the polynomial into tmp[jump][624] with atomicXor (tmp zeroed), mt_jump_store_kernel moves the result
// into place.  parts == 1: the result is written directly.
// Round 6: the taps of g are first unpacked into a LIST of sequence offsets in LDS (uint16, ascending, padded to a multiple of
// 16 with the offset of a zeroed block behind the sequence), then every output word folds 16 taps per trip: sixteen independent
// LDS reads in flight instead of one bit-scan + one dependent read per tap (mt_jump_v1_kernel below: ~87 cycles per tap at 2.5
// wavefronts per SIMD -- the LDS round trip -- i.e. ~0.36 ms per jump, 3.3 ms for the eleven tree levels of an RMAT-22 pass).
// `parts` > 1 slices the tap LIST (16-tap granules), not the polynomial's words.
constexpr int MT_TAP_CAP = 19968 + 16;               // >= 19937 taps rounded up to 16, + one granule read ahead
constexpr int MT_ZERO_OFF = MT_SEQ_BLOCKS * 624;     // seq[MT_ZERO_OFF + t] = 0 for t < 624: what a padding tap reads
__global__ void __launch_bounds__(640)
mt_jump_kernel(uint32_t *__restrict__ states, const uint64_t *__restrict__ poly, uint32_t src_stride,
               uint32_t dst_offset, uint32_t parts, uint32_t *tmp) {
    __shared__ uint32_t seq[(MT_SEQ_BLOCKS + 1) * 624];
    __shared__ __attribute__((aligned(16))) uint16_t taps[MT_TAP_CAP];
    __shared__ uint32_t wcnt[MT_POLY_WORDS];
    const int t = threadIdx.x;
    const uint32_t jump = blockIdx.x / parts, part = blockIdx.x % parts;
    const uint64_t src = (uint64_t)jump * src_stride;
    uint64_t word = 0;
    if (t < 624) { seq[t] = states[src * 624 + t]; seq[MT_ZERO_OFF + t] = 0u; }
    if (t < MT_POLY_WORDS) {
        word = poly[t];
        if (t == MT_POLY_WORDS - 1) word &= (1ull << (19937 - 64 * (MT_POLY_WORDS - 1))) - 1ull;   // (degree < 19937)
        wcnt[t] = (uint32_t)__popcll(word);
    }
    __syncthreads();
    uint32_t total = 0, pos = 0;
    for (int k = 0; k < MT_POLY_WORDS; k++) {   // (uniform reads: 312 broadcasts)
        const uint32_t c = wcnt[k];
        if (k < t) pos += c;
        total += c;
    }
    if (t < MT_POLY_WORDS)
        while (word) {
            const int b = __builtin_ctzll(word);
            word &= word - 1;
            taps[pos++] = (uint16_t)(t * 64 + b);
        }
    const uint32_t padded = (total + 15u) & ~15u;
    if ((uint32_t)t < padded - total) taps[total + (uint32_t)t] = (uint16_t)MT_ZERO_OFF;
    __syncthreads();
    // this part's slice of the list, in granules of 16 taps
    const uint32_t gran = padded / 16u;
    const uint32_t g0 = (uint32_t)((uint64_t)gran * part / parts), g1 = (uint32_t)((uint64_t)gran * (part + 1u) / parts);
    const uint32_t k0 = g0 * 16u, k1 = g1 * 16u;
    // it reads sequence words up to (its last real tap) + 623
    int need = 1;
    if (k1 > k0) {
        const uint32_t last = k1 <= total ? taps[k1 - 1u] : (total > k0 ? taps[total - 1u] : 0u);
        need = (int)((last + 623u) / 624u) + 1;
    }
    if (need > MT_SEQ_BLOCKS) need = MT_SEQ_BLOCKS;
    for (int blk = 1; blk < need; blk++) {
        uint32_t *nw = seq + blk * 624;
        const uint32_t *od = nw - 624;
        if (t < 227) nw[t] = mt_mix_dev(od[t], od[t + 1], od[t + 397]);
        __syncthreads();
        if (t >= 227 && t < 454) nw[t] = mt_mix_dev(od[t], od[t + 1], nw[t - 227]);
        __syncthreads();
        if (t >= 454 && t < 624) nw[t] = mt_mix_dev(od[t], (t < 623) ? od[t + 1] : nw[0], nw[t - 227]);
        __syncthreads();
    }
    if (t < 624) {
        const uint32_t *sb = seq + t;
        uint32_t acc0 = 0, acc1 = 0;
        uint4 qa = *(const uint4 *)(taps + k0), qb = *(const uint4 *)(taps + k0 + 8u);
        for (uint32_t k = k0; k < k1; k += 16u) {
            // (the next trip's offsets are requested before this trip's words: k + 16 <= MT_TAP_CAP - 16 lies inside the array)
            const uint4 na = *(const uint4 *)(taps + k + 16u), nb = *(const uint4 *)(taps + k + 24u);
            const uint32_t x0 = sb[qa.x & 0xffffu], x1 = sb[qa.x >> 16], x2 = sb[qa.y & 0xffffu], x3 = sb[qa.y >> 16];
            const uint32_t x4 = sb[qa.z & 0xffffu], x5 = sb[qa.z >> 16], x6 = sb[qa.w & 0xffffu], x7 = sb[qa.w >> 16];
            const uint32_t y0 = sb[qb.x & 0xffffu], y1 = sb[qb.x >> 16], y2 = sb[qb.y & 0xffffu], y3 = sb[qb.y >> 16];
            const uint32_t y4 = sb[qb.z & 0xffffu], y5 = sb[qb.z >> 16], y6 = sb[qb.w & 0xffffu], y7 = sb[qb.w >> 16];
            acc0 ^= (x0 ^ x1) ^ (x2 ^ x3) ^ (x4 ^ x5) ^ (x6 ^ x7);
            acc1 ^= (y0 ^ y1) ^ (y2 ^ y3) ^ (y4 ^ y5) ^ (y6 ^ y7);
            qa = na; qb = nb;
        }
        const uint32_t acc = acc0 ^ acc1;
        if (parts == 1) states[(src + dst_offset) * 624 + t] = acc;
        else if (acc) atomicXor(&tmp[(uint64_t)jump * 624 + t], acc);
    }
}

// (the first form: one bit scan and one dependent LDS read per tap; PECANPY_AMD_MT_JUMP_V1=1, kept for same-box comparisons)
__global__ void __launch_bounds__(640)
mt_jump_v1_kernel(uint32_t *__restrict__ states, const uint64_t *__restrict__ poly, uint32_t src_stride,
               uint32_t dst_offset, uint32_t parts, uint32_t *tmp) {
    __shared__ uint32_t seq[MT_SEQ_BLOCKS * 624];
    __shared__ uint64_t spoly[MT_POLY_WORDS];
    const int t = threadIdx.x;
    const uint32_t jump = blockIdx.x / parts, part = blockIdx.x % parts;
    const uint64_t src = (uint64_t)jump * src_stride;
    const int per = (MT_POLY_WORDS + (int)parts - 1) / (int)parts;
    const int jw0 = (int)part * per, jw1 = jw0 + per < MT_POLY_WORDS ? jw0 + per : MT_POLY_WORDS;
    if (t < 624) seq[t] = states[src * 624 + t];
    if (t < MT_POLY_WORDS) spoly[t] = poly[t];
    __syncthreads();
    // this part reads sequence words up to index (jw1 - 1) * 64 + 63 + 623
    int need = jw1 > jw0 ? ((jw1 - 1) * 64 + 63 + 623) / 624 + 1 : 1;
    if (need > MT_SEQ_BLOCKS) need = MT_SEQ_BLOCKS;
    for (int blk = 1; blk < need; blk++) {
        uint32_t *nw = seq + blk * 624;
        const uint32_t *od = nw - 624;
        if (t < 227) nw[t] = mt_mix_dev(od[t], od[t + 1], od[t + 397]);
        __syncthreads();
        if (t >= 227 && t < 454) nw[t] = mt_mix_dev(od[t], od[t + 1], nw[t - 227]);
        __syncthreads();
        if (t >= 454 && t < 624) nw[t] = mt_mix_dev(od[t], (t < 623) ? od[t + 1] : nw[0], nw[t - 227]);
        __syncthreads();
    }
    if (t < 624) {
        uint32_t acc = 0;
        for (int jw = jw0; jw < jw1; jw++) {
            uint64_t word = spoly[jw];
            const uint32_t *base = seq + t + jw * 64;
            while (word) {
                int b = __builtin_ctzll(word);
                word &= word - 1;
                acc ^= base[b];
            }
        }
        if (parts == 1) states[(src + dst_offset) * 624 + t] = acc;
        else if (acc) atomicXor(&tmp[(uint64_t)jump * 624 + t], acc);
    }
}

__global__ void __launch_bounds__(640)
mt_jump_store_kernel(uint32_t *__restrict__ states, uint32_t *tmp, uint32_t src_stride, uint32_t dst_offset) {
    const int t = threadIdx.x;
    if (t >= 624) return;
    const uint64_t src = (uint64_t)blockIdx.x * src_stride;
    states[(src + dst_offset) * 624 + t] = tmp[(uint64_t)blockIdx.x * 624 + t];
    tmp[(uint64_t)blockIdx.x * 624 + t] = 0;   // zero again for the next level
}

// ---------------------------------------------------------------------------------------------
// Per-row membership filters (walk_sparse.hip.h: filter_hash / filter_word / filter_bits) and the adjacency index
// (adj_hash / adj_lookup): one thread per CSR entry e = (u -> v) sets the two filter bits of v in row u's filter,
// writes the key stream entry kf[e], and inserts (v, position of e in row u) into row u's open-addressing table.
// (One wavefront per ROW, round 2: a quarter of the lanes busy at the average degree and a 64-bit division per
// entry -- 50 + 58 ms at RMAT-22.)  frac = floor(indptr[v] * 2^32 / nnz) through a float64 product: any monotone map
// of indptr[v] onto [0, 2^32) serves (only this kernel computes it; the walk kernels read it back from kf).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
membership_build_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                        const uint32_t *__restrict__ edge_row, const uint32_t *__restrict__ foff,
                        const uint64_t *__restrict__ tab_off, unsigned long long *fbits, uint2 *kf,
                        unsigned long long *slots, uint32_t nnz, double frac_scale) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const uint32_t u = edge_row[e], v = indices[e];
    const uint32_t f0 = foff[u];
    const uint32_t nw_mask = foff[u + 1] - f0 - 1u;
    double fr = (double)indptr[v] * frac_scale;   // frac_scale = 2^32 / nnz: fr < 2^32 (indptr[v] < nnz for a row that is somebody's neighbour ... or == nnz: clamp)
    if (fr > 4294967295.0) fr = 4294967295.0;
    const uint32_t fw = filter_fw((uint32_t)fr, v);
    kf[e] = make_uint2(v, fw);
    atomicOr(&fbits[f0 + filter_word(fw, nw_mask)], (unsigned long long)filter_bits(fw));
    const uint64_t off = tab_off[u];
    const uint32_t smask = (uint32_t)(tab_off[u + 1] - off) - 1u;
    const unsigned long long entry = ((unsigned long long)((uint32_t)e - indptr[u]) << 32) | v;
    uint32_t idx = adj_hash(v, smask);
    for (;;) {
        const unsigned long long old = atomicCAS(&slots[off + idx], (unsigned long long)SLOT_EMPTY, entry);
        if (old == (unsigned long long)SLOT_EMPTY) break;
        idx = (idx + 1) & smask;
    }
}

// Input checks of pw_csr_create, one lane per CSR entry (the reference's isnotin needs
// ascending, duplicate-free rows, src/pecanpy/rw/sparse_rw.py:142-230, and reads wherever an index points):
//   flags[0] = first entry whose column index is >= n_nodes      (atomicMin, ~0 = none)
//   flags[1] = first entry not strictly greater than its predecessor in the same row
//   flags[2] = 1 when some weight differs from 1.0f  (data == nullptr: all weights are 1)
//   flags[3] = 1 when the graph has a self loop
__global__ void __launch_bounds__(256)
csr_validate_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                    const float *__restrict__ data, const uint32_t *__restrict__ edge_row, uint32_t n_nodes,
                    uint32_t nnz, unsigned long long *flags) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const uint32_t u = edge_row[e], v = indices[e];
    if (v >= n_nodes) atomicMin(&flags[0], (unsigned long long)e);
    if (e > indptr[u] && indices[e - 1] >= v) atomicMin(&flags[1], (unsigned long long)e);
    if (data && data[e] != 1.0f) flags[2] = 1ull;
    if (data && !(data[e] >= 0.0f && data[e] <= 3.0e38f)) flags[4] = 1ull;   // negative, NaN or infinite weight
    if (u == v) flags[3] = 1ull;
}

// vrec[v] = { indptr[v], degree(v), foff[v], tab_off[v] / 2 } for v in [0, n_nodes]  (walk_sparse.hip.h: CsrDev::vrec)
__global__ void __launch_bounds__(256)
vrec_build_kernel(const uint32_t *__restrict__ indptr, const uint32_t *__restrict__ foff,
                  const uint64_t *__restrict__ tab_off, uint32_t n_nodes, uint4 *vrec) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > n_nodes) return;
    vrec[v] = make_uint4(indptr[v], v < n_nodes ? indptr[v + 1] - indptr[v] : 0u, foff[v], (uint32_t)(tab_off[v] >> 1));
}

// flag = index of the first start vertex that is not a vertex of the graph (atomicMin, ~0 = none)
__global__ void __launch_bounds__(256)
starts_check_kernel(const uint32_t *__restrict__ starts, uint64_t n_jobs, uint32_t n_nodes, unsigned long long *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_jobs && starts[i] >= n_nodes) atomicMin(flag, (unsigned long long)i);
}

__global__ void csr_edge_rows_kernel(const uint32_t *__restrict__ indptr, uint32_t n_nodes, uint32_t *edge_row) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const uint32_t n_waves = (gridDim.x * blockDim.x) / 64;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t v = wave; v < n_nodes; v += n_waves)
        for (uint32_t e = indptr[v] + lane; e < indptr[v + 1]; e += 64) edge_row[e] = v;
}

// ---------------------------------------------------------------------------------------------
// Stream offsets.  draws[i] = number of doubles job i consumes.  First pass assumes every walk
// from a start with neighbours runs its full length (always true on undirected graphs); repair
// passes use the lengths the walk kernel actually produced (out[i][L+1] - 1).
// Three small kernels = exclusive prefix sum over n_jobs 64-bit counts.
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// (indptr: the has-neighbours BITMAP of the graph, has_nbr_bits_kernel below)
__device__ __forceinline__ uint64_t job_draws(const uint32_t *indptr, const uint32_t *starts,
                                              const uint32_t *walks, uint32_t L, uint64_t i) {
    if (walks) return (uint64_t)walks[i * ((uint64_t)L + 2) + L + 1] - 1;
    uint32_t v = starts[i];
    return (indptr[v >> 5] >> (v & 31u)) & 1u ? (uint64_t)L : 0ull;
}

// bit v of has[] = vertex v has neighbours (512 KB at RMAT-22: cache resident, where the two scattered indptr reads per job of
// rounds 1-4 were not -- the offsets scan of a 41.9 M-job array 1.9 -> 0.5 ms)
__global__ void __launch_bounds__(256)
has_nbr_bits_kernel(const uint32_t *__restrict__ indptr, uint32_t n_nodes, uint32_t *has) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= (n_nodes + 31u) / 32u) return;
    uint32_t bits = 0;
    for (uint32_t b = 0; b < 32u; b++) {
        const uint32_t v = w * 32u + b;
        if (v < n_nodes && indptr[v] != indptr[v + 1]) bits |= 1u << b;
    }
    has[w] = bits;
}

__device__ __forceinline__ uint64_t block_reduce_u64(uint64_t v, uint64_t *sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int s = SCAN_BLOCK / 2; s > 0; s >>= 1) {
        if (t < s) sh[t] += sh[t + s];
        __syncthreads();
    }
    uint64_t r = sh[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(SCAN_BLOCK)
draws_tile_sums_kernel(const uint32_t *indptr, const uint32_t *starts, const uint32_t *walks,
                       uint32_t L, uint64_t n_jobs, uint64_t *tile_sums) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n_jobs) s += job_draws(indptr, starts, walks, L, base + k);
    s = block_reduce_u64(s, sh);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = s;
}

// single block: exclusive scan of tile sums in place; total goes to tile_sums[n_tiles]
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_tile_sums_kernel(uint64_t *tile_sums, uint64_t n_tiles) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    __shared__ uint64_t carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_tiles; base += SCAN_BLOCK) {
        uint64_t v = base + t < n_tiles ? tile_sums[base + t] : 0;
        sh[t] = v;
        __syncthreads();
        for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
            uint64_t add = t >= off ? sh[t - off] : 0;
            __syncthreads();
            sh[t] += add;
            __syncthreads();
        }
        uint64_t incl = sh[t];
        uint64_t c = carry;
        if (base + t < n_tiles) tile_sums[base + t] = c + incl - v;
        __syncthreads();
        if (t == SCAN_BLOCK - 1) carry = c + incl;
        __syncthreads();
    }
    if (t == 0) tile_sums[n_tiles] = carry;
}

// stream_off[i] = skip + exclusive prefix of draws.  If changed_list != nullptr, jobs whose offset
// changed are appended to it (repair passes re-run exactly those) and *first_mismatch (optional) receives the smallest such
// job index.  Block-wise repair (pw_simulate_device, sink-heavy directed graphs): the arrays are those of a WINDOW of the
// job array (starts / walks / stream_off advanced to its first job, n_jobs = its length, skip = the stream offset of its
// first job); `job_base` = index of that first job, added to what is reported.
__global__ void __launch_bounds__(SCAN_BLOCK)
draws_offsets_kernel(const uint32_t *indptr, const uint32_t *starts, const uint32_t *walks,
                     uint32_t L, uint64_t n_jobs, const uint64_t *tile_sums, uint64_t skip,
                     uint64_t *stream_off, uint32_t *changed_list,
                     unsigned long long *changed_count, uint64_t job_base, unsigned long long *first_mismatch) {
    __shared__ uint64_t sh[SCAN_BLOCK];
    const int t = threadIdx.x;
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)t * SCAN_ITEMS;
    uint64_t loc[SCAN_ITEMS];
    uint64_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        loc[k] = base + k < n_jobs ? job_draws(indptr, starts, walks, L, base + k) : 0;
        s += loc[k];
    }
    sh[t] = s;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        uint64_t add = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    uint64_t run = skip + tile_sums[blockIdx.x] + sh[t] - s;
    uint64_t first = ~0ull;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        uint64_t i = base + k;
        if (i < n_jobs) {
            if (changed_list) {
                if (stream_off[i] != run && loc[k] != 0) {
                    if (job_base + i < first) first = job_base + i;
                    unsigned long long slot = atomicAdd(changed_count, 1ull);
                    changed_list[slot] = (uint32_t)(job_base + i);
                }
            }
            stream_off[i] = run;
        }
        run += loc[k];
    }
    if (first_mismatch) {   // (one atomic per tile at most)
        __syncthreads();
        sh[t] = first;
        __syncthreads();
        for (int s2 = SCAN_BLOCK / 2; s2 > 0; s2 >>= 1) {
            if (t < s2 && sh[t + s2] < sh[t]) sh[t] = sh[t + s2];
            __syncthreads();
        }
        if (t == 0 && sh[0] != ~0ull) atomicMin(first_mismatch, (unsigned long long)sh[0]);
    }
}

}  // namespace pw
