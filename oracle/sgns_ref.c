/* sgns_ref.c -- CPU restatement of skip-gram with negative sampling (SGNS), the model the reference trains on its
 * walks: gensim Word2Vec(walks, vector_size, window, sg=1, min_count=0, workers, epochs)
 * (src/pecanpy/pecanpy.py:276-290, src/pecanpy/cli.py:307-325).  TEST INFRASTRUCTURE ONLY: nothing under pecanpy_amd/
 * links or calls this file; tests/ compare the HIP trainer (csrc/sgns.hip.h) with it.
 *
 * Third-party algorithm.  The arithmetic lives in gensim (requirements.txt:2 pins gensim==4.3.2; not under
 * /root/reference, not installed here), whose sg/negative path is Mikolov's word2vec.c (TrainModelThread, the `else`
 * branch of `if (cbow)`), restated here from the published algorithm:
 *   - vocabulary = every node id; unigram^0.75 noise distribution; frequent-word subsampling with
 *     keep(w) = (sqrt(f_w / (sample * T)) + 1) * (sample * T) / f_w   (word2vec.c: `ran`; gensim: sample_int);
 *   - a sentence (= one walk) is first thinned by that subsampling, THEN windows are taken over what is left;
 *   - per centre position a window of width window - b, b = rand % window;
 *   - for every context word c in it: input vector syn0[c]; targets = the centre (label 1) and `negative` noise words
 *     (label 0, a draw equal to the centre is skipped); f = syn0[c] . syn1[target];
 *     g = (label - sigma(f)) * lr, sigma saturated to 0 / 1 beyond |f| > 6 (word2vec.c MAX_EXP);
 *     neu1e += g * syn1[target]; syn1[target] += g * syn0[c]; afterwards syn0[c] += neu1e;
 *   - learning rate decaying linearly from alpha to min_alpha over all (epochs x occurrences);
 *   - syn0 initialised uniformly in (-0.5, 0.5) / dim, syn1 with zeros.
 * What is NOT gensim's: its random streams (NumPy + a 48-bit LCG whose use depends on job batching and threads) --
 * every random choice here is a hash of (seed, epoch, walk, position[, context, draw]), the same function the HIP
 * trainer evaluates, so that a SINGLE-WAVEFRONT run of the trainer visits the same updates in the same order and its
 * vectors can be compared with these within float tolerance (dot products are summed in a different order there).
 * gensim's result is not reproducible across thread counts either (hogwild); parity with it is statistical by nature.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
static float u24(uint64_t x) { return (float)(x >> 40) * (1.0f / 16777216.0f); }

/* vectors: float32[n_nodes * dim] (syn0 after training); loss (may be NULL): mean SGNS loss of the pairs trained in the
 * last epoch, evaluated before each update.  Returns 0, or -1 on a malformed walk matrix / allocation failure. */
EXPORT int sgns_ref_train(const uint32_t *walks, uint64_t n_walks, uint32_t L, uint32_t n_nodes, uint32_t dim,
                          uint32_t window, uint32_t negative, uint32_t epochs, float alpha, float min_alpha, float sample,
                          uint32_t seed, float *vectors, double *loss) {
    const uint32_t W = L + 2;
    const uint64_t n_items = n_walks * (uint64_t)(L + 1);
    uint64_t *cnt = calloc(n_nodes, sizeof(uint64_t));
    float *syn1 = calloc((size_t)n_nodes * dim, sizeof(float));
    float *keep = sample > 0 ? malloc(sizeof(float) * n_nodes) : NULL;
    float *neu1e = malloc(sizeof(float) * dim);
    if (!cnt || !syn1 || !neu1e || (sample > 0 && !keep)) return -1;
    /* vocabulary statistics */
    for (uint64_t wk = 0; wk < n_walks; wk++) {
        const uint32_t *row = walks + wk * W;
        if (row[L + 1] > L + 1) return -1;
        for (uint32_t p = 0; p < row[L + 1]; p++) {
            if (row[p] >= n_nodes) return -1;
            cnt[row[p]]++;
        }
    }
    double total = 0, pow_total = 0;
    for (uint32_t i = 0; i < n_nodes; i++) { total += (double)cnt[i]; pow_total += pow((double)cnt[i], 0.75); }
    if (!(total > 0)) return -1;
    /* noise table: word w owns a share pow(cnt, 0.75) / pow_total of the slots (word2vec.c InitUnigramTable) */
    uint64_t ts = 16ull * n_nodes;
    if (ts < (1ull << 16)) ts = 1ull << 16;
    if (ts > (1ull << 26)) ts = 1ull << 26;
    const uint32_t table_size = (uint32_t)ts;
    uint32_t *table = malloc(sizeof(uint32_t) * table_size);
    if (!table) return -1;
    {
        uint32_t t = 0, last = 0;
        double cum = 0;
        for (uint32_t w = 0; w < n_nodes; w++) {
            if (!cnt[w]) continue;
            last = w;
            cum += pow((double)cnt[w], 0.75) / pow_total;
            while (t < table_size && (double)(t + 1) / table_size <= cum) table[t++] = w;
        }
        while (t < table_size) table[t++] = last;   /* rounding of the last share */
    }
    if (keep) {
        const double thr = (double)sample * total;
        for (uint32_t i = 0; i < n_nodes; i++) {
            double k = cnt[i] ? (sqrt((double)cnt[i] / thr) + 1.0) * thr / (double)cnt[i] : 1.0;
            keep[i] = (float)(k < 1.0 ? k : 1.0);
        }
    }
    /* syn0 ~ U(-0.5, 0.5) / dim */
    {
        uint64_t x = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed << 17);
        for (size_t i = 0; i < (size_t)n_nodes * dim; i++) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            vectors[i] = (((float)((x >> 40) & 0xffffff) / 16777216.0f) - 0.5f) / (float)dim;
        }
    }
    const uint64_t item_total = n_items * epochs;
    double loss_sum = 0;
    uint64_t loss_n = 0;
    for (uint32_t ep = 0; ep < epochs; ep++) {
        const uint64_t item_base = n_items * ep;
        const int last_epoch = ep + 1 == epochs;
        for (uint64_t wk = 0; wk < n_walks; wk++) {
            const uint32_t *row = walks + wk * W;
            const uint32_t len = row[L + 1];
#define OCC(p) mix64((uint64_t)seed ^ (item_base + wk * (uint64_t)(L + 1) + (p)) * 0x9E3779B97F4A7C15ull)
#define KEPT(p) (!keep || u24(OCC(p)) < keep[row[p]])
            for (uint32_t pos = 0; pos < len; pos++) {
                if (!KEPT(pos)) continue;
                const uint64_t item = wk * (uint64_t)(L + 1) + pos;
                uint64_t rs = mix64(OCC(pos));
                const uint32_t eff = window - (uint32_t)(rs % window);
                float lr = alpha - (alpha - min_alpha) * (float)((double)(item_base + item) / (double)item_total);
                if (lr < min_alpha) lr = min_alpha;
                /* the window over the thinned sentence: up to eff kept positions on either side */
                uint32_t lo = pos, hi = pos, got = 0;
                for (uint32_t c = pos; c-- > 0 && got < eff;) if (KEPT(c)) { lo = c; got++; }
                got = 0;
                for (uint32_t c = pos + 1; c < len && got < eff; c++) if (KEPT(c)) { hi = c; got++; }
                const uint32_t centre = row[pos];
                for (uint32_t c = lo; c <= hi; c++) {
                    if (c == pos || !KEPT(c)) continue;
                    const uint32_t ctx = row[c];
                    rs = mix64(rs + c);
                    float *v = vectors + (size_t)ctx * dim;
                    memset(neu1e, 0, sizeof(float) * dim);
                    for (uint32_t ng = 0; ng <= negative; ng++) {
                        uint32_t target = centre;
                        if (ng) {
                            rs = mix64(rs + ng);
                            target = table[(uint32_t)(rs >> 16) % table_size];
                            if (target == centre) continue;
                        }
                        float *u = syn1 + (size_t)target * dim;
                        /* the dot product in the order the 64-lane trainer sums it: component k on lane k % 64, then a
                         * butterfly over the lanes (a float sum is not associative; everything else is elementwise) */
                        float part[64], tmp[64];
                        for (uint32_t ln = 0; ln < 64; ln++) {
                            part[ln] = 0.0f;
                            for (uint32_t k = ln; k < dim; k += 64) part[ln] += v[k] * u[k];
                        }
                        for (uint32_t off = 32; off > 0; off >>= 1) {
                            for (uint32_t ln = 0; ln < 64; ln++) tmp[ln] = part[ln] + part[ln ^ off];
                            memcpy(part, tmp, sizeof(part));
                        }
                        const float dot = part[0];
                        const float sig = dot > 6.0f ? 1.0f : (dot < -6.0f ? 0.0f : 1.0f / (1.0f + expf(-dot)));
                        if (last_epoch && loss) {
                            const double s = 1.0 / (1.0 + exp(-(double)dot));
                            loss_sum += ng == 0 ? -log(s > 1e-12 ? s : 1e-12) : -log(1.0 - s > 1e-12 ? 1.0 - s : 1e-12);
                            loss_n++;
                        }
                        const float g = ((ng == 0 ? 1.0f : 0.0f) - sig) * lr;
                        for (uint32_t k = 0; k < dim; k++) {
                            neu1e[k] += g * u[k];
                            u[k] += g * v[k];
                        }
                    }
                    for (uint32_t k = 0; k < dim; k++) v[k] += neu1e[k];
                }
            }
#undef OCC
#undef KEPT
        }
    }
    if (loss) *loss = loss_n ? loss_sum / (double)loss_n : 0.0;
    free(cnt); free(syn1); free(keep); free(neu1e); free(table);
    return 0;
}
