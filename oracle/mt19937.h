/*
 * oracle/mt19937.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * MT19937 exactly as the reference consumes it through Numba's thread-local generator:
 *   - twist / tempering constants : numba:_random.c:13-56, numba:cpython/randomimpl.py:108-131
 *   - seeding (init_genrand)      : numba:_random.c:58-73 (np.random.seed inside njit,
 *                                   reference call site src/pecanpy/pecanpy.py:177-178)
 *   - random()/rand() 53-bit      : numba:cpython/randomimpl.py:133-146 (call sites pecanpy.py:424,557,609,674)
 *   - randint masked rejection    : numba:cpython/randomimpl.py:148-187,320-387 (call sites pecanpy.py:307,673)
 * The same stream is NumPy's legacy RandomState(seed) stream.
 */
#ifndef PECAN_ORACLE_MT19937_H
#define PECAN_ORACLE_MT19937_H

#include <stdint.h>

typedef struct {
    uint32_t mt[624];
    int idx;
} orc_mt_t;

static inline void orc_mt_seed(orc_mt_t *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

static inline void orc_mt_twist(orc_mt_t *s) {
    uint32_t *mt = s->mt;
    for (int i = 0; i < 624; i++) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        uint32_t v = mt[(i + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        mt[i] = v;
    }
    s->idx = 0;
}

static inline uint32_t orc_mt_next32(orc_mt_t *s) {
    if (s->idx >= 624) orc_mt_twist(s);
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* np.random.random() / np.random.rand(): two words -> 53-bit double in [0,1). */
static inline double orc_mt_random(orc_mt_t *s) {
    uint32_t a = orc_mt_next32(s) >> 5;
    uint32_t b = orc_mt_next32(s) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

/* np.random.randint(n) inside njit ("np" flavour): n == 1 -> 0 words; otherwise keep the
 * low bit_length(n-1) bits of one word and reject while >= n. */
static inline uint64_t orc_mt_randint(orc_mt_t *s, uint64_t n) {
    if (n == 1) return 0;
    uint64_t nm1 = n - 1;
    int nbits = 0;
    while (nm1) { nbits++; nm1 >>= 1; }
    uint32_t mask = (nbits >= 32) ? 0xffffffffu : ((1u << nbits) - 1u);
    for (;;) {
        uint32_t r = orc_mt_next32(s) & mask;
        if ((uint64_t)r < n) return r;
    }
}

#endif
