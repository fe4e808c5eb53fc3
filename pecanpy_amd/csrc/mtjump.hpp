// mtjump.hpp -- host-side MT19937 state service: seeding, block twist and polynomial jump-ahead.
//
// The reference draws every random number of a seeded run from ONE MT19937 stream
// (np.random.seed(random_state) in Base._random_walks, reference src/pecanpy/pecanpy.py:177-178;
// np.random.random() per step, pecanpy.py:557/609).  To address that stream at an arbitrary
// offset (multi-GPU shards, parallel expansion on the device) we need the generator state after
// J single-word steps.  MT19937 is a linear recurrence over GF(2) with a degree-19937
// characteristic polynomial phi, so   state_J = g(A) state_0   with  g(x) = x^J mod phi(x)
// (A = one-word state transition).  phi is derived at first use with Berlekamp-Massey from the
// generator's own output bits -- no magic constants.
#pragma once
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace pw {

constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int MT_DEG = 19937;
constexpr int MT_PW = (MT_DEG + 64) / 64;  // words of a polynomial with degree <= 19937 (312)

inline void mt_seed_state(uint32_t *mt, uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < MT_N; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

inline uint32_t mt_mix(uint32_t hi, uint32_t lo, uint32_t far) {
    uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// one full block: words w[n..n+623] -> w[n+624..n+1247], in place
inline void mt_twist_block(uint32_t *mt) {
    for (int i = 0; i < MT_N; i++) mt[i] = mt_mix(mt[i], mt[(i + 1) % MT_N], mt[(i + MT_M) % MT_N]);
}

inline uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

struct MtPoly {
    uint64_t w[2 * MT_PW + 2];
    void clear() { memset(w, 0, sizeof(w)); }
    bool get(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
    void flip(int i) { w[i >> 6] ^= (uint64_t)1 << (i & 63); }
};

class MtJump {
  public:
    static MtJump &instance() {
        static MtJump j;
        return j;
    }

    // g(x) = x^J mod phi(x)
    void pow_x(uint64_t J, MtPoly &g) {
        ensure_phi();
        g.clear();
        g.flip(0);
        int top = 63;
        while (top >= 0 && !((J >> top) & 1)) top--;
        MtPoly sq;
        for (int b = top; b >= 0; b--) {
            // g = g^2
            sq.clear();
            for (int wi = 0; wi < MT_PW; wi++) {
                uint64_t v = g.w[wi];
                while (v) {
                    int bit = __builtin_ctzll(v);
                    v &= v - 1;
                    sq.flip(2 * (wi * 64 + bit));
                }
            }
            reduce(sq);
            g = sq;
            if ((J >> b) & 1) {  // g = g * x
                uint64_t carry = 0;
                for (int wi = 0; wi <= MT_PW; wi++) {
                    uint64_t nc = g.w[wi] >> 63;
                    g.w[wi] = (g.w[wi] << 1) | carry;
                    carry = nc;
                }
                if (g.get(MT_DEG)) {
                    for (int t : phi_terms_) g.flip(t);
                }
            }
        }
    }

    // state <- g(A) state   (Horner; state = 624-word window of the raw word sequence)
    void apply(const MtPoly &g, uint32_t *state) {
        // circular buffer of 624 words; "advance" produces the next raw word
        uint32_t acc[MT_N];
        memset(acc, 0, sizeof(acc));
        int head = 0;  // acc window starts at acc[head]
        for (int i = MT_DEG - 1; i >= 0; i--) {
            // acc <- A * acc  (generate one word, drop the oldest)
            uint32_t nw = mt_mix(acc[head], acc[(head + 1) % MT_N], acc[(head + MT_M) % MT_N]);
            acc[head] = nw;
            head = (head + 1) % MT_N;
            if (g.get(i)) {
                for (int k = 0; k < MT_N; k++) acc[(head + k) % MT_N] ^= state[k];
            }
        }
        for (int k = 0; k < MT_N; k++) state[k] = acc[(head + k) % MT_N];
    }

    // pre-twist state of block b (raw words w[624 b .. 624 b + 623]) for `seed`
    void state_at_block(uint32_t seed, uint64_t block, uint32_t *state) {
        mt_seed_state(state, seed);
        if (block == 0) return;
        if (block <= 64) {
            for (uint64_t b = 0; b < block; b++) mt_twist_block(state);
            return;
        }
        MtPoly g;
        pow_x(block * (uint64_t)MT_N, g);
        apply(g, state);
    }

    const std::vector<int> &phi_terms() {
        ensure_phi();
        return phi_terms_;
    }

    // Table of jump polynomials T[m] = x^(624 * 2^m) mod phi, m = 0..MAX_POW2 (MT_PW words each):
    // T[m] moves a generator forward by 2^m blocks.  Seed independent; built once per process.
    static constexpr int MAX_POW2 = 44;
    const uint64_t *pow2_table() {
        std::call_once(table_once_, [this] {
            ensure_phi();
            table_.assign((size_t)(MAX_POW2 + 1) * MT_PW, 0);
            MtPoly g;
            pow_x((uint64_t)MT_N, g);
            for (int m = 0; m <= MAX_POW2; m++) {
                memcpy(&table_[(size_t)m * MT_PW], g.w, sizeof(uint64_t) * MT_PW);
                if (m < MAX_POW2) square(g);
            }
        });
        return table_.data();
    }

    // g <- g^2 mod phi
    void square(MtPoly &g) {
        ensure_phi();
        MtPoly sq;
        sq.clear();
        for (int wi = 0; wi < MT_PW; wi++) {
            uint64_t v = g.w[wi];
            while (v) {
                int bit = __builtin_ctzll(v);
                v &= v - 1;
                sq.flip(2 * (wi * 64 + bit));
            }
        }
        reduce(sq);
        g = sq;
    }

  private:
    std::once_flag once_, table_once_;
    std::vector<uint64_t> table_;
    std::vector<int> phi_terms_;  // exponents with a non-zero coefficient, excluding nothing

    void ensure_phi() { std::call_once(once_, [this] { derive_phi(); }); }

    // reduce a polynomial of degree < 2*MT_DEG modulo phi
    void reduce(MtPoly &a) {
        for (int i = 2 * MT_DEG - 2; i >= MT_DEG; i--) {
            if (a.get(i)) {
                int sh = i - MT_DEG;
                for (int t : phi_terms_) a.flip(t + sh);
            }
        }
    }

    // Berlekamp-Massey over GF(2) on one output bit of the generator.
    void derive_phi() {
        const int NB = 2 * MT_DEG + 64;
        const int WORDS = (NB + 63) / 64 + 1;
        std::vector<uint64_t> s(WORDS, 0);
        {
            uint32_t st[MT_N];
            mt_seed_state(st, 5489u);
            int pos = MT_N;
            for (int n = 0; n < NB; n++) {
                if (pos == MT_N) { mt_twist_block(st); pos = 0; }
                uint32_t y = mt_temper(st[pos++]);
                if (y & 1u) s[n >> 6] |= (uint64_t)1 << (n & 63);
            }
        }
        // C(x), B(x) as bitsets; standard BM with discrepancy computed by AND+popcount on a
        // reversed window of the sequence.
        const int PW2 = (MT_DEG + 2 + 63) / 64 + 1;
        std::vector<uint64_t> Cp(PW2, 0), Bp(PW2, 0), Tp(PW2, 0);
        Cp[0] = 1;
        Bp[0] = 1;
        int Lc = 0, m = 1;
        // reversed sequence buffer: rev bit j at step n holds s[n - j]
        std::vector<uint64_t> rev(PW2 + 1, 0);
        for (int n = 0; n < NB; n++) {
            // shift rev left by one and insert s[n] at bit 0
            uint64_t carry = (s[n >> 6] >> (n & 63)) & 1;
            for (int wi = 0; wi < PW2; wi++) {
                uint64_t nc = rev[wi] >> 63;
                rev[wi] = (rev[wi] << 1) | carry;
                carry = nc;
            }
            int nw = Lc / 64 + 1;
            uint64_t acc = 0;
            for (int wi = 0; wi < nw && wi < PW2; wi++) acc ^= Cp[wi] & rev[wi];
            int dsc = __builtin_parityll(acc);
            if (dsc == 0) {
                m++;
            } else if (2 * Lc <= n) {
                Tp = Cp;
                xor_shifted(Cp, Bp, m);
                Lc = n + 1 - Lc;
                Bp = Tp;
                m = 1;
            } else {
                xor_shifted(Cp, Bp, m);
                m++;
            }
        }
        // Lc must be 19937; phi(x) = x^L * C(1/x)  =>  coefficient of x^(L-i) is c_i
        phi_terms_.clear();
        for (int i = 0; i <= Lc; i++)
            if ((Cp[i >> 6] >> (i & 63)) & 1) phi_terms_.push_back(Lc - i);
        degree_ = Lc;
    }

    static void xor_shifted(std::vector<uint64_t> &dst, const std::vector<uint64_t> &src, int sh) {
        int ws = sh >> 6, bs = sh & 63;
        int n = (int)dst.size();
        for (int wi = n - 1; wi >= ws; wi--) {
            uint64_t v = src[wi - ws] << bs;
            if (bs && wi - ws - 1 >= 0) v |= src[wi - ws - 1] >> (64 - bs);
            dst[wi] ^= v;
        }
    }

  public:
    int degree_ = 0;
};

// doubles #offset.. of the seeded stream, starting from a jumped state (host, sequential after
// the jump).  Used for pw_mt_random_sample and for cross-checking the device expansion.
inline void mt_random_sample_host(uint32_t seed, uint64_t offset, uint64_t n, double *out) {
    uint64_t word0 = 2 * offset;
    uint64_t block = word0 / MT_N;
    int pos = (int)(word0 % MT_N);
    uint32_t st[MT_N];
    MtJump::instance().state_at_block(seed, block, st);
    mt_twist_block(st);
    for (uint64_t i = 0; i < n; i++) {
        uint32_t ab[2];
        for (int k = 0; k < 2; k++) {
            if (pos == MT_N) { mt_twist_block(st); pos = 0; }
            ab[k] = mt_temper(st[pos++]);
        }
        out[i] = ((double)(ab[0] >> 5) * 67108864.0 + (double)(ab[1] >> 6)) / 9007199254740992.0;
    }
}

}  // namespace pw
