// slk_mtjump.hip -- host-side GF(2) polynomial arithmetic for MT19937 jump-ahead.
//
// The reference draws every negative from ONE serial MT19937 stream (spotlight/sampling.py:34
// through numpy's RandomState).  To generate that same stream on many CUs at once, workgroup w
// (k_mt_jump, slk_rng.hip) must find the state block 624*(w*L) words ahead.  The MT19937
// transition T is linear over GF(2) on its 19937-bit state with characteristic polynomial phi;
// if g_m(x) = x^(624 m - 1) mod phi then, for the raw word stream x_0, x_1, ... of the current
// state,
//        x[624 m + j] = XOR over { i : coefficient i of g_m is 1 } of x[1 + i + j],  j = 0..623
// (every bit sequence of (x_n), n >= 1, obeys the recurrence phi; n >= 1 because the low 31
// bits of x_0 are not part of the state).  This file computes phi by Berlekamp-Massey on
// 2*19937 bits of an MT19937 stream and the table g_{L}, g_{2L}, ... by square-and-multiply;
// the kernel evaluates the XOR-convolution from a 33-block prefix held in LDS.
// Restates no reference code: the reference has no parallel sampler.
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "slk_common.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

const int DEG = 19937;
const int NW = 312;  // 64-bit words holding DEG+1 bits (19968)

typedef std::vector<uint64_t> poly;  // little-endian bit order: coefficient i at bit i

inline bool getbit(const uint64_t *a, int i) { return (a[i >> 6] >> (i & 63)) & 1ull; }
inline void flipbit(uint64_t *a, int i) { a[i >> 6] ^= 1ull << (i & 63); }

// raw MT19937 words x_0.. from init_genrand(seed) (numpy/random/src/mt19937/mt19937.c)
void mt_words(uint32_t seed, std::vector<uint32_t> &x, size_t n) {
    x.resize(n);
    x[0] = seed;
    for (int i = 1; i < 624; ++i) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
    for (size_t k = 0; k + 624 < n; ++k) {
        const uint32_t y = (x[k] & 0x80000000u) | (x[k + 1] & 0x7fffffffu);
        x[k + 624] = x[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

// Berlekamp-Massey over GF(2): connection polynomial C (C_0 = 1) of s[0..n), returns L.
int berlekamp_massey(const std::vector<uint8_t> &s, poly &C) {
    const int n = (int)s.size();
    const int W = n / 64 + 2;
    poly B(W, 0), T(W, 0), win(W, 0);  // win bit i = s[cur - 1 - i]
    C.assign(W, 0);
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int cur = 0; cur < n; ++cur) {
        // discrepancy d = s[cur] + sum_{i=1..L} C_i s[cur-i] = s[cur] + parity(C>>1 & win)
        uint64_t acc = 0;
        const int lw = L / 64 + 1;
        for (int w = 0; w < lw; ++w) {
            const uint64_t c = (C[w] >> 1) | (C[w + 1] << 63);
            acc ^= c & win[w];
        }
        const int d = (int)(s[cur] ^ (__builtin_popcountll(acc) & 1));
        if (d) {
            const bool grow = 2 * L <= cur;
            if (grow) T = C;
            // C ^= B << m
            const int ws = m >> 6, bs = m & 63;
            for (int w = W - 1 - ws; w >= 0; --w) {
                uint64_t v = B[w] << bs;
                if (bs && w > 0) v |= B[w - 1] >> (64 - bs);
                C[w + ws] ^= v;
            }
            if (grow) {
                L = cur + 1 - L;
                B.swap(T);
                m = 1;
            } else {
                ++m;
            }
        } else {
            ++m;
        }
        // slide the window: win = (win << 1) | s[cur]
        for (int w = W - 1; w > 0; --w) win[w] = (win[w] << 1) | (win[w - 1] >> 63);
        win[0] = (win[0] << 1) | (uint64_t)s[cur];
    }
    return L;
}

struct Field {
    std::vector<int> terms;  // exponents t < DEG with phi_t = 1 (x^DEG == sum x^t)

    // a has 2*NW words (degree < 2*DEG); reduce modulo phi in place (result in bits [0,DEG)).
    void reduce(uint64_t *a) const {
        for (int k = 2 * DEG - 2; k >= DEG; --k) {
            if (!getbit(a, k)) continue;
            flipbit(a, k);
            const int base = k - DEG;
            for (int t : terms) flipbit(a, base + t);
        }
    }
};

#if defined(__x86_64__)
__attribute__((target("pclmul,sse2"))) void clmul_hw(uint64_t a, uint64_t b, uint64_t *lo, uint64_t *hi) {
    const __m128i r = _mm_clmulepi64_si128(_mm_set_epi64x(0, (long long)a), _mm_set_epi64x(0, (long long)b), 0);
    *lo = (uint64_t)_mm_cvtsi128_si64(r);
    *hi = (uint64_t)_mm_cvtsi128_si64(_mm_srli_si128(r, 8));
}
#endif

void clmul_sw(uint64_t a, uint64_t b, uint64_t *lo, uint64_t *hi) {
    uint64_t l = 0, h = 0;
    for (int i = 0; i < 64; ++i)
        if ((a >> i) & 1ull) {
            l ^= b << i;
            if (i) h ^= b >> (64 - i);
        }
    *lo = l;
    *hi = h;
}

typedef void (*clmul_fn)(uint64_t, uint64_t, uint64_t *, uint64_t *);

clmul_fn pick_clmul() {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("pclmul")) return clmul_hw;
#endif
    return clmul_sw;
}

// c (NW words) = a * b mod phi
void mulmod(const Field &F, clmul_fn cm, const uint64_t *a, const uint64_t *b, uint64_t *c) {
    std::vector<uint64_t> p(2 * NW + 1, 0);
    for (int i = 0; i < NW; ++i) {
        if (!a[i]) continue;
        for (int j = 0; j < NW; ++j) {
            if (!b[j]) continue;
            uint64_t lo, hi;
            cm(a[i], b[j], &lo, &hi);
            p[i + j] ^= lo;
            p[i + j + 1] ^= hi;
        }
    }
    F.reduce(p.data());
    memcpy(c, p.data(), NW * 8);
}

// r = x^e mod phi
void xpow(const Field &F, clmul_fn cm, uint64_t e, uint64_t *r) {
    std::vector<uint64_t> acc(NW, 0), tmp(2 * NW + 1, 0);
    acc[0] = 1;
    for (int b = 63; b >= 0; --b) {
        std::vector<uint64_t> sq(NW);
        mulmod(F, cm, acc.data(), acc.data(), sq.data());
        acc = sq;
        if ((e >> b) & 1ull) {  // acc *= x
            std::fill(tmp.begin(), tmp.end(), 0);
            for (int w = 0; w < NW; ++w) {
                tmp[w] |= acc[w] << 1;
                tmp[w + 1] |= acc[w] >> 63;
            }
            F.reduce(tmp.data());
            memcpy(acc.data(), tmp.data(), NW * 8);
        }
    }
    memcpy(r, acc.data(), NW * 8);
}

std::mutex g_mu;
// per level: [SLK_MT_JUMP_WG - 1][SLK_MT_JUMP_TERMS] exponents of the set coefficients of g_w, padded with SLK_MT_JUMP_PAD
std::vector<uint32_t> g_table[SLK_MT_JUMP_LEVELS];
bool g_ok[SLK_MT_JUMP_LEVELS] = {false};
Field g_field;
bool g_field_ok = false;

}  // namespace

// Host table of jump polynomials g_w = x^(624 * L * w - 1) mod phi, w = 1 .. SLK_MT_JUMP_WG-1, L = slk_mt_jump_blocks(level)
// state blocks per stream, stored as exponent lists (~10k set coefficients each, padded to SLK_MT_JUMP_TERMS with
// SLK_MT_JUMP_PAD, which the kernel maps onto a zeroed LDS block); computed once per process and level (~0.1-1 s).
const uint32_t *slk_mt_jump_table(slk_ctx *ctx, int level) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (level < 0 || level >= SLK_MT_JUMP_LEVELS) return nullptr;
    if (g_ok[level]) return g_table[level].data();

    if (!g_field_ok) {
        // phi from 2*DEG bits of the stream (bit 0 of x_{n+1})
        std::vector<uint32_t> x;
        mt_words(5489u, x, 2 * DEG + 700);
        std::vector<uint8_t> s(2 * DEG);
        for (int n = 0; n < 2 * DEG; ++n) s[n] = (uint8_t)(x[n + 1] & 1u);
        poly C;
        const int L = berlekamp_massey(s, C);
        if (L != DEG) {
            slk_fail(ctx, SLK_EIO, "MT19937 minimal polynomial has degree %d, expected %d", L, DEG);
            return nullptr;
        }
        // characteristic polynomial phi(x) = x^L C(1/x): phi_{L-i} = C_i
        for (int i = 1; i <= DEG; ++i)
            if (getbit(C.data(), i)) g_field.terms.push_back(DEG - i);
        g_field_ok = true;
    }
    const Field &F = g_field;
    const clmul_fn cm = pick_clmul();

    const uint64_t stride = 624ull * (uint64_t)slk_mt_jump_blocks(level);
    std::vector<uint64_t> h(NW), g(NW), nxt(NW);
    xpow(F, cm, stride, h.data());      // x^(624 L)
    xpow(F, cm, stride - 1, g.data());  // x^(624 L - 1)

    // self-check against the stream itself: x[624 L + j] == XOR_i g_i x[1 + i + j]
    {
        std::vector<uint32_t> y;
        mt_words(5489u, y, (size_t)stride + 700);
        for (int j = 0; j < 3; ++j) {
            uint32_t acc = 0;
            for (int i = 0; i < DEG; ++i)
                if (getbit(g.data(), i)) acc ^= y[1 + i + j];
            if (acc != y[stride + j]) {
                slk_fail(ctx, SLK_EIO, "MT19937 jump polynomial self-check failed");
                return nullptr;
            }
        }
    }
    std::vector<uint32_t> &tab = g_table[level];
    tab.assign((size_t)(SLK_MT_JUMP_WG - 1) * SLK_MT_JUMP_TERMS, (uint32_t)SLK_MT_JUMP_PAD);
    for (int w = 1; w < SLK_MT_JUMP_WG; ++w) {
        uint32_t *dst = tab.data() + (size_t)(w - 1) * SLK_MT_JUMP_TERMS;
        int n = 0;
        for (int i = 0; i < DEG; ++i)
            if (getbit(g.data(), i)) {
                if (n >= SLK_MT_JUMP_TERMS - 48) {
                    slk_fail(ctx, SLK_EIO, "MT19937 jump polynomial has more than %d terms", SLK_MT_JUMP_TERMS);
                    return nullptr;
                }
                dst[n++] = (uint32_t)i;
            }
        dst[SLK_MT_JUMP_TERMS - 1] = (uint32_t)((n + 15) / 16 * 16);  // rounded length: the kernel takes the list 8 at a time
        if (w + 1 < SLK_MT_JUMP_WG) {
            mulmod(F, cm, g.data(), h.data(), nxt.data());
            g = nxt;
        }
    }
    g_ok[level] = true;
    return tab.data();
}
