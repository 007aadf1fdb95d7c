// slk_rng.hip -- on-GPU, bit-exact numpy legacy RandomState.randint(0, num_items, ...).
//
// Replaces spotlight/sampling.py:8-36 (sample_items) + the per-minibatch H2D copy at
// spotlight/factorization/implicit.py:256-260.  numpy draws each output by masked
// rejection over the raw MT19937 32-bit stream: v = next32() & mask until v <= num_items-1.
// The k-th accepted word is the k-th output, so the whole draw is
//     raw stream  ->  temper  ->  mask/accept  ->  order-preserving stream compaction.
//   k_mt_prefix / k_mt_jump / k_mt_stream   the raw stream on up to 256 CUs at once: GF(2) jump-ahead
//                   to one start block per stream, then one wavefront per stream advances the 624-word
//                   state block by block and streams the UNTEMPERED blocks to HBM;
//   k_accept_count / k_scan_counts / k_accept_scatter   all CUs temper, test and compact;
//   k_rng_finalize  restores (key, pos) to exactly what numpy would hold after the draw:
//                   the state block containing the last CONSUMED word, pos = offset + 1.
#include <math.h>
#include <stddef.h>

#include "slk_common.h"

#define SLK_MT_N 624
#define SLK_TILE 2048  // words per block in the compaction kernels (256 threads x 8)

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// ---- the generator: three launches per draw (round 5; rounds 1-4: one kernel whose workgroups regenerated the prefix,
// jumped and then advanced block by block with ten waves and three s_barriers per block -- every barrier's release fence
// also waited for the block's global stores, 1.1 us per block, 0.22 ms per 8.8 M words).
//   k_mt_prefix   ONE wavefront: the 33-block prefix of the stream every jump is evaluated from (the twist has 227-way
//                 parallelism and a single wave needs no s_barrier: a workgroup of 64 threads synchronises by program order)
//   k_mt_jump     workgroup w (1 .. W-1): the state block 624 * (w L) words ahead, x[624 m + j] = XOR_{i in g_m} x[1 + i + j]
//                 (slk_mtjump.hip).  6.2 M word-XORs per jump: the kernel is bound by the instructions it issues per XOR, not by
//                 bytes (measured, profiles/r05_*: one output word per lane and a dword read per term -- rounds 1-4 -- spends
//                 three VALU instructions per XOR on unpacking the exponent and forming the address).  So a lane owns TWO
//                 adjacent output words and a term costs it one address add, one ds_read_b64 (256 B/clk/CU against
//                 ds_read_b32's 128) and two XORs: the prefix is held twice, once shifted by a word, so that every window is
//                 8-byte aligned, in two passes over the exponent range (10 240 exponents each: half the prefix per pass); the
//                 term list is staged as ready-made byte offsets; the terms of a pass are split between three groups of five
//                 waves (four waves per SIMD), whose partial sums meet in LDS.  ~90 us for the 255 jumps of a draw
//   k_mt_stream   workgroup w = ONE wavefront: blocks [w L, (w+1) L) from its start block, streamed UNTEMPERED to HBM
#define SLK_MT_JUMP_GROUPS 3      // term groups of five waves; lane l < 312 of a group owns the output words 2l and 2l + 1
#define SLK_MT_JUMP_GROUP 320
#define SLK_MT_JUMP_THREADS (SLK_MT_JUMP_GROUPS * SLK_MT_JUMP_GROUP)
#define SLK_MT_PREFIX_BLOCKS 33  // 33*624 = 20592 >= 1 + 19936 + 624 words feed the jump
#define SLK_MT_PREFIX_WORDS (SLK_MT_PREFIX_BLOCKS * SLK_MT_N)
#define SLK_MT_WIN 10240                  // exponents per pass (two passes: 20480 > 19937)
#define SLK_MT_WINW (SLK_MT_WIN + 640)    // words per window copy: index (a - wb) + 2l + 1 <= WIN + 639
// LDS of k_mt_jump: window copy A | copy B (shifted by one word) | zero block [640] (target of the padding terms) |
// the pass's term codes (byte offsets, uint32) | the partial sums of the groups behind the first [x 640]
#define SLK_MT_LDS_WORDS (2 * SLK_MT_WINW + 640 + SLK_MT_JUMP_TERMS + (SLK_MT_JUMP_GROUPS - 1) * 640 + 4)

// One regeneration by ONE wavefront: n[0..624) = next state block of o[0..624) (both in LDS, SLK_MT_PAD words each), the
// new words also to dst[0..624) in HBM from the registers they were formed in.  The twist
// x[k+624] = f(x[k], x[k+1], x[k+397]) has 227-way parallelism: words [0,227) need only old words, [227,454) the first
// round, [454,624) the second -- three dependent LDS round trips per block.  Every round computes whole 64-lane rows (256,
// 256, 192 words): the words past a round's range are garbage that the next round overwrites in LDS before anything valid
// reads them (the rows' reads and writes stay inside the padded buffers); only the stores to HBM are predicated.
// SLK_WAVE_SYNC orders the wave's own LDS writes before its later reads (no s_barrier, no wait for the stores in flight).
#define SLK_MT_PAD 704
__device__ __forceinline__ void mt_regen_wave(const uint32_t *o, uint32_t *n, int lane, uint32_t *dst) {
    {
        uint32_t v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = lane + 64 * r;
            v[r] = mt_twist(o[i], o[i + 1], o[i + 397]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = lane + 64 * r;
            n[i] = v[r];
            if (r < 3 || i < 227) dst[i] = v[r];
        }
    }
    SLK_WAVE_SYNC();
    {
        uint32_t v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 227 + lane + 64 * r;
            v[r] = mt_twist(o[i], o[i + 1], n[i - 227]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 227 + lane + 64 * r;
            n[i] = v[r];
            if (r < 3 || i < 454) dst[i] = v[r];
        }
    }
    SLK_WAVE_SYNC();
    {
        uint32_t v[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = 454 + lane + 64 * r;
            const uint32_t nx = (i == SLK_MT_N - 1) ? n[0] : o[i + 1];
            v[r] = mt_twist(o[i], nx, n[i - 227]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = 454 + lane + 64 * r;
            n[i] = v[r];
            if (r < 2 || i < SLK_MT_N) dst[i] = v[r];
        }
    }
    SLK_WAVE_SYNC();
}

// pre[b*624 ..] = state block b of the stream that starts with key_src, b = 0 .. 32
__global__ __launch_bounds__(64) void k_mt_prefix(const uint32_t *key_src, uint32_t *pre) {
    __shared__ uint32_t pp[2][SLK_MT_PAD];
    const int lane = threadIdx.x;
    for (int i = lane; i < SLK_MT_PAD; i += 64) {
        const uint32_t v = i < SLK_MT_N ? key_src[i] : 0u;
        pp[0][i] = v;
        pp[1][i] = 0u;
        if (i < SLK_MT_N) pre[i] = v;
    }
    SLK_WAVE_SYNC();
    for (int b = 1; b < SLK_MT_PREFIX_BLOCKS; ++b) mt_regen_wave(pp[(b - 1) & 1], pp[b & 1], lane, pre + (size_t)b * SLK_MT_N);
}

// start[(w-1)*624 ..] = state block w*L of the stream whose first 33 blocks are pre[], w = blockIdx.x + 1; polys row w - 1 is
// the ascending exponent list of g_{w L} (wave-uniform), padded with SLK_MT_JUMP_PAD.
__global__ __launch_bounds__(SLK_MT_JUMP_THREADS) void k_mt_jump(const uint32_t *pre, const uint32_t *polys, uint32_t *start) {
    // NO static __shared__ in this kernel: eight bytes of it in front of the dynamic region took the region off its 16-byte
    // alignment and every ds_read_b128 of the codes was replayed at ~50 clocks -- the convolution ran 0.40 ms instead of 0.09
    // (profiles/r05_f_sampler_jump_variants.jsonl, r05_h_*; cdna_hip_programming.md, guideline 17).  The two counters live
    // behind the partial sums.
    HIP_DYNAMIC_SHARED(uint32_t, lds)
    uint32_t *A = lds, *B = lds + SLK_MT_WINW, *Z = lds + 2 * SLK_MT_WINW;
    uint32_t *cx = Z + 640;
    uint32_t *comb = cx + SLK_MT_JUMP_TERMS;
    uint32_t *s_cnt = comb + (SLK_MT_JUMP_GROUPS - 1) * 640;
    const int t = threadIdx.x, g = t / SLK_MT_JUMP_GROUP, l = t - g * SLK_MT_JUMP_GROUP;
    const uint32_t *e = polys + (size_t)blockIdx.x * SLK_MT_JUMP_TERMS;
    const int nterms = (int)e[SLK_MT_JUMP_TERMS - 1];
    if (t < 2) s_cnt[t] = 0u;
    if (t < 640) Z[t] = 0u;
    __syncthreads();
    {
        uint32_t c0 = 0, c1 = 0;
        for (int i = t; i < nterms; i += SLK_MT_JUMP_THREADS) {
            const uint32_t x = e[i];
            c0 += x < (uint32_t)SLK_MT_WIN ? 1u : 0u;
            c1 += x != (uint32_t)SLK_MT_JUMP_PAD ? 1u : 0u;
        }
        if (c0) atomicAdd(&s_cnt[0], c0);
        if (c1) atomicAdd(&s_cnt[1], c1);
    }
    __syncthreads();
    const int split = (int)s_cnt[0], nreal = (int)s_cnt[1];
    uint32_t a0 = 0, a1 = 0;
    for (int p = 0; p < 2; ++p) {
        const int lo = p ? split : 0, T = (p ? nreal : split) - lo, Tpad = (T + 7) & ~7;
        const uint32_t wb = (uint32_t)p * SLK_MT_WIN;
        if (p) __syncthreads();
        for (int k = t; k < SLK_MT_WINW / 4; k += SLK_MT_JUMP_THREADS) {
            const uint32_t idx = wb + 4u * (uint32_t)k;
            reinterpret_cast<uint4 *>(A)[k] = idx < (uint32_t)SLK_MT_PREFIX_WORDS ? *reinterpret_cast<const uint4 *>(pre + idx)
                                                                                 : make_uint4(0u, 0u, 0u, 0u);
        }
        for (int k = t; k < SLK_MT_WINW; k += SLK_MT_JUMP_THREADS) {
            const uint32_t idx = wb + (uint32_t)k + 1u;
            B[k] = idx < (uint32_t)SLK_MT_PREFIX_WORDS ? pre[idx] : 0u;
        }
        // term code = LDS byte offset of the window of output word 0: exponent i, a = 1 + i: x[a + 2l], x[a + 2l + 1] is the
        // aligned pair A[(a - wb) + 2l] when a is even, B[(a - 1 - wb) + 2l] when it is odd; padding terms read the zero block
        for (int i = t; i < Tpad; i += SLK_MT_JUMP_THREADS) {
            uint32_t c = 2u * SLK_MT_WINW;
            if (i < T) {
                const uint32_t a = e[lo + i] + 1u;
                c = ((a & 1u) ? (uint32_t)SLK_MT_WINW : 0u) + ((a - wb) & ~1u);
            }
            cx[i] = 4u * c;
        }
        __syncthreads();
        // the codes are wave-uniform: four per ds_read_b128 (same address in every lane: a broadcast), each the offset of one
        // ds_read_b64 of the lane's two words.  (A/B, profiles/r05_h_*: a code per lane by ds_read_b32 + v_readlane is 3 % slower.)
        const char *xl = reinterpret_cast<const char *>(lds + 2 * l);
        for (int k = g * 8; k < Tpad; k += 8 * SLK_MT_JUMP_GROUPS) {
            const uint4 c0 = *reinterpret_cast<const uint4 *>(cx + k), c1 = *reinterpret_cast<const uint4 *>(cx + k + 4);  // eight codes
#define SLK_MT_LD(off_) (*reinterpret_cast<const uint2 *>(xl + (off_)))
            const uint2 r0 = SLK_MT_LD(c0.x), r1 = SLK_MT_LD(c0.y), r2 = SLK_MT_LD(c0.z), r3 = SLK_MT_LD(c0.w);
            const uint2 r4 = SLK_MT_LD(c1.x), r5 = SLK_MT_LD(c1.y), r6 = SLK_MT_LD(c1.z), r7 = SLK_MT_LD(c1.w);
#undef SLK_MT_LD
            a0 ^= ((r0.x ^ r1.x) ^ (r2.x ^ r3.x)) ^ ((r4.x ^ r5.x) ^ (r6.x ^ r7.x));
            a1 ^= ((r0.y ^ r1.y) ^ (r2.y ^ r3.y)) ^ ((r4.y ^ r5.y) ^ (r6.y ^ r7.y));
        }
    }
    if (g > 0) {
        comb[(g - 1) * 640 + 2 * l] = a0;
        comb[(g - 1) * 640 + 2 * l + 1] = a1;
    }
    __syncthreads();
    if (g == 0 && 2 * l < SLK_MT_N) {
        uint32_t *dst = start + (size_t)blockIdx.x * SLK_MT_N;
        for (int q = 0; q < SLK_MT_JUMP_GROUPS - 1; ++q) {
            a0 ^= comb[q * 640 + 2 * l];
            a1 ^= comb[q * 640 + 2 * l + 1];
        }
        dst[2 * l] = a0;
        dst[2 * l + 1] = a1;
    }
}

// raw[b*624 ..] = state block b (untempered), b = 0 .. nblocks-1, block 0 = key_src itself.  Workgroup w (one wavefront)
// owns blocks [w*L, (w+1)*L): it starts from key_src (w = 0) or from the block k_mt_jump left in start[] and advances
// block by block through two LDS buffers; the blocks leave as coalesced stores nothing in the loop waits for.
__global__ __launch_bounds__(64) void k_mt_stream(const uint32_t *key_src, const uint32_t *start, uint32_t *raw, int nblocks, int L) {
    __shared__ uint32_t pp[2][SLK_MT_PAD];
    const int lane = threadIdx.x;
    const int first = (int)blockIdx.x * L;
    if (first >= nblocks) return;
    const int last = (first + L < nblocks) ? first + L : nblocks;
    const uint32_t *src = blockIdx.x == 0 ? key_src : start + (size_t)(blockIdx.x - 1) * SLK_MT_N;
    uint32_t *dst = raw + (size_t)first * SLK_MT_N;
    for (int i = lane; i < SLK_MT_PAD; i += 64) {
        const uint32_t v = i < SLK_MT_N ? src[i] : 0u;
        pp[0][i] = v;
        pp[1][i] = 0u;
        if (i < SLK_MT_N) dst[i] = v;
    }
    SLK_WAVE_SYNC();
    int cur = 0;
    for (int b = first + 1; b < last; ++b) {
        mt_regen_wave(pp[cur], pp[cur ^ 1], lane, raw + (size_t)b * SLK_MT_N);
        cur ^= 1;
    }
}

struct slk_accept_args {
    const uint32_t *raw;
    const slk_rng_dev *st;
    unsigned long long total_words;  // words generated (nblocks * 624)
    uint32_t mask, rng;
};

// The eight words of thread t of tile b (stream order: word = tile * 2048 + t * 8 + j), tempered and masked; ok bit j set
// when word j is part of the draw (at or behind the stream position, inside the generated range, value accepted).
__device__ __forceinline__ unsigned accept_words8(const slk_accept_args &a, unsigned long long base, int pos0, uint32_t *v) {
    uint32_t w[8];
    if (base + 8 <= a.total_words) {
        const uint4 lo = *reinterpret_cast<const uint4 *>(a.raw + base), hi = *reinterpret_cast<const uint4 *>(a.raw + base + 4);
        w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w;
        w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = base + j < a.total_words ? a.raw[base + j] : 0u;
    }
    unsigned ok = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = mt_temper(w[j]) & a.mask;
        const unsigned long long t = base + j;
        if (t >= (unsigned long long)pos0 && t < a.total_words && v[j] <= a.rng) ok |= 1u << j;
    }
    return ok;
}

// exclusive scan of one value per thread over a workgroup of 256 threads; *total = the workgroup's sum
__device__ __forceinline__ uint32_t accept_excl_scan_256(uint32_t v, uint32_t *s_wsum, uint32_t *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wsum[w] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < w) off += s_wsum[i];
        tot += s_wsum[i];
    }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

__global__ __launch_bounds__(256) void k_accept_count(slk_accept_args a, uint32_t *cnt) {
    __shared__ uint32_t s_w[4];
    const int pos0 = a.st->pos;
    const unsigned long long base = (unsigned long long)blockIdx.x * SLK_TILE + (unsigned long long)threadIdx.x * 8;
    uint32_t v[8];
    uint32_t c = (uint32_t)__popc(accept_words8(a, base, pos0, v));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of cnt[nb] -> tile offsets (64-bit); total -> st->accepted.  One workgroup: thread t adds up its
// contiguous share of the tiles, one scan over the 256 shares, then every thread writes its tiles' offsets.
__global__ __launch_bounds__(256) void k_scan_counts(const uint32_t *cnt, unsigned long long *off, int nb,
                                                     slk_rng_dev *st) {
    __shared__ unsigned long long s[256];
    const int t = threadIdx.x;
    const int per = (nb + 255) / 256;
    const int i0 = t * per < nb ? t * per : nb, i1 = i0 + per < nb ? i0 + per : nb;
    unsigned long long sum = 0;
    for (int i = i0; i < i1; ++i) sum += cnt[i];
    s[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned long long x = (t >= d) ? s[t - d] : 0ull;
        __syncthreads();
        s[t] += x;
        __syncthreads();
    }
    unsigned long long run = s[t] - sum;
    for (int i = i0; i < i1; ++i) {
        off[i] = run;
        run += cnt[i];
    }
    if (t == 255) st->accepted = s[255];
}

__global__ __launch_bounds__(256) void k_accept_scatter(slk_accept_args a, const unsigned long long *off,
                                                        unsigned long long count, uint32_t *out32,
                                                        int64_t *out64, slk_rng_dev *st) {
    __shared__ uint32_t s_w[4];
    const int t = threadIdx.x;
    const int pos0 = a.st->pos;
    const unsigned long long base = (unsigned long long)blockIdx.x * SLK_TILE + (unsigned long long)t * 8;
    uint32_t v[8];
    const unsigned ok = accept_words8(a, base, pos0, v);
    uint32_t tot;
    unsigned long long rank = off[blockIdx.x] + (unsigned long long)accept_excl_scan_256((uint32_t)__popc(ok), s_w, &tot);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (!((ok >> j) & 1u)) continue;
        if (rank < count) {
            if (out32) out32[rank] = v[j];
            if (out64) out64[rank] = (int64_t)v[j];
            if (rank == count - 1) st->t_last = base + j;
        }
        ++rank;
    }
}

__global__ __launch_bounds__(256) void k_rng_finalize(slk_rng_dev *st, const uint32_t *raw,
                                                      unsigned long long count) {
    __shared__ int enough;
    if (threadIdx.x == 0) {
        enough = st->accepted >= count;
        if (!enough) st->insufficient = 1;
    }
    __syncthreads();
    if (!enough) return;
    const unsigned long long tl = st->t_last;
    const unsigned long long blk = tl / SLK_MT_N;
    for (int i = threadIdx.x; i < SLK_MT_N; i += 256) st->key[i] = raw[blk * SLK_MT_N + i];
    __syncthreads();
    if (threadIdx.x == 0) st->pos = (int32_t)(tl % SLK_MT_N) + 1;
}

__global__ __launch_bounds__(256) void k_fill_zero(uint32_t *out32, int64_t *out64, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        if (out32) out32[i] = 0;
        if (out64) out64[i] = 0;
    }
}

// ctx->raw[b*624 ..] = state block b (untempered) for b = 0 .. nblocks-1, block 0 = the ctx's
// current key block (ctx->raw must hold nblocks*624 words).
int slk_mt_level_for(const slk_ctx *ctx, unsigned long long nblocks) {
    return nblocks >= (unsigned long long)ctx->opt_mt_long_min_blocks ? 1 : 0;
}

int slk_mt_generate_blocks(slk_ctx *ctx, unsigned long long nblocks, hipStream_t s) {
    uint32_t *raw = (uint32_t *)ctx->raw.p;
    // block 0 of every launch group is its input key block; further groups (> 10.2 M words with 64 blocks per stream, > 40.9 M
    // with 256) restart from the last block of the previous one
    const int level = slk_mt_level_for(ctx, nblocks);
    const unsigned long long L = (unsigned long long)slk_mt_jump_blocks(level);
    const unsigned long long cap = (unsigned long long)SLK_MT_JUMP_WG * L;
    const size_t lds_bytes = (size_t)SLK_MT_LDS_WORDS * 4;
    if (!ctx->mt_attr_set) {  // per ctx, i.e. per device: function attributes live in the device's context
        SLK_HIP(ctx, hipFuncSetAttribute((const void *)k_mt_jump, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        ctx->mt_attr_set = true;
    }
    int rc;
    if (nblocks > L && (rc = slk_mt_jump_reserve(ctx, level))) return rc;
    uint32_t *pre = (uint32_t *)ctx->mt_tmp.p;
    uint32_t *start = pre ? pre + (size_t)SLK_MT_PREFIX_BLOCKS * SLK_MT_N : nullptr;
    unsigned long long first = 0;  // index of the group's block 0
    const uint32_t *key_src = ctx->d_rng->key;
    while (true) {
        const unsigned long long nb_l = (nblocks - first < cap) ? nblocks - first : cap;
        const unsigned wgs = (unsigned)((nb_l + L - 1) / L);
        if (wgs > 1) {
            hipLaunchKernelGGL(k_mt_prefix, dim3(1), dim3(64), 0, s, key_src, pre);
            SLK_LAUNCH_CHECK(ctx, "k_mt_prefix");
            hipLaunchKernelGGL(k_mt_jump, dim3(wgs - 1), dim3(SLK_MT_JUMP_THREADS), lds_bytes, s, (const uint32_t *)pre,
                               (const uint32_t *)ctx->d_jump[level], start);
            SLK_LAUNCH_CHECK(ctx, "k_mt_jump");
        }
        hipLaunchKernelGGL(k_mt_stream, dim3(wgs), dim3(64), 0, s, key_src, (const uint32_t *)start, raw + first * SLK_MT_N, (int)nb_l,
                           (int)L);
        SLK_LAUNCH_CHECK(ctx, "k_mt_stream");
        if (first + nb_l >= nblocks) break;
        first += nb_l - 1;
        key_src = raw + first * SLK_MT_N;
    }
    return SLK_OK;
}

// the jump polynomial table of a stream length class on the device + the generator's scratch (prefix, start blocks):
// allocated once per ctx
int slk_mt_jump_reserve(slk_ctx *ctx, int level) {
    int rc;
    if ((rc = slk_ensure(ctx, ctx->mt_tmp, ((size_t)SLK_MT_PREFIX_BLOCKS + SLK_MT_JUMP_WG) * SLK_MT_N * 4))) return rc;
    if (!ctx->d_jump[level]) {
        const uint32_t *tab = slk_mt_jump_table(ctx, level);
        if (!tab) return SLK_EIO;
        const size_t bytes = (size_t)(SLK_MT_JUMP_WG - 1) * SLK_MT_JUMP_TERMS * 4;
        SLK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_jump[level]), bytes));
        SLK_HIP(ctx, hipMemcpy(ctx->d_jump[level], tab, bytes, hipMemcpyHostToDevice));
    }
    return SLK_OK;
}

int slk_sample_u32(slk_ctx *ctx, int64_t num_items, int64_t count, uint32_t *d_out32, int64_t *d_out64,
                   hipStream_t s) {
    if (num_items < 1 || num_items > (int64_t)1 << 32)
        return slk_fail(ctx, SLK_EINVAL, "slk_sample_items: num_items %lld outside [1, 2^32]", (long long)num_items);
    if (count < 0) return slk_fail(ctx, SLK_EINVAL, "slk_sample_items: negative count");
    if (count == 0) return SLK_OK;
    ctx->last_stream = s;
    slk_prof_begin(ctx, SLK_K_SAMPLE, s);
    if (num_items == 1) {
        // numpy: rng == 0 returns zeros and consumes nothing
        hipLaunchKernelGGL(k_fill_zero, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, d_out32, d_out64,
                           (unsigned long long)count);
        SLK_LAUNCH_CHECK(ctx, "k_fill_zero");
        slk_prof_end(ctx, s);
        return SLK_OK;
    }
    const uint32_t rng = (uint32_t)(num_items - 1);
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    const double p = ((double)rng + 1.0) / ((double)mask + 1.0);
    // words needed: mean count/p plus 12 sigma of the negative-binomial spread plus slack;
    // the first block (current key) may contribute nothing when pos == 624.
    double need = (double)count / p + 12.0 * sqrt((double)count * (1.0 - p)) / p + 64.0;
    if (p == 1.0) need = (double)count;
    const unsigned long long nblocks = 1ull + (unsigned long long)((need + SLK_MT_N - 1) / SLK_MT_N);
    const unsigned long long total_words = nblocks * SLK_MT_N;
    if (nblocks > 0x7fffffffull) return slk_fail(ctx, SLK_EINVAL, "slk_sample_items: count too large for one call");
    const unsigned long long nb = (total_words + SLK_TILE - 1) / SLK_TILE;
    int rc;
    if ((rc = slk_ensure(ctx, ctx->raw, total_words * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->cnt, nb * 4 + nb * 8 + 64))) return rc;
    uint32_t *raw = (uint32_t *)ctx->raw.p;
    unsigned long long *off = (unsigned long long *)ctx->cnt.p;
    uint32_t *cnt = (uint32_t *)(off + nb);

    if ((rc = slk_mt_generate_blocks(ctx, nblocks, s))) return rc;
    slk_accept_args a;
    a.raw = raw;
    a.st = ctx->d_rng;
    a.total_words = total_words;
    a.mask = mask;
    a.rng = rng;
    hipLaunchKernelGGL(k_accept_count, dim3((unsigned)nb), dim3(256), 0, s, a, cnt);
    SLK_LAUNCH_CHECK(ctx, "k_accept_count");
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(256), 0, s, (const uint32_t *)cnt, off, (int)nb, ctx->d_rng);
    SLK_LAUNCH_CHECK(ctx, "k_scan_counts");
    hipLaunchKernelGGL(k_accept_scatter, dim3((unsigned)nb), dim3(256), 0, s, a, (const unsigned long long *)off,
                       (unsigned long long)count, d_out32, d_out64, ctx->d_rng);
    SLK_LAUNCH_CHECK(ctx, "k_accept_scatter");
    hipLaunchKernelGGL(k_rng_finalize, dim3(1), dim3(256), 0, s, ctx->d_rng, (const uint32_t *)raw,
                       (unsigned long long)count);
    SLK_LAUNCH_CHECK(ctx, "k_rng_finalize");
    slk_prof_end(ctx, s);
    // the stream position after this draw is final on the device once this event has fired (slk_rng_get_state_sampled)
    if (!ctx->ev_sampled) SLK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_sampled, hipEventDisableTiming));
    SLK_HIP(ctx, hipEventRecord(ctx->ev_sampled, s));
    ctx->sampled_valid = true;
    return SLK_OK;
}

// scratch of a later slk_sample_u32(num_items, count): same sizing rule as above
int slk_sample_reserve(slk_ctx *ctx, int64_t num_items, int64_t count) {
    if (num_items < 2 || count <= 0) return SLK_OK;
    const uint32_t rng = (uint32_t)(num_items - 1);
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    const double p = ((double)rng + 1.0) / ((double)mask + 1.0);
    double need = (double)count / p + 12.0 * sqrt((double)count * (1.0 - p)) / p + 64.0;
    if (p == 1.0) need = (double)count;
    const unsigned long long nblocks = 1ull + (unsigned long long)((need + SLK_MT_N - 1) / SLK_MT_N);
    const unsigned long long total_words = nblocks * SLK_MT_N;
    const unsigned long long nb = (total_words + SLK_TILE - 1) / SLK_TILE;
    int rc;
    if ((rc = slk_ensure(ctx, ctx->raw, total_words * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->cnt, nb * 4 + nb * 8 + 64))) return rc;
    if (nblocks > (unsigned long long)slk_mt_jump_blocks(slk_mt_level_for(ctx, nblocks)) &&
        (rc = slk_mt_jump_reserve(ctx, slk_mt_level_for(ctx, nblocks))))
        return rc;
    return SLK_OK;
}

SLK_EXPORT int slk_rng_set_state(slk_ctx *ctx, const uint32_t *h_key, int32_t pos) {
    if (!ctx || !h_key) return SLK_EINVAL;
    if (pos < 0 || pos > SLK_MT_N) return slk_fail(ctx, SLK_EINVAL, "slk_rng_set_state: pos %d outside [0, 624]", pos);
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    // the stream the ctx's kernels were last enqueued on -- which may be the null stream (torch's default): a null handle is
    // a stream to wait for, not "no stream" (the state copy below no longer synchronises with it implicitly)
    SLK_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    // (the stream is idle: the sticky flags and the sampler's bookkeeping start afresh, as they always did here)
    ctx->sampled_valid = false;
    ctx->pf.valid = false;
    slk_rng_dev h;
    memset(&h, 0, sizeof(h));
    memcpy(h.key, h_key, sizeof(h.key));
    h.pos = pos;
    hipStream_t cs = slk_copy_stream(ctx);
    SLK_HIP(ctx, hipMemcpyAsync(ctx->d_rng, &h, sizeof(h), hipMemcpyHostToDevice, cs));
    SLK_HIP(ctx, hipStreamSynchronize(cs));
    return SLK_OK;
}

// (key, pos) only: the sticky flags are not touched -- a kernel still running may raise one
int slk_rng_write_state(slk_ctx *ctx, const uint32_t *h_key, int32_t pos) {
    ctx->sampled_valid = false;
    ctx->pf.valid = false;  // a chunk prepared ahead was drawn from the stream this call replaces
    struct {
        uint32_t key[SLK_MT_N];
        int32_t pos;
    } h;
    static_assert(offsetof(slk_rng_dev, pos) == sizeof(uint32_t) * SLK_MT_N, "slk_rng_dev layout");
    memcpy(h.key, h_key, sizeof(h.key));
    h.pos = pos;
    // stream-ordered, not a null-stream hipMemcpy: that one would wait for (and hold up) the work of every other stream of
    // the process -- e.g. the training kernels another ctx has in flight while this one prepares the next epoch
    hipStream_t cs = slk_copy_stream(ctx);
    SLK_HIP(ctx, hipMemcpyAsync(ctx->d_rng, &h, sizeof(h), hipMemcpyHostToDevice, cs));
    SLK_HIP(ctx, hipStreamSynchronize(cs));
    return SLK_OK;
}

static int rng_read_state(slk_ctx *ctx, uint32_t *h_key, int32_t *pos);

SLK_EXPORT int slk_rng_get_state(slk_ctx *ctx, uint32_t *h_key, int32_t *pos) {
    if (!ctx || !h_key || !pos) return SLK_EINVAL;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    // the stream the ctx's kernels were last enqueued on -- which may be the null stream (torch's default): a null handle is
    // a stream to wait for, not "no stream" (the state copy below no longer synchronises with it implicitly)
    SLK_HIP(ctx, hipStreamSynchronize(ctx->last_stream));
    return rng_read_state(ctx, h_key, pos);
}

// The stream position after the LAST DRAW this ctx enqueued (negatives of a training call, slk_sample_items), without waiting
// for the kernels that consume those negatives: a training call draws a chunk's negatives well ahead of its passes, so when
// slk_bilinear_train returns the last draw has completed (the call waits for that chunk's long-run flags, produced behind
// the draw) while up to two chunks of passes are still queued.  The caller (fit()) uses it to start the NEXT epoch's shuffle
// -- which continues the same MT19937 stream, torch_utils.py:46-47 -- on another ctx / stream beside those passes.  Errors
// raised by kernels still running are reported by the next slk_rng_get_state.
SLK_EXPORT int slk_rng_get_state_sampled(slk_ctx *ctx, uint32_t *h_key, int32_t *pos) {
    if (!ctx || !h_key || !pos) return SLK_EINVAL;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->sampled_valid || !ctx->ev_sampled) return slk_rng_get_state(ctx, h_key, pos);  // nothing drawn since the state was set
    SLK_HIP(ctx, hipEventSynchronize(ctx->ev_sampled));
    return rng_read_state(ctx, h_key, pos);
}

static int rng_read_state(slk_ctx *ctx, uint32_t *h_key, int32_t *pos) {
    slk_rng_dev h;
    hipStream_t cs = slk_copy_stream(ctx);
    SLK_HIP(ctx, hipMemcpyAsync(&h, ctx->d_rng, sizeof(h), hipMemcpyDeviceToHost, cs));
    SLK_HIP(ctx, hipStreamSynchronize(cs));
    if (h.insufficient || h.epoch_abort || h.sort_abort) {
        // Reported ONCE: the flags are cleared on the device (they used to stay raised until the next slk_rng_set_state, which
        // a pipelined fit() never issues on its training ctx -- every later epoch then failed).  A barrier time-out also
        // retires the persistent route on this ctx: the per-minibatch launches need no co-residency.
        const int32_t zero2[2] = {0, 0};
        SLK_HIP(ctx, hipMemcpyAsync(&ctx->d_rng->insufficient, zero2, 4, hipMemcpyHostToDevice, cs));
        SLK_HIP(ctx, hipMemcpyAsync(&ctx->d_rng->epoch_abort, zero2 + 1, 4, hipMemcpyHostToDevice, cs));
        SLK_HIP(ctx, hipMemcpyAsync(&ctx->d_rng->sort_abort, zero2, 4, hipMemcpyHostToDevice, cs));
        SLK_HIP(ctx, hipStreamSynchronize(cs));
        if (h.sort_abort)
            return slk_fail(ctx, SLK_EIO,
                            "radix sort abandoned a look-back: a tile never published its digit counts (the device is shared or a "
                            "workgroup was lost).  The tables may hold partial updates: re-initialise the model");
        if (h.epoch_abort) {
            ctx->epoch_refused = true;
            return slk_fail(ctx, SLK_EIO,
                            "persistent epoch kernel abandoned a launch at its grid barrier %d (minibatch %d of that launch): a "
                            "workgroup never arrived -- the device is shared or the grid was not resident.  Minibatches before it "
                            "are applied in full, that one in part: re-initialise the model.  This ctx now uses the per-minibatch "
                            "launches (as with option epoch_kernel=0)",
                            (int)h.epoch_abort, (int)((h.epoch_abort - 1) / (ctx->epoch_bars_per_mb > 0 ? ctx->epoch_bars_per_mb : 2)));
        }
        return slk_fail(ctx, SLK_EIO, "sampler ran out of generated words (rejection tail > 12 sigma)");
    }
    memcpy(h_key, h.key, sizeof(h.key));
    *pos = h.pos;
    return SLK_OK;
}

SLK_EXPORT int slk_sample_items(slk_ctx *ctx, int64_t num_items, int64_t count, int64_t *d_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (!d_out && count > 0) return slk_fail(ctx, SLK_EINVAL, "slk_sample_items: d_out is NULL");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    return slk_sample_u32(ctx, num_items, count, nullptr, d_out, (hipStream_t)stream);
}
