// slk_rng.hip -- on-GPU, bit-exact numpy legacy RandomState.randint(0, num_items, ...).
//
// Replaces spotlight/sampling.py:8-36 (sample_items) + the per-minibatch H2D copy at
// spotlight/factorization/implicit.py:256-260.  numpy draws each output by masked
// rejection over the raw MT19937 32-bit stream: v = next32() & mask until v <= num_items-1.
// The k-th accepted word is the k-th output, so the whole draw is
//     raw stream  ->  temper  ->  mask/accept  ->  order-preserving stream compaction.
//   k_mt_generate   one workgroup advances the 624-word state block by block (the twist has
//                   227-way parallelism: words [0,227) need only old words, [227,454) need
//                   the first round, [454,624) the second) and streams the UNTEMPERED blocks
//                   to HBM;
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

#define SLK_MT_THREADS 640  // 10 waves: one lane per state word in the jump convolution

// One regeneration: n[0..624) = next state block of o[0..624) (both in LDS).  The twist
// x[k+624] = f(x[k], x[k+1], x[k+397]) has 227-way parallelism: words [0,227) need only old
// words, [227,454) need the first round, [454,624) the second.  Ends with a barrier.
__device__ __forceinline__ void mt_regen_block(const uint32_t *o, uint32_t *n, int t) {
    if (t < 227) n[t] = mt_twist(o[t], o[t + 1], o[t + 397]);
    __syncthreads();
    if (t < 227) {
        const int i = t + 227;
        n[i] = mt_twist(o[i], o[i + 1], n[i - 227]);
    }
    __syncthreads();
    if (t < 170) {
        const int i = t + 454;
        const uint32_t nx = (i == SLK_MT_N - 1) ? n[0] : o[i + 1];
        n[i] = mt_twist(o[i], nx, n[i - 227]);
    }
    __syncthreads();
}

#define SLK_MT_PREFIX_BLOCKS 33  // 33*624 = 20592 >= 1 + 19936 + 624 words feed the jump
// LDS: prefix X[33*624] | zero block [624] (target of the padding exponent) | ping | pong |
// the workgroup's exponent list as uint16 (SLK_MT_JUMP_TERMS / 2 words)
#define SLK_MT_LDS_WORDS ((SLK_MT_PREFIX_BLOCKS + 3) * SLK_MT_N + SLK_MT_JUMP_TERMS / 2)

// raw[b*624 ..] = state block b (untempered), b = 0 .. nblocks-1, block 0 = key_src itself.
// Workgroup w owns blocks [w*L, (w+1)*L).  w > 0 first jumps to block w*L:
//   x[624 m + j] = XOR_{i in g_m} x[1 + i + j]   (slk_mtjump.hip), evaluated from a 33-block
// prefix of the stream that every workgroup regenerates for itself in LDS; the exponent list
// of g_m is wave-uniform (scalar loads), lane j XORs one LDS word per term.
__global__ __launch_bounds__(SLK_MT_THREADS) void k_mt_generate_jump(const uint32_t *key_src,
                                                                     const uint32_t *polys, uint32_t *raw,
                                                                     int nblocks) {
    HIP_DYNAMIC_SHARED(uint32_t, lds)
    uint32_t *X = lds;
    uint32_t *zero = lds + SLK_MT_PREFIX_BLOCKS * SLK_MT_N;
    uint32_t *pp0 = zero + SLK_MT_N;
    uint32_t *pp1 = pp0 + SLK_MT_N;
    const int t = threadIdx.x;
    const int first = (int)blockIdx.x * SLK_MT_JUMP_BLOCKS;
    if (first >= nblocks) return;
    const int last = (first + SLK_MT_JUMP_BLOCKS < nblocks) ? first + SLK_MT_JUMP_BLOCKS : nblocks;

    if (t < SLK_MT_N) {
        X[t] = key_src[t];
        zero[t] = 0u;
    }
    const uint32_t *cur = X;
    if (blockIdx.x > 0) {
        __syncthreads();
        for (int b = 1; b < SLK_MT_PREFIX_BLOCKS; ++b)
            mt_regen_block(X + (b - 1) * SLK_MT_N, X + b * SLK_MT_N, t);
        const uint32_t *e = polys + (size_t)(blockIdx.x - 1) * SLK_MT_JUMP_TERMS;
        const uint32_t *xj = X + 1 + (t < SLK_MT_N ? t : 0);
        // The exponent list (every workgroup its own 40 KB: no reuse in the scalar cache, and a
        // dependent scalar load per group of terms cost more than the LDS reads it fed) is staged
        // in LDS as 16-bit values by coalesced vector loads; the loop then reads eight exponents
        // with one broadcast ds_read_b128.  Padding exponents hit the zero block.
        const int nterms = (int)e[SLK_MT_JUMP_TERMS - 1];  // list length rounded up to 16
        uint16_t *sx = reinterpret_cast<uint16_t *>(pp1 + SLK_MT_N);
        for (int i = t; i < nterms; i += SLK_MT_THREADS) sx[i] = (uint16_t)e[i];
        __syncthreads();
        uint32_t acc = 0;
        for (int k = 0; k < nterms; k += 8) {
            const uint4 ev = *reinterpret_cast<const uint4 *>(sx + k);
            uint32_t a0 = xj[ev.x & 0xffffu] ^ xj[ev.y & 0xffffu];
            uint32_t a1 = xj[ev.x >> 16] ^ xj[ev.y >> 16];
            a0 ^= xj[ev.z & 0xffffu] ^ xj[ev.w & 0xffffu];
            a1 ^= xj[ev.z >> 16] ^ xj[ev.w >> 16];
            acc ^= a0 ^ a1;
        }
        if (t < SLK_MT_N) pp0[t] = acc;
        cur = pp0;
    }
    __syncthreads();
    uint32_t *dst = raw + (size_t)first * SLK_MT_N;
    if (t < SLK_MT_N) dst[t] = cur[t];
    for (int b = first + 1; b < last; ++b) {
        uint32_t *nxt = (cur == pp0) ? pp1 : pp0;
        mt_regen_block(cur, nxt, t);
        dst = raw + (size_t)b * SLK_MT_N;
        if (t < SLK_MT_N) dst[t] = nxt[t];
        cur = nxt;
    }
}

struct slk_accept_args {
    const uint32_t *raw;
    const slk_rng_dev *st;
    unsigned long long total_words;  // words generated (nblocks * 624)
    uint32_t mask, rng;
};

__device__ __forceinline__ bool accept_word(const slk_accept_args &a, unsigned long long t, int pos0,
                                            uint32_t *v) {
    if (t < (unsigned long long)pos0 || t >= a.total_words) return false;
    *v = mt_temper(a.raw[t]) & a.mask;
    return *v <= a.rng;
}

__global__ __launch_bounds__(256) void k_accept_count(slk_accept_args a, uint32_t *cnt) {
    __shared__ double red[256];
    const int pos0 = a.st->pos;
    const unsigned long long base = (unsigned long long)blockIdx.x * SLK_TILE + (unsigned long long)threadIdx.x * 8;
    unsigned c = 0;
    for (int j = 0; j < 8; ++j) {
        uint32_t v;
        c += accept_word(a, base + j, pos0, &v) ? 1u : 0u;
    }
    const double tot = slk_block_sum_256((double)c, red);
    if (threadIdx.x == 0) cnt[blockIdx.x] = (uint32_t)tot;
}

// exclusive scan of cnt[nb] in place -> block offsets (64-bit); total -> st->accepted
__global__ __launch_bounds__(256) void k_scan_counts(const uint32_t *cnt, unsigned long long *off, int nb,
                                                     slk_rng_dev *st) {
    __shared__ unsigned long long s[256];
    __shared__ unsigned long long carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 256) {
        const int i = base + t;
        const unsigned long long v = (i < nb) ? cnt[i] : 0ull;
        s[t] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const unsigned long long x = (t >= d) ? s[t - d] : 0ull;
            __syncthreads();
            s[t] += x;
            __syncthreads();
        }
        if (i < nb) off[i] = carry + s[t] - v;
        __syncthreads();
        if (t == 255) carry += s[255];
        __syncthreads();
    }
    if (t == 0) st->accepted = carry;
}

__global__ __launch_bounds__(256) void k_accept_scatter(slk_accept_args a, const unsigned long long *off,
                                                        unsigned long long count, uint32_t *out32,
                                                        int64_t *out64, slk_rng_dev *st) {
    __shared__ unsigned s[256];
    const int t = threadIdx.x;
    const int pos0 = a.st->pos;
    const unsigned long long base = (unsigned long long)blockIdx.x * SLK_TILE + (unsigned long long)t * 8;
    uint32_t v[8];
    bool ok[8];
    unsigned c = 0;
    for (int j = 0; j < 8; ++j) {
        ok[j] = accept_word(a, base + j, pos0, &v[j]);
        c += ok[j] ? 1u : 0u;
    }
    s[t] = c;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned x = (t >= d) ? s[t - d] : 0u;
        __syncthreads();
        s[t] += x;
        __syncthreads();
    }
    unsigned long long rank = off[blockIdx.x] + (unsigned long long)(s[t] - c);
    for (int j = 0; j < 8; ++j) {
        if (!ok[j]) continue;
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
int slk_mt_generate_blocks(slk_ctx *ctx, unsigned long long nblocks, hipStream_t s) {
    uint32_t *raw = (uint32_t *)ctx->raw.p;
    {
        // block 0 of every launch is its input key block; further launches (> 10.2 M words)
        // restart from the last block of the previous one
        const unsigned long long cap = (unsigned long long)SLK_MT_JUMP_WG * SLK_MT_JUMP_BLOCKS;
        const size_t lds_bytes = (size_t)SLK_MT_LDS_WORDS * 4;
        if (!ctx->mt_attr_set) {  // per ctx, i.e. per device: function attributes live in the device's context
            SLK_HIP(ctx, hipFuncSetAttribute((const void *)k_mt_generate_jump,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            ctx->mt_attr_set = true;
        }
        unsigned long long start = 0;  // index of the launch's block 0
        const uint32_t *key_src = ctx->d_rng->key;
        while (true) {
            const unsigned long long nb_l = (nblocks - start < cap) ? nblocks - start : cap;
            const unsigned wgs = (unsigned)((nb_l + SLK_MT_JUMP_BLOCKS - 1) / SLK_MT_JUMP_BLOCKS);
            if (wgs > 1 && !ctx->d_jump) {
                const uint32_t *tab = slk_mt_jump_table(ctx);
                if (!tab) return SLK_EIO;
                const size_t bytes = (size_t)(SLK_MT_JUMP_WG - 1) * SLK_MT_JUMP_TERMS * 4;
                SLK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_jump), bytes));
                SLK_HIP(ctx, hipMemcpy(ctx->d_jump, tab, bytes, hipMemcpyHostToDevice));
            }
            hipLaunchKernelGGL(k_mt_generate_jump, dim3(wgs), dim3(SLK_MT_THREADS), lds_bytes, s, key_src,
                               (const uint32_t *)ctx->d_jump, raw + start * SLK_MT_N, (int)nb_l);
            SLK_LAUNCH_CHECK(ctx, "k_mt_generate_jump");
            if (start + nb_l >= nblocks) break;
            start += nb_l - 1;
            key_src = raw + start * SLK_MT_N;
        }
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
    if (nblocks > SLK_MT_JUMP_BLOCKS && !ctx->d_jump) {
        const uint32_t *tab = slk_mt_jump_table(ctx);
        if (!tab) return SLK_EIO;
        const size_t bytes = (size_t)(SLK_MT_JUMP_WG - 1) * SLK_MT_JUMP_TERMS * 4;
        SLK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_jump), bytes));
        SLK_HIP(ctx, hipMemcpy(ctx->d_jump, tab, bytes, hipMemcpyHostToDevice));
    }
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
