// slk_sort.hip -- the engine's own stable LSD radix sort for gfx950 (round 4: replaces rocPRIM's radix_sort_pairs, the one
// library kernel that was on the training path).
//
// What the row-owner passes need (slk_bilinear.hip, slk_seq.hip): interactions grouped by (minibatch, user) and occurrences
// grouped by (minibatch, item), every group in position order -- the sparse-gradient coalesce of the reference's
// `loss.backward(); optimizer.step()` (spotlight/factorization/implicit.py:242-243), made deterministic.  Three things a
// general-purpose library sort cannot know are used here:
//   * the minibatch is implied by the position (minibatch m = positions [m B, (m+1) B)), so its bits are never sorted: the
//     sort is SEGMENTED -- every minibatch of a chunk is sorted on the id bits only, all of them in the same launches
//     (C2: 24 bits = 3 passes instead of 27 = 4; 20 bits instead of 23 on the item side);
//   * the keys do not exist yet: the first pass builds them from the id arrays it reads anyway (k_build_user_keys /
//     k_build_item_keys are gone: one write and one read of every key and payload less);
//   * digits are split evenly (20 bits = 7 + 7 + 6, not 8 + 8 + 4): fewer buckets per pass = longer runs per (tile, bucket),
//     and the scatter's partial-line writes are what a radix pass pays for.
//
// Structure of one sort of n pairs in P passes: ONE histogram kernel (every pass's digit counts per segment, LDS atomics) ->
// ONE scan kernel (bucket bases) -> P scatter kernels.  A scatter workgroup takes a tile (ticket = dependency order), ranks
// its keys with wave-wide digit matches (ballots; stable by construction: element order = wave, round, lane), publishes the
// tile's digit counts and finds the counts of the tiles before it by decoupled look-back over one 32-bit {flag, count} word per
// (tile, digit) (sc1 stores / loads: the producer may sit on another XCD), exchanges keys then payloads through LDS so that
// every (tile, bucket) run leaves as consecutive lanes, and writes.  HBM-bound: 8 B (u32 payload) / 12 B (u64) read + written
// per pair and pass, + 4 B for the histogram.
#include "slk_common.h"

namespace {

constexpr int RS_RB = 8;                  // widest digit
constexpr int RS_RADIX = 1 << RS_RB;
constexpr uint32_t RS_VAL_MASK = (1u << 30) - 1u;  // status word: flag << 30 | count (segments hold < 2^30 pairs)
constexpr uint32_t RS_AGG = 1u, RS_INC = 2u;
constexpr int RS_MAX_PASS = 8;
constexpr int RS_HIST_KEYS = 8192;        // keys per histogram workgroup
constexpr int RS_LBW = 16;                // look-back: status words in flight per digit and round trip
constexpr uint32_t RS_MAX_SPINS = 1u << 22;  // x (s_sleep + one fabric round trip): seconds; then the sort gives up (sticky flag)

enum { RS_LOAD_PLAIN = 0, RS_LOAD_USER_FAT, RS_LOAD_USER_IDX, RS_LOAD_ITEM_OCC, RS_LOAD_FY_STEPS };

struct rs_args {
    const void *kin, *vin;
    void *kout, *vout;
    // sources of the fused first pass (the training prep, slk_bilinear.hip)
    const int64_t *users, *items;
    const uint32_t *neg32, *uit;
    uint32_t n, seg_len, tps, ntiles, hbps;  // pairs, pairs per segment, tiles / histogram workgroups per segment
    uint32_t kseg;                           // fused keys of an UNSEGMENTED sort: pairs per minibatch (0: the segment is the minibatch)
    unsigned idbits;                         // fused keys: (minibatch << idbits) | id
    uint32_t fy_n;                           // RS_LOAD_FY_STEPS: elements of the shuffle (slk_shuffle.hip)
    int npass, pass;
    int shift[RS_MAX_PASS], width[RS_MAX_PASS];
    uint32_t *hist, *base;   // [segment][pass][RS_RADIX]: digit counts / first output position of the bucket
    uint32_t *ticket;        // [pass]
    uint32_t *status;        // this pass: [tile][1 << width]
    int32_t *abort_flag;     // sticky (slk_rng_dev::sort_abort): a look-back gave up
    int xcd;                 // 1: a segment's tiles are taken by the workgroups of ONE XCD (segment s -> XCD s % 8)
    uint32_t nseg;
    int debug;               // measurement only (option "sort_debug"): 1 no look-back walk, 2 ranks from LDS atomics
};

template <class KeyT>
__device__ __forceinline__ uint32_t rs_digit(KeyT k, int shift, uint32_t mask) {
    return (uint32_t)(k >> shift) & mask;
}

// element gi of the sort's input: (key, payload)
template <class KeyT, class ValT, int LOADER>
__device__ __forceinline__ void rs_load(const rs_args &a, uint32_t gi, uint32_t seg, KeyT &k, ValT &v) {
    if (LOADER == RS_LOAD_FY_STEPS) {
        // the epoch shuffle's swaps (slk_shuffle.hip): step g (i = n - 1 - g) writes position j = J[g]: key = j, value = i;
        // a step that swaps with itself gets the sentinel key n
        const uint32_t i = a.fy_n - 1u - gi, j = a.uit[gi];
        k = (KeyT)(j == i ? a.fy_n : j);
        v = (ValT)i;
        return;
    }
    if (LOADER != RS_LOAD_PLAIN && a.kseg) seg = gi / a.kseg;
    if (LOADER == RS_LOAD_PLAIN) {
        k = ((const KeyT *)a.kin)[gi];
        v = ((const ValT *)a.vin)[gi];
    } else if (LOADER == RS_LOAD_USER_FAT) {
        // key = (minibatch-in-chunk, user); payload = the (positive, negative) item pair the user pass reads with the key
        k = (KeyT)((seg << a.idbits) | (uint32_t)a.users[gi]);
        v = (ValT)(((uint64_t)a.neg32[gi] << 32) | (uint64_t)(uint32_t)a.items[gi]);
    } else if (LOADER == RS_LOAD_USER_IDX) {
        k = (KeyT)((seg << a.idbits) | (uint32_t)a.users[gi]);
        v = (ValT)gi;
    } else {
        // occurrence r = (user-sorted position q, slot) of the packed item list: key = (minibatch, item), payload = r
        k = (KeyT)((seg << a.idbits) | a.uit[gi]);
        v = (ValT)gi;
    }
}

template <class KeyT, int LOADER>
__device__ __forceinline__ KeyT rs_load_key(const rs_args &a, uint32_t gi, uint32_t seg) {
    if (LOADER == RS_LOAD_PLAIN) return ((const KeyT *)a.kin)[gi];
    if (LOADER == RS_LOAD_FY_STEPS) {
        const uint32_t j = a.uit[gi];
        return (KeyT)(j == a.fy_n - 1u - gi ? a.fy_n : j);
    }
    if (a.kseg) seg = gi / a.kseg;
    if (LOADER == RS_LOAD_ITEM_OCC) return (KeyT)((seg << a.idbits) | a.uit[gi]);
    return (KeyT)((seg << a.idbits) | (uint32_t)a.users[gi]);
}

// ---- digit counts of every pass, per segment ------------------------------------------------------------------------------
template <class KeyT, int LOADER>
__global__ __launch_bounds__(256) void k_rs_hist(rs_args a) {
    __shared__ uint32_t sh[RS_MAX_PASS * RS_RADIX];
    const uint32_t seg = blockIdx.x / a.hbps, hb = blockIdx.x - seg * a.hbps;
    const uint32_t s0 = seg * a.seg_len;
    const uint32_t s1 = (a.n - s0 < a.seg_len) ? a.n : s0 + a.seg_len;
    const uint32_t b0 = s0 + hb * (uint32_t)RS_HIST_KEYS;
    const uint32_t b1 = (b0 >= s1 || s1 - b0 < (uint32_t)RS_HIST_KEYS) ? s1 : b0 + (uint32_t)RS_HIST_KEYS;  // (a short last segment)
    for (int i = threadIdx.x; i < a.npass * RS_RADIX; i += 256) sh[i] = 0u;
    __syncthreads();
    // eight keys per thread in flight (the kernel is otherwise one dependent load -> LDS-atomics chain per key)
    for (uint32_t g0 = b0 + threadIdx.x; g0 < b1; g0 += 8u * 256u) {
        KeyT k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t gi = g0 + (uint32_t)u * 256u;
            k[u] = gi < b1 ? rs_load_key<KeyT, LOADER>(a, gi, seg) : (KeyT)0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (g0 + (uint32_t)u * 256u >= b1) continue;
            for (int p = 0; p < a.npass; ++p)
                atomicAdd(&sh[p * RS_RADIX + rs_digit<KeyT>(k[u], a.shift[p], (1u << a.width[p]) - 1u)], 1u);
        }
    }
    __syncthreads();
    uint32_t *gh = a.hist + (size_t)seg * a.npass * RS_RADIX;
    for (int i = threadIdx.x; i < a.npass * RS_RADIX; i += 256) {
        const uint32_t c = sh[i];
        if (c) atomicAdd(&gh[i], c);
    }
}

// exclusive scan of one value per thread over the workgroup (THREADS a multiple of 64)
template <int THREADS>
__device__ __forceinline__ uint32_t rs_block_excl_scan(uint32_t v, uint32_t *s_wsum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wsum[w] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int i = 0; i < THREADS / 64; ++i)
        if (i < w) off += s_wsum[i];
    __syncthreads();
    return off + inc - v;
}

// bucket bases: base[segment][pass][d] = segment start + (pairs of the segment whose digit is < d)
__global__ __launch_bounds__(256) void k_rs_scan(rs_args a) {
    __shared__ uint32_t s_wsum[4];
    const uint32_t seg = blockIdx.x / (uint32_t)a.npass;
    const size_t o = (size_t)blockIdx.x * RS_RADIX + threadIdx.x;
    const uint32_t c = a.hist[o];
    a.base[o] = seg * a.seg_len + rs_block_excl_scan<256>(c, s_wsum);
}

// the XCD this workgroup runs on (0..7)
__device__ __forceinline__ uint32_t rs_xcc_id() {
#if defined(__HIPCC__)
    return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;  // HW_REG_XCC_ID[3:0]
#else
    return blockIdx.x & 7u;
#endif
}

// lanes of the wave that hold the same digit as this one (valid lanes only)
__device__ __forceinline__ unsigned long long rs_match(uint32_t d, int nbits, bool valid) {
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return valid ? m : 0ull;
}

// ---- one pass: tile -> ranks -> look-back -> LDS exchange -> scatter ----------------------------------------------------------
// The counts of the segment's tiles before this one, for digit d: RS_LBW status words per round trip, back to the nearest tile
// that knows its own inclusive count.  Returns false when a tile never published (the sort gives up: sticky flag).
__device__ __forceinline__ bool rs_walk(const rs_args &a, const uint32_t *row0, uint32_t radix, uint32_t tis, uint32_t &before) {
    int32_t t = (int32_t)tis - 1;
    uint32_t spins = 0;
    bool done = false;
    while (!done) {
        uint32_t v[RS_LBW];
#pragma unroll
        for (int i = 0; i < RS_LBW; ++i) {
            const int32_t ti = t - i;
            v[i] = ti >= 0 ? __hip_atomic_load(row0 + (size_t)ti * radix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
        int consumed = 0;
#pragma unroll
        for (int i = 0; i < RS_LBW; ++i) {
            const uint32_t f = v[i] >> 30;
            if (done || f == 0u || consumed != i) continue;
            before += v[i] & RS_VAL_MASK;
            ++consumed;
            if (f == RS_INC) done = true;
        }
        t -= consumed;
        if (!done && consumed == 0) {
            if (++spins > RS_MAX_SPINS) {  // a tile before this one never published: give up, loudly
                *a.abort_flag = 1;
                return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return true;
}

// SINGLE: the whole input is one tile (no histogram, no look-back: bucket bases are the tile's own scan).
// (A dedicated WALKER wave -- the workgroup's last wave holds no keys, publishes the counts, walks back with four digits per
// lane and forms the bucket offsets while the other seven rank -- was built and measured in round 4: 0.235 / 0.314 ms against
// 0.227 / 0.304 ms for the C2 user / item sorts without it, profiles/r04_l_*.  The walk of the FIRST wave of tiles is a chain
// through the tiles before them whoever walks it; the tiles behind find inclusive counts at once.  Removed.)
template <class KeyT, class ValT, int THREADS, int KPT, int LOADER, bool SINGLE>
__global__ __launch_bounds__(THREADS) SLK_WAVES_PER_EU_RANGE(4, 6) void k_rs_scatter(rs_args a) {
    constexpr int WAVES = THREADS / 64, KW = WAVES, TILE = KW * 64 * KPT;
    constexpr int OPT = (TILE + THREADS - 1) / THREADS;  // output positions per thread
    constexpr int XB = sizeof(KeyT) > sizeof(ValT) ? sizeof(KeyT) : sizeof(ValT);
    __shared__ uint32_t s_cnt[KW * RS_RADIX];     // per key wave: running digit counts, then the wave's first rank in the tile's bucket
    __shared__ uint32_t s_lbase[RS_RADIX];        // the tile's digit counts, then the first position of the bucket in the tile's sorted order
    __shared__ uint32_t s_goff[RS_RADIX];         // output index of the bucket's element at tile position i = s_goff + i
    __shared__ uint32_t s_wsum[WAVES];
    __shared__ uint32_t s_tile;
    __shared__ __attribute__((aligned(16))) unsigned char s_x[(size_t)TILE * XB];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool is_key = true;  // (every wave holds keys)
    const int shift = a.shift[a.pass], nbits = a.width[a.pass];
    const uint32_t radix = 1u << nbits, dmask = radix - 1u;
    if (!SINGLE && threadIdx.x == 0) {
        if (a.xcd) {
            // Segment s belongs to XCD s % 8: its tiles' runs meet in ONE L2 (adjacent tiles complete each other's partial lines
            // before they are written back) and its look-back words never cross the fabric.  A workgroup whose XCD has run out
            // of tiles takes one of the next XCD's (the grid has exactly one workgroup per tile, so every tile is taken); the
            // order of the tickets of one XCD is still the order of the dependency chain of its segments.
            const uint32_t x0 = rs_xcc_id();
            uint32_t t = 0xffffffffu;
            for (uint32_t i = 0; i < 8u && t == 0xffffffffu; ++i) {
                const uint32_t x = (x0 + i) & 7u;
                const uint32_t nsx = a.nseg > x ? (a.nseg - x + 7u) / 8u : 0u;
                if (!nsx) continue;
                const uint32_t k = atomicAdd(a.ticket + 8 + a.pass * 8 + x, 1u);
                if (k < nsx * a.tps) t = (x + 8u * (k / a.tps)) * a.tps + k % a.tps;
            }
            s_tile = t;
        } else {
            s_tile = atomicAdd(a.ticket + a.pass, 1u);
        }
    }
    for (int i = threadIdx.x; i < KW * RS_RADIX; i += THREADS) s_cnt[i] = 0u;
    if (threadIdx.x < RS_RADIX) s_lbase[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t tile = SINGLE ? 0u : s_tile;
    if (tile == 0xffffffffu) return;  // (cannot happen: one workgroup per tile)
    const uint32_t seg = tile / a.tps, tis = tile - seg * a.tps;
    const uint32_t s0 = seg * a.seg_len;
    const uint32_t s1 = (a.n - s0 < a.seg_len) ? a.n : s0 + a.seg_len;
    const uint32_t t0 = s0 + tis * (uint32_t)TILE;
    const uint32_t cnt = t0 >= s1 ? 0u : ((s1 - t0 < (uint32_t)TILE) ? s1 - t0 : (uint32_t)TILE);  // (0: a tile past a short last segment)

    KeyT key[KPT];
    ValT val[KPT];
    uint32_t pos2[KPT / 2];  // tile positions (< 2^16), two per register
    const uint32_t e0 = (uint32_t)wave * 64u * KPT + (uint32_t)lane;
    if (is_key) {
        // every key load first, then every payload load: the keys are waited for alone, the payloads land behind the ranking
        if (LOADER == RS_LOAD_PLAIN) {
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t e = e0 + (uint32_t)j * 64u;
                key[j] = e < cnt ? ((const KeyT *)a.kin)[t0 + e] : (KeyT)0;
            }
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t e = e0 + (uint32_t)j * 64u;
                val[j] = e < cnt ? ((const ValT *)a.vin)[t0 + e] : (ValT)0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const uint32_t e = e0 + (uint32_t)j * 64u;
                key[j] = 0;
                val[j] = 0;
                if (e < cnt) rs_load<KeyT, ValT, LOADER>(a, t0 + e, seg, key[j], val[j]);
            }
        }
        // the tile's digit counts: published BEFORE the ranking, so the tiles behind this one add them up while it ranks
        if (!SINGLE) {
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                if (e0 + (uint32_t)j * 64u < cnt) atomicAdd(&s_lbase[rs_digit<KeyT>(key[j], shift, dmask)], 1u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            key[j] = 0;
            val[j] = 0;
        }
    }
    if (!SINGLE) __syncthreads();

    uint32_t tcount = 0, before = 0;
    uint32_t *wcnt = s_cnt + (is_key ? wave : 0) * RS_RADIX;
    auto rank_keys = [&]() {
        // ranks inside the wave's 64 * KPT elements, in element order (round j, then lane)
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            if ((j & 1) == 0) pos2[j / 2] = 0u;
            if ((uint32_t)wave * 64u * KPT + (uint32_t)j * 64u >= cnt) continue;  // (wave-uniform: a round past the tile's end)
            const bool valid = e0 + (uint32_t)j * 64u < cnt;
            const uint32_t dj = rs_digit<KeyT>(key[j], shift, dmask);
            uint32_t r;
            if (a.debug & 2) {  // measurement only: ranks from LDS atomics (unstable order)
                r = valid ? atomicAdd(&wcnt[dj], 1u) : 0u;
            } else {
                const unsigned long long m = rs_match(dj, nbits, valid);
                const int leader = m ? __ffsll((long long)m) - 1 : lane;
                uint32_t first = 0;
                if (m && lane == leader) first = atomicAdd(&wcnt[dj], (uint32_t)__popcll(m));
                first = __shfl(first, leader);
                r = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            }
            pos2[j / 2] = (j & 1) ? (pos2[j / 2] | (r << 16)) : r;
        }
    };
    {
        const uint32_t d = threadIdx.x;
        if (!SINGLE && d < radix) {
            tcount = s_lbase[d];
            __hip_atomic_store(a.status + (size_t)tile * radix + d, ((tis == 0 ? RS_INC : RS_AGG) << 30) | tcount, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            // the digit owners (the first `radix` threads) walk back NOW, before they rank: the workgroup's other waves -- and
            // the other workgroups of the CU -- rank meanwhile
            if (tis != 0 && !(a.debug & 1)) {
                rs_walk(a, a.status + (size_t)(tile - tis) * radix + d, radix, tis, before);
                __hip_atomic_store(a.status + (size_t)tile * radix + d, (RS_INC << 30) | ((before + tcount) & RS_VAL_MASK), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        rank_keys();
        __syncthreads();
        // per digit: waves' counts -> the wave's first rank in the tile's bucket
        if (d < radix) {
            uint32_t run = 0;
            for (int w = 0; w < KW; ++w) {
                const uint32_t c = s_cnt[w * RS_RADIX + d];
                s_cnt[w * RS_RADIX + d] = run;
                run += c;
            }
            if (SINGLE) tcount = run;
        }
        const uint32_t lb = rs_block_excl_scan<THREADS>(tcount, s_wsum);
        if (d < radix) {
            uint32_t gbase = lb;
            if (!SINGLE) gbase = a.base[((size_t)seg * a.npass + a.pass) * RS_RADIX + d];
            s_lbase[d] = lb;
            s_goff[d] = gbase + before - lb;
        }
    }
    __syncthreads();
    // keys through LDS into the tile's sorted order, then out: thread i takes tile positions i, i + THREADS, ...
    KeyT *xk = reinterpret_cast<KeyT *>(s_x);
    if (is_key) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            if (e0 + (uint32_t)j * 64u < cnt) {
                const uint32_t dj = rs_digit<KeyT>(key[j], shift, dmask);
                const uint32_t pj = ((pos2[j / 2] >> ((j & 1) * 16)) & 0xffffu) + s_lbase[dj] + wcnt[dj];
                pos2[j / 2] = (j & 1) ? ((pos2[j / 2] & 0xffffu) | (pj << 16)) : ((pos2[j / 2] & 0xffff0000u) | pj);
                xk[pj] = key[j];
            }
        }
    }
    __syncthreads();
    uint32_t gidx[OPT];
#pragma unroll
    for (int j = 0; j < OPT; ++j) {
        const uint32_t i = (uint32_t)j * THREADS + threadIdx.x;
        gidx[j] = 0;
        if (i < cnt) {
            const KeyT k = xk[i];
            gidx[j] = s_goff[rs_digit<KeyT>(k, shift, dmask)] + i;
            ((KeyT *)a.kout)[gidx[j]] = k;
        }
    }
    __syncthreads();
    ValT *xv = reinterpret_cast<ValT *>(s_x);
    if (is_key) {
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if (e0 + (uint32_t)j * 64u < cnt) xv[(pos2[j / 2] >> ((j & 1) * 16)) & 0xffffu] = val[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < OPT; ++j) {
        const uint32_t i = (uint32_t)j * THREADS + threadIdx.x;
        if (i < cnt) ((ValT *)a.vout)[gidx[j]] = xv[i];
    }
}

struct rs_plan {
    uint32_t n = 0, seg_len = 0, nseg = 0, tps = 0, ntiles = 0, hbps = 0;
    int npass = 0, shift[RS_MAX_PASS], width[RS_MAX_PASS];
    size_t off_ticket = 0, off_status[RS_MAX_PASS], zero_bytes = 0, off_base = 0, off_data = 0, ctl_bytes = 0;
};

// digits: `bits` key bits in ceil(bits / 8) passes of (nearly) equal width
static void rs_split_bits(unsigned bits, int *npass, int *shift, int *width) {
    int P = (int)((bits + RS_RB - 1) / RS_RB);
    if (P < 1) P = 1;
    const int lo = (int)bits / P, rem = (int)bits % P;
    int s = 0;
    for (int p = 0; p < P; ++p) {
        width[p] = lo + (p < rem ? 1 : 0);
        if (width[p] < 1) width[p] = 1;
        shift[p] = s;
        s += width[p];
    }
    *npass = P;
}

static void rs_make_plan(rs_plan &pl, size_t n, size_t seg_len, unsigned bits, unsigned tile) {
    pl.n = (uint32_t)n;
    pl.seg_len = (uint32_t)(seg_len && seg_len < n ? seg_len : n);
    pl.nseg = (uint32_t)((n + pl.seg_len - 1) / pl.seg_len);
    pl.tps = (pl.seg_len + tile - 1) / tile;
    pl.ntiles = pl.nseg * pl.tps;
    pl.hbps = (pl.seg_len + RS_HIST_KEYS - 1) / RS_HIST_KEYS;
    rs_split_bits(bits, &pl.npass, pl.shift, pl.width);
    size_t o = (size_t)pl.nseg * pl.npass * RS_RADIX * 4;  // hist
    pl.off_ticket = o;
    o += 512;  // [0, 8): one ticket per pass; [8 + 8 p, 16 + 8 p): pass p's tickets per XCD
    for (int p = 0; p < pl.npass; ++p) {
        pl.off_status[p] = o;
        o += (size_t)pl.ntiles * ((size_t)1 << pl.width[p]) * 4;
    }
    pl.zero_bytes = o;
    o = (o + 255) & ~(size_t)255;
    pl.off_base = o;
    o += (size_t)pl.nseg * pl.npass * RS_RADIX * 4;
    pl.off_data = (o + 255) & ~(size_t)255;
    pl.ctl_bytes = pl.off_data;
}

// the large sorts' tile shape (build-time: the A/Bs of round 4 and round 6 are in profiles/)
#ifndef RS_BIG_THREADS
#define RS_BIG_THREADS 512
#endif
#ifndef RS_BIG_KPT
#define RS_BIG_KPT 16
#endif

template <class KeyT, class ValT, int LOADER>
struct rs_kernels {
    // two tile shapes: 256 threads x 16 keys, and 512 x 16 (twice the run length per (tile, bucket): the large sorts)
    static void launch(int cfg, bool single, unsigned grid, hipStream_t s, const rs_args &a) {
        if (single)
            hipLaunchKernelGGL((k_rs_scatter<KeyT, ValT, 256, 16, RS_LOAD_PLAIN, true>), dim3(1), dim3(256), 0, s, a);
        else if (cfg == 1)
            hipLaunchKernelGGL((k_rs_scatter<KeyT, ValT, RS_BIG_THREADS, RS_BIG_KPT, LOADER, false>), dim3(grid), dim3(RS_BIG_THREADS), 0, s, a);
        else
            hipLaunchKernelGGL((k_rs_scatter<KeyT, ValT, 256, 16, LOADER, false>), dim3(grid), dim3(256), 0, s, a);
    }
};

static inline unsigned rs_tile_of(int cfg) { return cfg == 1 ? (unsigned)(RS_BIG_THREADS * RS_BIG_KPT) : 4096u; }

// The sort.  Pass p reads what pass p - 1 wrote; the last pass writes (kout, vout).  Intermediate passes alternate between
// (kout, vout) and a second buffer pair: (kalt, valt) when the caller has one (the training prep's double buffers, or an
// input the caller allows to be overwritten), else space behind the control block in `scratch`.
template <class KeyT, class ValT, int LOADER>
static int rs_sort(slk_ctx *ctx, slk_buf &scratch, rs_args a, size_t n, size_t seg_len, unsigned bits, void *kalt, void *valt,
                   hipStream_t s) {
    if (n == 0) return SLK_OK;
    if (n >= ((size_t)1 << 31)) return slk_fail(ctx, SLK_EINVAL, "sort of %zu pairs: at most 2^31 - 1", n);
    if (bits > 8 * sizeof(KeyT)) bits = 8 * sizeof(KeyT);
    if (bits < 1) bits = 1;
    // (one tile, one workgroup -- for an UNSEGMENTED sort only: a segmented request of <= 4096 pairs keeps its segments)
    const bool single = LOADER == RS_LOAD_PLAIN && n <= 4096 && (seg_len == 0 || seg_len >= n);
    const int cfg = (!single && n >= (size_t)ctx->opt_sort_big_min) ? 1 : 0;
    const unsigned tile = rs_tile_of(cfg);
    rs_plan pl;
    rs_make_plan(pl, n, single ? 0 : seg_len, bits, tile);
    if (pl.seg_len >= (1u << 30)) return slk_fail(ctx, SLK_EINVAL, "sort segment of %u pairs: at most 2^30 - 1", pl.seg_len);
    // With the input as the second buffer pair (`clobber`) and an even number of passes the first pass would have to write over
    // its own input: that one hop goes through the scratch instead.
    const bool inplace0 = kalt && kalt == a.kin && (pl.npass & 1) == 0;
    const bool need_tmp = pl.npass > 1 && (!kalt || inplace0);
    const size_t koff = pl.off_data, voff = (koff + n * sizeof(KeyT) + 255) & ~(size_t)255;
    int rc = slk_ensure(ctx, scratch, need_tmp ? voff + n * sizeof(ValT) : pl.ctl_bytes);
    if (rc) return rc;
    char *base = (char *)scratch.p;
    void *k0dst = nullptr, *v0dst = nullptr;  // destination of pass 0 when it is neither the final nor the alternate pair
    if (need_tmp && !kalt) {
        kalt = base + koff;
        valt = base + voff;
    } else if (inplace0) {
        k0dst = base + koff;
        v0dst = base + voff;
    }
    a.n = pl.n;
    a.seg_len = pl.seg_len;
    a.tps = pl.tps;
    a.ntiles = pl.ntiles;
    a.hbps = pl.hbps;
    a.npass = pl.npass;
    for (int p = 0; p < pl.npass; ++p) {
        a.shift[p] = pl.shift[p];
        a.width[p] = pl.width[p];
    }
    a.hist = (uint32_t *)base;
    a.base = (uint32_t *)(base + pl.off_base);
    a.ticket = (uint32_t *)(base + pl.off_ticket);
    a.abort_flag = &ctx->d_rng->sort_abort;
    a.debug = ctx->opt_sort_debug;
    a.nseg = pl.nseg;
    a.xcd = pl.nseg >= 2 ? 1 : 0;
    void *kfinal = a.kout, *vfinal = a.vout;
    if (!single) {
        SLK_HIP(ctx, hipMemsetAsync(base, 0, pl.zero_bytes, s));
        a.pass = 0;
        hipLaunchKernelGGL((k_rs_hist<KeyT, LOADER>), dim3(pl.nseg * pl.hbps), dim3(256), 0, s, a);
        SLK_LAUNCH_CHECK(ctx, "k_rs_hist");
        hipLaunchKernelGGL(k_rs_scan, dim3(pl.nseg * (unsigned)pl.npass), dim3(256), 0, s, a);
        SLK_LAUNCH_CHECK(ctx, "k_rs_scan");
    }
    for (int p = 0; p < pl.npass; ++p) {
        const bool to_final = ((pl.npass - 1 - p) & 1) == 0;
        a.pass = p;
        a.kout = to_final ? kfinal : kalt;
        a.vout = to_final ? vfinal : valt;
        if (p == 0 && k0dst) {
            a.kout = k0dst;
            a.vout = v0dst;
        }
        a.status = (uint32_t *)(base + pl.off_status[p]);
        if (p == 0)
            rs_kernels<KeyT, ValT, LOADER>::launch(cfg, single, pl.ntiles, s, a);
        else
            rs_kernels<KeyT, ValT, RS_LOAD_PLAIN>::launch(cfg, single, pl.ntiles, s, a);
        SLK_LAUNCH_CHECK(ctx, "k_rs_scatter");
        a.kin = a.kout;
        a.vin = a.vout;
    }
    return SLK_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// generic stable pair sorts by the key bits [0, end_bit).  `clobber`: the input arrays may be used as the second buffer pair.
// ---------------------------------------------------------------------------------------------------------------------------
int slk_sort_pairs_u32_u32_in(slk_ctx *ctx, slk_buf &scratch, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                              uint32_t *vout, size_t n, unsigned end_bit, hipStream_t s, bool clobber) {
    rs_args a;
    memset(&a, 0, sizeof(a));
    a.kin = kin;
    a.vin = vin;
    a.kout = kout;
    a.vout = vout;
    return rs_sort<uint32_t, uint32_t, RS_LOAD_PLAIN>(ctx, scratch, a, n, 0, end_bit, clobber ? (void *)kin : nullptr,
                                                      clobber ? (void *)vin : nullptr, s);
}

// 64-bit keys (timestamps whose range needs more than 32 bits, slk_seqprep.hip)
int slk_sort_pairs_u64_u32_in(slk_ctx *ctx, slk_buf &scratch, const uint64_t *kin, uint64_t *kout, const uint32_t *vin,
                              uint32_t *vout, size_t n, unsigned end_bit, hipStream_t s, bool clobber) {
    rs_args a;
    memset(&a, 0, sizeof(a));
    a.kin = kin;
    a.vin = vin;
    a.kout = kout;
    a.vout = vout;
    return rs_sort<uint64_t, uint32_t, RS_LOAD_PLAIN>(ctx, scratch, a, n, 0, end_bit, clobber ? (void *)kin : nullptr,
                                                      clobber ? (void *)vin : nullptr, s);
}

int slk_sort_pairs_u32_u32(slk_ctx *ctx, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, size_t n,
                           unsigned end_bit, hipStream_t s, bool clobber) {
    return slk_sort_pairs_u32_u32_in(ctx, ctx->sort_tmp, kin, kout, vin, vout, n, end_bit, s, clobber);
}

int slk_sort_pairs_u32_u64(slk_ctx *ctx, const uint32_t *kin, uint32_t *kout, const uint64_t *vin, uint64_t *vout, size_t n,
                           unsigned end_bit, hipStream_t s, bool clobber) {
    rs_args a;
    memset(&a, 0, sizeof(a));
    a.kin = kin;
    a.vin = vin;
    a.kout = kout;
    a.vout = vout;
    return rs_sort<uint32_t, uint64_t, RS_LOAD_PLAIN>(ctx, ctx->sort_tmp, a, n, 0, end_bit, clobber ? (void *)kin : nullptr,
                                                      clobber ? (void *)vin : nullptr, s);
}

// The epoch shuffle's (target position, step) pairs, formed by the first pass from the draws J[0 .. n - 2] themselves (rounds
// 1-5: a kernel wrote both arrays first) and sorted by target position, stable: result in (key[1], val[1]), (key[0], val[0]) is
// the second buffer pair.  `scratch`: the shuffle's own sort scratch (it may run on another stream than the training sorts).
int slk_sort_fy_steps(slk_ctx *ctx, slk_buf &scratch, const uint32_t *J, uint32_t n, uint32_t *const key[2], uint32_t *const val[2],
                      hipStream_t s) {
    rs_args a;
    memset(&a, 0, sizeof(a));
    a.uit = J;
    a.fy_n = n;
    a.kout = key[1];
    a.vout = val[1];
    return rs_sort<uint32_t, uint32_t, RS_LOAD_FY_STEPS>(ctx, scratch, a, (size_t)n - 1, 0, slk_bits_for((uint64_t)n), key[0], val[0], s);
}

// measurement / test entry (slk_probe.hip): kind 0 = u32 keys + u32 payloads, 1 = u32 + u64, 2 = u64 + u32; seg_len > 0 sorts
// every segment of seg_len pairs on its own (on the key bits [0, bits))
int slk_sort_pairs_any(slk_ctx *ctx, int kind, const void *kin, void *kout, const void *vin, void *vout, size_t n, size_t seg_len,
                       unsigned bits, hipStream_t s, bool clobber) {
    rs_args a;
    memset(&a, 0, sizeof(a));
    a.kin = kin;
    a.vin = vin;
    a.kout = kout;
    a.vout = vout;
    void *ka = clobber ? (void *)kin : nullptr, *va = clobber ? (void *)vin : nullptr;
    if (kind == 0) return rs_sort<uint32_t, uint32_t, RS_LOAD_PLAIN>(ctx, ctx->sort_tmp, a, n, seg_len, bits, ka, va, s);
    if (kind == 1) return rs_sort<uint32_t, uint64_t, RS_LOAD_PLAIN>(ctx, ctx->sort_tmp, a, n, seg_len, bits, ka, va, s);
    if (kind == 2) return rs_sort<uint64_t, uint32_t, RS_LOAD_PLAIN>(ctx, ctx->sort_tmp, a, n, seg_len, bits, ka, va, s);
    return slk_fail(ctx, SLK_EINVAL, "slk_sort_pairs_any: kind %d", kind);
}

// ---------------------------------------------------------------------------------------------------------------------------
// the training prep's sorts (slk_bilinear.hip::do_sort): keys built by the first pass, one segment per minibatch
// ---------------------------------------------------------------------------------------------------------------------------
// Is a chunk of minibatches of `seg_len` pairs each sorted as segments (on the id bits only)?  Short segments are not: their
// last tiles would be mostly empty -- the chunk is then one array sorted on (minibatch, id).
static bool rs_segmented(size_t n, size_t seg_len) { return seg_len < n && seg_len >= 32768; }

template <class KeyT, class ValT, int LOADER>
static int rs_sort_fused(slk_ctx *ctx, rs_args a, size_t n, size_t seg_len, unsigned idbits, unsigned mbbits, void *const key[2],
                         void *const val[2], hipStream_t s) {
    a.idbits = idbits;
    a.kout = key[1];
    a.vout = val[1];
    const bool seg = rs_segmented(n, seg_len), one = seg_len >= n;  // one: a single minibatch, its number (0) needs no bits
    a.kseg = seg ? 0u : (uint32_t)seg_len;
    return rs_sort<KeyT, ValT, LOADER>(ctx, ctx->sort_tmp, a, n, seg ? seg_len : 0, (seg || one) ? idbits : idbits + mbbits, key[0],
                                       val[0], s);
}

// interactions of a chunk -> (key = (minibatch, user), payload = (negative << 32 | positive)) sorted by key, stable; the result
// in (key[1], val[1]), (key[0], val[0]) is scratch
int slk_sort_user_fat(slk_ctx *ctx, const int64_t *users, const int64_t *items, const uint32_t *neg32, size_t nc, size_t bsz,
                      unsigned ubits, unsigned mbbits, uint32_t *const key[2], uint64_t *const val[2], hipStream_t s) {
    rs_args a;
    memset(&a, 0, sizeof(a));
    a.users = users;
    a.items = items;
    a.neg32 = neg32;
    return rs_sort_fused<uint32_t, uint64_t, RS_LOAD_USER_FAT>(ctx, a, nc, bsz, ubits, mbbits, (void *const *)key, (void *const *)val, s);
}

// same keys, payload = the interaction's index in the chunk (the routes whose dL/dscore exists before the user pass)
int slk_sort_user_idx(slk_ctx *ctx, const int64_t *users, size_t nc, size_t bsz, unsigned ubits, unsigned mbbits,
                      uint32_t *const key[2], uint32_t *const val[2], hipStream_t s) {
    rs_args a;
    memset(&a, 0, sizeof(a));
    a.users = users;
    return rs_sort_fused<uint32_t, uint32_t, RS_LOAD_USER_IDX>(ctx, a, nc, bsz, ubits, mbbits, (void *const *)key, (void *const *)val, s);
}

// occurrences r of the packed item list uit[nocc] (NP per user-sorted position) -> (key = (minibatch, item), payload = r)
int slk_sort_item_occ(slk_ctx *ctx, const uint32_t *uit, size_t nocc, size_t bsz, int NP, unsigned ibits, unsigned mbbits,
                      uint32_t *const key[2], uint32_t *const val[2], hipStream_t s) {
    rs_args a;
    memset(&a, 0, sizeof(a));
    a.uit = uit;
    return rs_sort_fused<uint32_t, uint32_t, RS_LOAD_ITEM_OCC>(ctx, a, nocc, bsz * (size_t)NP, ibits, mbbits, (void *const *)key,
                                                               (void *const *)val, s);
}

// Sizes ctx->sort_tmp's control block (histograms, tickets, look-back words) for sorts of up to n pairs, so that a training
// call allocates nothing; the second buffer pair of a sort without one is added on demand.
int slk_sort_reserve(slk_ctx *ctx, size_t n) {
    if (n == 0) return SLK_OK;
    rs_plan pl;
    rs_make_plan(pl, n, 0, 32, 4096);
    size_t need = pl.ctl_bytes;
    // a segmented plan of the same size has one partial tile per segment more: bounded by 2 x the tiles
    need += (size_t)pl.ntiles * RS_RADIX * 4 * 4;
    return slk_ensure(ctx, ctx->sort_tmp, need);
}
