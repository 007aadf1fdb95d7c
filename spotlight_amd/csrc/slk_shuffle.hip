// slk_shuffle.hip -- numpy's legacy RandomState.shuffle(arange(n)) on the GPU, bit-exact.
//
// Replaces the host half of spotlight/torch_utils.py:35-52 (shuffle): the reference permutes the
// epoch's ids with numpy's serial Fisher-Yates,
//     for i = n-1 .. 1:  j = rk_interval(i)  (v = next32() & mask(i) until v <= i);  swap(x[i], x[j])
// which costs 20-30 ns per interaction on the host -- 50-100x the time the GPU needs to TRAIN on
// that interaction.  Both halves of the loop are sequential as written; both
// have an exact parallel form:
//
//  (1) THE DRAWS.  Whether raw word t is accepted depends on how many words before it were
//      (i = n-1 - #accepted).  Per power-of-two range of i (constant mask) the decisions
//      d(t) = [A(t) < need  and  v_t <= hi - A(t)],  A = exclusive prefix sum of d
//      are found by fixpoint iteration from the expected curve A(t) = (hi+1)(1 - exp(-t/(mask+1))):
//      if A is right on [0, t) then so is the next iterate on [0, t], so the first wrong index
//      advances every sweep (no cycles), and because shifting A by delta flips only ~0.7 delta
//      decisions the error contracts geometrically -- ~20 sweeps of an 8-byte-per-word scan
//      instead of 1e8 dependent steps.  The last 4095 draws are taken in order by one lane.
//      Round 2 (default): the sweeps run over a BAND only.  A(t) stays within a few standard deviations of the curve
//      x(t) above (a sum of t Bernoulli decisions: sigma <= sqrt(t)/2), so a word whose threshold hi - v_t lies
//      outside [x(t) - D(t), x(t) + D(t)], D(t) = 4 sqrt(t) + 16 (8 sigma), is accepted or rejected whatever the exact
//      count is -- all but ~3 * 2^(k/2) words of a range with mask 2^k - 1.  One pass classifies and counts, the
//      uncertain words (threshold and the number of certain accepts before them) are compacted into a list that ONE
//      workgroup resolves (every thread walks its chunk of the list exactly from a starting count; the fixpoint is over
//      the 1024 starting counts only), and a last pass emits the draws and CHECKS |A(t) - x(t)| <= D(t) for every word it
//      used: if the band was ever left (it is not, short of an 8-sigma walk) the range is redone with the full sweeps.
//      Three passes over the words instead of ~20: 10^8 ids 30.7 -> see profiles/README.md.
//  (2) THE SWAPS.  Step i makes x[i] final and moves the value that sat at i into j_i.  Hence, with
//      T_p = the steps i' > p with j_i' = p in time order:  X[p] (the value at p when step p
//      runs) = X[last of T_p] or p if none -- a forest of pointers to larger indices, resolved by
//      walking each (short) chain -- and  final[first of T_p] = p,  final[next of T_p] = X[previous of T_p],
//      final[i] = X[i] for self-swaps and for position 0.  T_p = one stable radix sort of (j_i, i).
//
// The RNG state afterwards is exactly numpy's (block of the last consumed word, pos = offset + 1),
// so the negatives drawn next continue the same stream.
#include <math.h>

#include "slk_common.h"

#define SLK_MT_N 624
#define FY_TILE 2048       // words per workgroup in the scan kernels (256 threads x 8)
#define FY_TAIL 4096       // draws for i < FY_TAIL are taken by one wavefront (k_fy_tail)

enum { FY_B0 = 26, FY_B1, FY_B2, FY_B3, FY_B4, FY_SMALL, FY_SORT };  // ctx->extra slots (32 in all)

__device__ __forceinline__ uint32_t fy_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

struct fy_args {
    const uint32_t *raw;     // untempered stream, word 0 = first word of the ctx's key block
    unsigned long long w0;   // absolute index of the window's first word
    uint32_t W;              // window length
    uint32_t mask, hi, need; // this range: draws for i = hi, hi-1, ..., hi-need+1 (all share mask)
    const uint32_t *A_old;   // exclusive count of accepted words before t (current iterate)
    uint32_t *A_new;
};

__device__ __forceinline__ bool fy_decide(const fy_args &a, uint32_t t, uint32_t *v) {
    if (t >= a.W) return false;
    const uint32_t acc = a.A_old[t];
    if (acc >= a.need) return false;  // the range is complete: later words belong to the next one
    *v = fy_temper(a.raw[a.w0 + t]) & a.mask;
    return *v <= a.hi - acc;
}

__global__ __launch_bounds__(256) void k_fy_init(uint32_t *A, uint32_t W, uint32_t need, uint32_t hi, uint32_t mask) {
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < W; t += gridDim.x * 256) {
        const double x = ((double)hi + 1.0) * (1.0 - exp(-(double)t / ((double)mask + 1.0)));
        A[t] = x >= (double)need ? need : (uint32_t)x;
    }
}

__global__ __launch_bounds__(256) void k_fy_count(fy_args a, uint32_t *cnt) {
    __shared__ unsigned s[256];
    const uint32_t base = blockIdx.x * FY_TILE + threadIdx.x * 8;
    unsigned c = 0;
    for (int j = 0; j < 8; ++j) {
        uint32_t v;
        c += fy_decide(a, base + j, &v) ? 1u : 0u;
    }
    s[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[blockIdx.x] = s[0];
}

// exclusive scan of cnt[nb] -> off[nb]; off[nb] = the total.  One workgroup: thread t adds up its contiguous share of the
// tiles, ONE scan over the 256 shares, then every thread writes its tiles' offsets (rounds 1-4 scanned 256 tiles at a time,
// sixteen barriers per 256: 47 us for the 10^4 tiles of a 2^25-id shuffle's widest range, twice per range).
__global__ __launch_bounds__(256) void k_fy_scan(const uint32_t *cnt, uint32_t *off, int nb) {
    __shared__ uint32_t s[256];
    const int t = threadIdx.x;
    const int per = (nb + 255) / 256;
    const int i0 = t * per < nb ? t * per : nb, i1 = i0 + per < nb ? i0 + per : nb;
    uint32_t sum = 0;
    for (int i = i0; i < i1; ++i) sum += cnt[i];
    s[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t x = (t >= d) ? s[t - d] : 0u;
        __syncthreads();
        s[t] += x;
        __syncthreads();
    }
    uint32_t run = s[t] - sum;
    for (int i = i0; i < i1; ++i) {
        off[i] = run;
        run += cnt[i];
    }
    if (t == 255) off[nb] = s[255];
}

// A_new = exclusive prefix sum of the decisions taken with A_old; *changed |= (A_new != A_old).
// The same pass counts, per tile, the decisions A_new itself implies -- the next sweep's tile counts --
// so a sweep is one scan of nb counters plus this kernel (12 bytes per word instead of 20).
__global__ __launch_bounds__(256) void k_fy_apply(fy_args a, const uint32_t *off, uint32_t *cnt_next, int *changed) {
    __shared__ unsigned s[256];
    const int t = threadIdx.x;
    const uint32_t base = blockIdx.x * FY_TILE + t * 8;
    bool ok[8];
    uint32_t vals[8];
    unsigned c = 0;
    for (int j = 0; j < 8; ++j) {
        vals[j] = 0u;
        ok[j] = fy_decide(a, base + j, &vals[j]);
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
    uint32_t run = off[blockIdx.x] + (s[t] - c);
    bool diff = false;
    unsigned cn = 0;
    for (int j = 0; j < 8; ++j) {
        if (base + j < a.W) {
            diff |= a.A_old[base + j] != run;
            a.A_new[base + j] = run;
            // the decision the next sweep will take for this word (same word, new count)
            const uint32_t v = fy_temper(a.raw[a.w0 + base + j]) & a.mask;
            cn += (run < a.need && v <= a.hi - run) ? 1u : 0u;
        }
        run += ok[j] ? 1u : 0u;
    }
    if (diff) *changed = 1;
    __syncthreads();
    s[t] = cn;
    __syncthreads();
    for (int offn = 128; offn >= 1; offn >>= 1) {
        if (t < offn) s[t] += s[t + offn];
        __syncthreads();
    }
    if (t == 0) cnt_next[blockIdx.x] = s[0];
}

// ---- the banded form of (1) -----------------------------------------------------------------------------------------
// Classification of the 8 consecutive words one thread owns (t0 .. t0 + 7).  x advances by its own derivative per word
// ((hi + 1 - x) / (mask + 1): the curve's ODE; eight Euler steps of 1e-8 relative size), D is taken at the thread's last
// word.  Every kernel below calls this with the same arguments, so they agree on every word.
//   cls: 1 = accepted for every count in the band, 0 = rejected for every count in the band, 2 = uncertain
struct fy_cls8 {
    int thr[8];        // accepted iff A(t) <= thr (v_t <= hi - A(t))
    uint32_t v[8];
    unsigned char cls[8];
    double x[8];       // the curve at the word
    double D;
};

__device__ __forceinline__ void fy_classify8(const fy_args &a, uint32_t t0, int band_div, fy_cls8 &c) {
    const double M = (double)a.mask + 1.0, H = (double)a.hi + 1.0;
    double x = H * (1.0 - exp(-(double)t0 / M));
    c.D = (4.0 * sqrt((double)t0 + 8.0) + 16.0) / (double)band_div;  // band_div > 1: a test hook that provokes the fall-back
    for (int j = 0; j < 8; ++j) {
        const uint32_t t = t0 + (uint32_t)j;
        c.x[j] = x;
        c.cls[j] = 0;
        c.thr[j] = -1;
        c.v[j] = 0u;
        if (t < a.W) {
            const uint32_t v = fy_temper(a.raw[a.w0 + t]) & a.mask;
            const int thr = (int)a.hi - (int)v;  // both < 2^31
            c.v[j] = v;
            c.thr[j] = thr;
            if ((double)thr >= x + c.D + 1.0) c.cls[j] = 1;
            else if ((double)thr < x - c.D - 1.0) c.cls[j] = 0;
            else c.cls[j] = 2;
        }
        x += (H - x) / M;
    }
}

// pass 1: per tile, the number of certain accepts and of uncertain words
__global__ __launch_bounds__(256) void k_fyb_count(fy_args a, int band_div, uint32_t *cnt_cert, uint32_t *cnt_unc) {
    __shared__ unsigned s[256], u[256];
    fy_cls8 c;
    fy_classify8(a, blockIdx.x * FY_TILE + threadIdx.x * 8, band_div, c);
    unsigned nc = 0, nu = 0;
    for (int j = 0; j < 8; ++j) {
        nc += c.cls[j] == 1 ? 1u : 0u;
        nu += c.cls[j] == 2 ? 1u : 0u;
    }
    s[threadIdx.x] = nc;
    u[threadIdx.x] = nu;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s[threadIdx.x] += s[threadIdx.x + off];
            u[threadIdx.x] += u[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cnt_cert[blockIdx.x] = s[0];
        cnt_unc[blockIdx.x] = u[0];
    }
}

// exclusive prefix of two per-thread counts over the 256 threads of a block (returns this thread's two prefixes)
__device__ __forceinline__ void fy_block_excl2(unsigned a_in, unsigned b_in, unsigned *sa, unsigned *sb, unsigned &a_out,
                                               unsigned &b_out) {
    const int t = threadIdx.x;
    sa[t] = a_in;
    sb[t] = b_in;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned xa = (t >= d) ? sa[t - d] : 0u, xb = (t >= d) ? sb[t - d] : 0u;
        __syncthreads();
        sa[t] += xa;
        sb[t] += xb;
        __syncthreads();
    }
    a_out = sa[t] - a_in;
    b_out = sb[t] - b_in;
}

// pass 2: the uncertain words, in stream order: their threshold and the number of certain accepts before them
__global__ __launch_bounds__(256) void k_fyb_collect(fy_args a, int band_div, const uint32_t *off_cert, const uint32_t *off_unc,
                                                     int *u_thr, uint32_t *u_cert) {
    __shared__ unsigned sa[256], sb[256];
    fy_cls8 c;
    fy_classify8(a, blockIdx.x * FY_TILE + threadIdx.x * 8, band_div, c);
    unsigned nc = 0, nu = 0;
    for (int j = 0; j < 8; ++j) {
        nc += c.cls[j] == 1 ? 1u : 0u;
        nu += c.cls[j] == 2 ? 1u : 0u;
    }
    unsigned pc, pu;
    fy_block_excl2(nc, nu, sa, sb, pc, pu);
    if (nu == 0) return;
    uint32_t cert = off_cert[blockIdx.x] + pc, k = off_unc[blockIdx.x] + pu;
    for (int j = 0; j < 8; ++j) {
        if (c.cls[j] == 2) {
            u_thr[k] = c.thr[j];
            u_cert[k] = cert;
            ++k;
        }
        cert += c.cls[j] == 1 ? 1u : 0u;
    }
}

// The uncertain words' decisions, by ONE workgroup: word k is accepted iff u_cert[k] + (accepted uncertain words before k)
// <= u_thr[k].  Thread i owns the list chunk [i * len, (i + 1) * len) and walks it exactly from a starting count; the
// starting counts are iterated (each round: walk, block-scan the chunks' accept counts) until none changes -- chunk 0 is
// right after the first round, and a wrong start perturbs few decisions, so a handful of rounds do.  u_acc[k] = accepted
// uncertain words before k (u_acc[L] = all of them).
__global__ __launch_bounds__(1024) void k_fyb_resolve(const uint32_t *off_unc, int nb, const int *u_thr, const uint32_t *u_cert,
                                                      uint32_t *u_acc, int *status) {
    __shared__ uint32_t start[1024], cnt[1024];
    __shared__ int changed;
    const uint32_t L = off_unc[nb];
    const int t = threadIdx.x;
    const uint32_t len = (L + 1023u) / 1024u;
    const uint32_t k0 = (uint32_t)t * len < L ? (uint32_t)t * len : L, k1 = k0 + len < L ? k0 + len : L;
    start[t] = 0;
    __syncthreads();
    for (int round = 0; round <= 1024; ++round) {
        uint32_t acc = start[t];
        for (uint32_t k = k0; k < k1; ++k) acc += ((int)(u_cert[k] + acc) <= u_thr[k]) ? 1u : 0u;
        cnt[t] = acc - start[t];
        if (t == 0) changed = 0;
        __syncthreads();
        // exclusive scan of the chunk counts -> the next starting counts
        for (int d = 1; d < 1024; d <<= 1) {
            const uint32_t x = (t >= d) ? cnt[t - d] : 0u;
            __syncthreads();
            cnt[t] += x;
            __syncthreads();
        }
        const uint32_t ns = t ? cnt[t - 1] : 0u;
        if (ns != start[t]) changed = 1;
        __syncthreads();
        start[t] = ns;
        const int ch = changed;
        __syncthreads();
        if (!ch) break;
        if (round == 1024 && t == 0) *status = 1;  // cannot happen: chunk i is final after round i
    }
    uint32_t acc = start[t];
    for (uint32_t k = k0; k < k1; ++k) {
        u_acc[k] = acc;
        acc += ((int)(u_cert[k] + acc) <= u_thr[k]) ? 1u : 0u;
    }
    if (k1 == L && (k0 < L || t == 0)) u_acc[L] = acc;  // the owner of the last element (thread 0 if the list is empty)
}

// pass 3: every word's decision is known -> J[g0 + A(t)] = v_t for the accepted words with A(t) < need; *consumed = words
// used by the range; *status |= 2 if a word the range used lay outside the band the classification assumed
__global__ __launch_bounds__(256) void k_fyb_emit(fy_args a, int band_div, const uint32_t *off_cert, const uint32_t *off_unc,
                                                  const int *u_thr, const uint32_t *u_cert, const uint32_t *u_acc, uint32_t *J,
                                                  uint32_t g0, uint32_t *consumed, int *status) {
    __shared__ unsigned sa[256], sb[256];
    fy_cls8 c;
    fy_classify8(a, blockIdx.x * FY_TILE + threadIdx.x * 8, band_div, c);
    unsigned nu = 0;
    for (int j = 0; j < 8; ++j) nu += c.cls[j] == 2 ? 1u : 0u;
    // this thread's uncertain words are list entries ku .. ku + nu - 1; their decisions from the resolved counts
    unsigned dummy, pu;
    fy_block_excl2(0u, nu, sa, sb, dummy, pu);
    const uint32_t ku0 = off_unc[blockIdx.x] + pu;
    bool acc[8];
    unsigned na = 0;
    {
        uint32_t ku = ku0;
        for (int j = 0; j < 8; ++j) {
            if (c.cls[j] == 2) {
                acc[j] = (int)(u_cert[ku] + u_acc[ku]) <= u_thr[ku];
                ++ku;
            } else {
                acc[j] = c.cls[j] == 1;
            }
            na += acc[j] ? 1u : 0u;
        }
    }
    unsigned pa;
    fy_block_excl2(na, 0u, sa, sb, pa, dummy);
    // accepts before this tile = certain ones + uncertain ones accepted
    uint32_t A = off_cert[blockIdx.x] + u_acc[off_unc[blockIdx.x]] + pa;
    bool viol = false;
    const uint32_t t0 = blockIdx.x * FY_TILE + threadIdx.x * 8;
    for (int j = 0; j < 8; ++j) {
        if (t0 + j < a.W && A < a.need) {
            const double dev = (double)A - c.x[j];
            viol |= dev > c.D || dev < -c.D;
            if (acc[j]) {
                J[g0 + A] = c.v[j];
                if (A == a.need - 1) *consumed = t0 + j + 1;
            }
        }
        A += acc[j] ? 1u : 0u;
    }
    if (viol) atomicOr(status, 2);
}

// the converged decisions: J[g0 + A(t)] = v_t for the accepted words; *consumed = words used by the range
__global__ __launch_bounds__(256) void k_fy_emit(fy_args a, uint32_t *J, uint32_t g0, uint32_t *consumed) {
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < a.W; t += gridDim.x * 256) {
        uint32_t v;
        if (fy_decide(a, t, &v)) {
            const uint32_t acc = a.A_old[t];
            J[g0 + acc] = v;
            if (acc == a.need - 1) *consumed = t + 1;
        }
    }
}

// the last draws (i = i_start .. 1) by ONE wavefront, range by range (constant mask).  A block of 512 raw words at a time:
// every lane walks ITS 8 consecutive words exactly (numpy's rk_interval loop: accept v <= i, then --i) from a starting
// count; the starting counts are the wave-wide exclusive prefix of the lanes' accept counts, iterated until no lane's
// changes -- lane l is right after round l + 1 at the latest, usually after a few.  (A single lane walking the ~5900
// words in order took 0.67 ms: two thirds of the whole shuffle of a MovieLens-100K-sized epoch.)
#define FY_TAIL_C 8
__global__ __launch_bounds__(64) void k_fy_tail(const uint32_t *raw, unsigned long long w0,
                                                unsigned long long total_words, uint32_t i_start, uint32_t *J,
                                                uint32_t g0, uint32_t *consumed) {
    const int lane = threadIdx.x;
    unsigned long long w = w0;  // next unconsumed word
    uint32_t i = i_start, g = g0;
    bool starved = false;
    while (i >= 1 && !starved) {
        uint32_t mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const uint32_t lo = (mask >> 1) + 1, hi = i;
        const uint32_t need = hi - lo + 1;  // the draws for i = hi, hi - 1, ..., lo
        uint32_t a0 = 0;                    // accepted so far in this range
        while (a0 < need) {
            if (w >= total_words) {
                starved = true;
                break;
            }
            uint32_t v[FY_TAIL_C];
            bool valid[FY_TAIL_C];
#pragma unroll
            for (int j = 0; j < FY_TAIL_C; ++j) {
                const unsigned long long idx = w + (unsigned long long)(lane * FY_TAIL_C + j);
                valid[j] = idx < total_words;
                v[j] = valid[j] ? (fy_temper(raw[idx]) & mask) : 0u;
            }
            uint32_t s = a0;
            for (int round = 0; round < 66; ++round) {
                uint32_t acc = s;
#pragma unroll
                for (int j = 0; j < FY_TAIL_C; ++j) acc += (valid[j] && acc < need && v[j] <= hi - acc) ? 1u : 0u;
                const uint32_t cnt = acc - s;
                uint32_t inc = cnt;  // inclusive prefix over the lanes
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t = __shfl_up(inc, (unsigned)d, 64);
                    if (lane >= d) inc += t;
                }
                const uint32_t ns = a0 + inc - cnt;
                uint32_t ch = ns != s ? 1u : 0u;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) ch |= __shfl_xor(ch, m, 64);
                s = ns;
                if (!ch) break;
            }
            uint32_t acc = s, last_pos = 0xffffffffu;
#pragma unroll
            for (int j = 0; j < FY_TAIL_C; ++j) {
                if (valid[j] && acc < need && v[j] <= hi - acc) {
                    J[g + acc] = v[j];
                    if (acc == need - 1) last_pos = (uint32_t)(lane * FY_TAIL_C + j);
                    ++acc;
                }
            }
            const uint32_t tot = __shfl(acc, 63, 64);  // a0 + the block's accepts
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const uint32_t o = __shfl_xor(last_pos, m, 64);
                last_pos = o < last_pos ? o : last_pos;
            }
            if (tot >= need) {  // the range ends inside this block: the next one starts at the following word
                w += (unsigned long long)last_pos + 1ull;
                a0 = need;
            } else {
                a0 = tot;
                w += 64ull * FY_TAIL_C;
            }
        }
        if (starved) break;
        g += need;
        i = lo - 1;
    }
    if (lane == 0) *consumed = starved ? 0xffffffffu : (uint32_t)(w - w0);
}

__global__ __launch_bounds__(256) void k_fy_iota(uint32_t *R, uint32_t n) {
    for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) R[p] = p;
}

// R[p] = the LAST step (smallest i' > p) that wrote position p, if any
__global__ __launch_bounds__(256) void k_fy_pred(const uint32_t *key, const uint32_t *val, uint32_t m, uint32_t n,
                                                 uint32_t *R) {
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < m; e += gridDim.x * 256) {
        const uint32_t k = key[e];
        if (k != n && (e + 1 == m || key[e + 1] != k)) R[k] = val[e];
    }
}

// X[p] = the value standing at p when step p runs = the end of the chain p -> R[p] -> R[R[p]] -> ...
// (R[q] > q unless q is terminal, R[q] == q).  The chains are short (a position is the target of ln(n/p) later steps on average
// and the chain hops to ever larger indices), so whoever needs X[p] simply walks p's chain.
// Round 6: walked ON DEMAND by k_fy_final.  Rounds 1-5 resolved every position into an array X first (k_fy_resolve: n walks of
// ~2 random reads each + 8 n bytes of streaming) and k_fy_final then read X[.] at random once more; but X is only needed for
// the steps that are not the FIRST writer of their target -- about half of them (a position p is never written with
// probability (p + 1) / n) -- so the fused form makes ~n random reads instead of ~2.5 n.
__device__ __forceinline__ uint32_t fy_walk(const uint32_t *R, uint32_t p) {
    uint32_t r = R[p];
    while (r != p) {
        p = r;
        r = R[p];
    }
    return p;
}

__global__ __launch_bounds__(256) void k_fy_final(const uint32_t *key, const uint32_t *val, uint32_t m, uint32_t n,
                                                  const uint32_t *R, int64_t *perm) {
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < m; e += gridDim.x * 256) {
        const uint32_t k = key[e], i = val[e];
        uint32_t out;
        if (k == n) out = fy_walk(R, i);                         // swap with itself: what stands at i when step i runs
        else if (e == 0 || key[e - 1] != k) out = k;             // first writer of p takes p's own value
        else out = fy_walk(R, val[e - 1]);                       // ... the next one what the previous moved in
        perm[i] = (int64_t)out;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) perm[0] = (int64_t)fy_walk(R, 0u);
}

__global__ __launch_bounds__(256) void k_fy_rng_finalize(slk_rng_dev *st, const uint32_t *raw, unsigned long long t_last) {
    const unsigned long long blk = t_last / SLK_MT_N;
    for (int i = threadIdx.x; i < SLK_MT_N; i += 256) st->key[i] = raw[blk * SLK_MT_N + i];
    __syncthreads();
    if (threadIdx.x == 0) st->pos = (int32_t)(t_last % SLK_MT_N) + 1;
}

template <class T>
__global__ __launch_bounds__(256) void k_gather_rows(const T *src, const int64_t *perm, int64_t n, int64_t row_len,
                                                     T *dst) {
    const int64_t total = n * row_len;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / row_len, c = e - r * row_len;
        dst[e] = src[perm[r] * row_len + c];
    }
}

// row_len == 1 (the id and rating arrays of fit()): no index arithmetic -- the 64-bit division per element of the general
// kernel above cost more than the gather's memory traffic
template <class T>
__global__ __launch_bounds__(256) void k_gather_elems(const T *src, const int64_t *perm, int64_t n, T *dst) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) dst[e] = src[perm[e]];
}

// fit()'s two id arrays through ONE permutation (torch_utils.py:35-52 applies the same shuffle_indices to every array): the ids
// are packed once per fit() into (user, item) pairs of 32 bits each, so a shuffled interaction costs one random 8-byte read
// (one 64-byte sector) instead of two, and the permutation is read once.  Four interactions in flight per thread.
__global__ __launch_bounds__(256) void k_pack_id_pairs(const int64_t *users, const int64_t *items, int64_t n, uint2 *pairs) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        pairs[e] = make_uint2((uint32_t)users[e], (uint32_t)items[e]);
}

__global__ __launch_bounds__(256) void k_gather_id_pairs(const uint2 *pairs, const int64_t *perm, int64_t n, int64_t *users,
                                                         int64_t *items) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; e + 3 * stride < n; e += 4 * stride) {
        int64_t p[4];
        uint2 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = perm[e + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = pairs[p[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            users[e + k * stride] = (int64_t)v[k].x;
            items[e + k * stride] = (int64_t)v[k].y;
        }
    }
    for (; e < n; e += stride) {
        const uint2 v = pairs[perm[e]];
        users[e] = (int64_t)v.x;
        items[e] = (int64_t)v.y;
    }
}

static inline unsigned fy_grid(const slk_ctx *ctx, size_t n) {
    size_t b = (n + 255) / 256, cap = (size_t)ctx->num_cus * 16;
    if (b > cap) b = cap;
    return (unsigned)(b ? b : 1);
}

// expected raw words for the draws i = hi .. lo (one mask): sum (mask+1)/(i+1)
static double fy_expected_words(double mask, double lo, double hi) { return (mask + 1.0) * log((hi + 1.0) / lo); }

SLK_EXPORT int slk_shuffle_perm(slk_ctx *ctx, int64_t n, int64_t *d_perm_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (n < 0 || n >= ((int64_t)1 << 31)) return slk_fail(ctx, SLK_EINVAL, "slk_shuffle_perm: n %lld outside [0, 2^31)", (long long)n);
    if (n == 0) return SLK_OK;
    if (!d_perm_out) return slk_fail(ctx, SLK_EINVAL, "slk_shuffle_perm: d_perm_out is NULL");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    ctx->sampled_valid = false;  // the shuffle moves the stream position: only slk_rng_get_state (stream sync) sees its end
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const uint32_t N = (uint32_t)n;
    int rc;
    if (N == 1) {  // nothing to draw
        SLK_HIP(ctx, hipMemsetAsync(d_perm_out, 0, 8, s));
        return SLK_OK;
    }
    slk_prof_begin(ctx, SLK_K_SAMPLE, s);
    // ---- raw words: expected count of every range + 12 sigma (variance per draw <= 2) + slack
    double need_words = 0.0;
    for (uint32_t hi = N - 1; hi >= 1;) {
        uint32_t mask = hi;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        const uint32_t lo = (mask >> 1) + 1;  // 2^k
        need_words += fy_expected_words((double)mask, (double)lo, (double)hi);
        hi = lo - 1;
    }
    need_words += 12.0 * sqrt(2.0 * (double)N) + 4096.0;
    int32_t pos0 = 0;
    SLK_HIP(ctx, hipMemcpyAsync(&pos0, &ctx->d_rng->pos, sizeof(pos0), hipMemcpyDeviceToHost, s));
    SLK_HIP(ctx, hipStreamSynchronize(s));
    const unsigned long long nblocks = 1ull + (unsigned long long)((need_words + (double)pos0) / SLK_MT_N) + 1ull;
    const unsigned long long total_words = nblocks * SLK_MT_N;
    if ((rc = slk_ensure(ctx, ctx->raw, total_words * 4))) return rc;
    for (int b = 0; b < 5; ++b)
        if ((rc = slk_ensure(ctx, ctx->extra[FY_B0 + b], (size_t)N * 4 + 64))) return rc;
    const size_t nb_max = ((size_t)N * 2 + FY_TILE - 1) / FY_TILE + 2;
    if ((rc = slk_ensure(ctx, ctx->extra[FY_SMALL], nb_max * 16 + 128))) return rc;  // 2 x (counts, offsets + total), flags
    if ((rc = slk_mt_generate_blocks(ctx, nblocks, s))) return rc;
    const uint32_t *raw = (const uint32_t *)ctx->raw.p;
    uint32_t *J = (uint32_t *)ctx->extra[FY_B0].p;
    uint32_t *Abuf[2] = {(uint32_t *)ctx->extra[FY_B1].p, (uint32_t *)ctx->extra[FY_B2].p};
    // small area: cnt[nb_max] | off[nb_max + 1] (+ pad) | cnt_unc[nb_max] | off_unc[nb_max + 1] | status, consumed
    uint32_t *cnt = (uint32_t *)ctx->extra[FY_SMALL].p, *off = cnt + nb_max;
    int *d_flag = (int *)(off + nb_max + 4 + nb_max + nb_max + 4);
    uint32_t *d_consumed = (uint32_t *)(d_flag + 1);

    // ---- (1) the draws, range by range (constant mask), largest i first
    unsigned long long w = (unsigned long long)pos0;  // next unconsumed word
    uint32_t g = 0;                                   // draws emitted so far (draw g <-> i = N-1-g)
    uint32_t hi = N - 1;
    int total_sweeps = 0;
    ctx->fy_fallbacks = 0;
    while (hi >= FY_TAIL) {
        uint32_t mask = hi;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t lo = (mask >> 1) + 1;
        if (lo < FY_TAIL) lo = FY_TAIL;
        const uint32_t need = hi - lo + 1;
        double wd = fy_expected_words((double)mask, (double)lo, (double)hi) + 12.0 * sqrt(2.0 * (double)need) + 64.0;
        if (wd > (double)(total_words - w)) wd = (double)(total_words - w);
        if (wd >= 4294967295.0) return slk_fail(ctx, SLK_EINVAL, "slk_shuffle_perm: window too large");
        fy_args a;
        memset(&a, 0, sizeof(a));
        a.raw = raw;
        a.w0 = w;
        a.W = (uint32_t)wd;
        a.mask = mask;
        a.hi = hi;
        a.need = need;
        const unsigned nb = (a.W + FY_TILE - 1) / FY_TILE;
        uint32_t consumed = 0;
        bool done = false;
        if (ctx->opt_shuffle_band) {
            // ---- banded: classify + count, compact the uncertain words, resolve them, emit + verify (one host round trip)
            const int band_div = ctx->opt_shuffle_band;
            uint32_t *cnt_unc = off + nb_max + 4, *off_unc = cnt_unc + nb_max;  // second counter / offset pair
            int *u_thr = (int *)Abuf[0];
            uint32_t *u_cert = Abuf[1], *u_acc = (uint32_t *)ctx->extra[FY_B3].p;
            SLK_HIP(ctx, hipMemsetAsync(d_flag, 0, 8, s));  // status + consumed
            hipLaunchKernelGGL(k_fyb_count, dim3(nb), dim3(256), 0, s, a, band_div, cnt, cnt_unc);
            SLK_LAUNCH_CHECK(ctx, "k_fyb_count");
            hipLaunchKernelGGL(k_fy_scan, dim3(1), dim3(256), 0, s, (const uint32_t *)cnt, off, (int)nb);
            hipLaunchKernelGGL(k_fy_scan, dim3(1), dim3(256), 0, s, (const uint32_t *)cnt_unc, off_unc, (int)nb);
            hipLaunchKernelGGL(k_fyb_collect, dim3(nb), dim3(256), 0, s, a, band_div, (const uint32_t *)off,
                               (const uint32_t *)off_unc, u_thr, u_cert);
            SLK_LAUNCH_CHECK(ctx, "k_fyb_collect");
            hipLaunchKernelGGL(k_fyb_resolve, dim3(1), dim3(1024), 0, s, (const uint32_t *)off_unc, (int)nb, (const int *)u_thr,
                               (const uint32_t *)u_cert, u_acc, d_flag);
            SLK_LAUNCH_CHECK(ctx, "k_fyb_resolve");
            hipLaunchKernelGGL(k_fyb_emit, dim3(nb), dim3(256), 0, s, a, band_div, (const uint32_t *)off,
                               (const uint32_t *)off_unc, (const int *)u_thr, (const uint32_t *)u_cert, (const uint32_t *)u_acc, J,
                               g, d_consumed, d_flag);
            SLK_LAUNCH_CHECK(ctx, "k_fyb_emit");
            int h2[2] = {0, 0};
            SLK_HIP(ctx, hipMemcpyAsync(h2, d_flag, 8, hipMemcpyDeviceToHost, s));
            SLK_HIP(ctx, hipStreamSynchronize(s));
            consumed = (uint32_t)h2[1];
            done = h2[0] == 0 && consumed != 0;  // otherwise: the band was left (or the window ran out) -- full sweeps decide
            if (!done) ++ctx->fy_fallbacks;
        }
        if (!done) {
            int cur = 0;
            hipLaunchKernelGGL(k_fy_init, dim3(fy_grid(ctx, a.W)), dim3(256), 0, s, Abuf[0], a.W, need, hi, mask);
            SLK_LAUNCH_CHECK(ctx, "k_fy_init");
            const int check_every = a.W > (1u << 20) ? 1 : 3;
            int sweeps = 0, flag = 1;
            a.A_old = Abuf[0];
            hipLaunchKernelGGL(k_fy_count, dim3(nb), dim3(256), 0, s, a, cnt);  // later counts come from k_fy_apply
            SLK_LAUNCH_CHECK(ctx, "k_fy_count");
            while (flag) {
                for (int k = 0; k < check_every; ++k) {
                    a.A_old = Abuf[cur];
                    a.A_new = Abuf[cur ^ 1];
                    if (k == check_every - 1) SLK_HIP(ctx, hipMemsetAsync(d_flag, 0, sizeof(int), s));
                    hipLaunchKernelGGL(k_fy_scan, dim3(1), dim3(256), 0, s, (const uint32_t *)cnt, off, (int)nb);
                    hipLaunchKernelGGL(k_fy_apply, dim3(nb), dim3(256), 0, s, a, (const uint32_t *)off, cnt, d_flag);
                    SLK_LAUNCH_CHECK(ctx, "k_fy_apply");
                    cur ^= 1;
                    ++sweeps;
                }
                SLK_HIP(ctx, hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
                SLK_HIP(ctx, hipStreamSynchronize(s));
                if (sweeps > 2000) return slk_fail(ctx, SLK_EIO, "slk_shuffle_perm: acceptance fixpoint did not converge");
            }
            total_sweeps += sweeps;
            a.A_old = Abuf[cur];
            SLK_HIP(ctx, hipMemsetAsync(d_consumed, 0, 4, s));
            hipLaunchKernelGGL(k_fy_emit, dim3(fy_grid(ctx, a.W)), dim3(256), 0, s, a, J, g, d_consumed);
            SLK_LAUNCH_CHECK(ctx, "k_fy_emit");
            SLK_HIP(ctx, hipMemcpyAsync(&consumed, d_consumed, 4, hipMemcpyDeviceToHost, s));
            SLK_HIP(ctx, hipStreamSynchronize(s));
        }
        if (consumed == 0)
            return slk_fail(ctx, SLK_EIO, "slk_shuffle_perm: ran out of generated words (rejection tail > 12 sigma)");
        w += consumed;
        g += need;
        hi = lo - 1;
    }
    if (hi >= 1) {
        uint32_t consumed = 0;
        hipLaunchKernelGGL(k_fy_tail, dim3(1), dim3(64), 0, s, raw, w, total_words, hi, J, g, d_consumed);
        SLK_LAUNCH_CHECK(ctx, "k_fy_tail");
        SLK_HIP(ctx, hipMemcpyAsync(&consumed, d_consumed, 4, hipMemcpyDeviceToHost, s));
        SLK_HIP(ctx, hipStreamSynchronize(s));
        if (consumed == 0xffffffffu)
            return slk_fail(ctx, SLK_EIO, "slk_shuffle_perm: ran out of generated words (rejection tail > 12 sigma)");
        w += consumed;
    }
    ctx->fy_sweeps = total_sweeps;
    hipLaunchKernelGGL(k_fy_rng_finalize, dim3(1), dim3(256), 0, s, ctx->d_rng, raw, w - 1);
    SLK_LAUNCH_CHECK(ctx, "k_fy_rng_finalize");
    slk_prof_end(ctx, s);

    // ---- (2) the swaps
    slk_prof_begin(ctx, SLK_K_PREP, s);
    const uint32_t m = N - 1;
    uint32_t *key0 = (uint32_t *)ctx->extra[FY_B1].p, *val0 = (uint32_t *)ctx->extra[FY_B2].p;
    uint32_t *key1 = (uint32_t *)ctx->extra[FY_B3].p, *val1 = (uint32_t *)ctx->extra[FY_B4].p;
    // (target position, step) pairs sorted by target, stable; the sort's first pass forms them from the draws (k_fy_keys is gone)
    // own temporary storage: the shuffle may be prepared on another stream than the training passes
    {
        uint32_t *const keys[2] = {key0, key1}, *const vals[2] = {val0, val1};
        if ((rc = slk_sort_fy_steps(ctx, ctx->extra[FY_SORT], (const uint32_t *)J, N, keys, vals, s))) return rc;
    }
    uint32_t *R = key0;  // the sort's inputs are free again
    hipLaunchKernelGGL(k_fy_iota, dim3(fy_grid(ctx, N)), dim3(256), 0, s, R, N);
    hipLaunchKernelGGL(k_fy_pred, dim3(fy_grid(ctx, m)), dim3(256), 0, s, (const uint32_t *)key1, (const uint32_t *)val1,
                       m, N, R);
    SLK_LAUNCH_CHECK(ctx, "k_fy_pred");
    hipLaunchKernelGGL(k_fy_final, dim3(fy_grid(ctx, m)), dim3(256), 0, s, (const uint32_t *)key1, (const uint32_t *)val1,
                       m, N, (const uint32_t *)R, d_perm_out);
    SLK_LAUNCH_CHECK(ctx, "k_fy_final");
    slk_prof_end(ctx, s);
    return SLK_OK;
}

SLK_EXPORT int slk_gather_rows_i64(slk_ctx *ctx, const int64_t *d_src, const int64_t *d_perm, int64_t n,
                                   int64_t row_len, int64_t *d_dst, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (n < 0 || row_len < 1 || (n > 0 && (!d_src || !d_perm || !d_dst)))
        return slk_fail(ctx, SLK_EINVAL, "slk_gather_rows_i64: bad arguments");
    if (n == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    if (row_len == 1)
        hipLaunchKernelGGL((k_gather_elems<int64_t>), dim3(fy_grid(ctx, (size_t)n)), dim3(256), 0, s, d_src, d_perm, n, d_dst);
    else
        hipLaunchKernelGGL((k_gather_rows<int64_t>), dim3(fy_grid(ctx, (size_t)(n * row_len))), dim3(256), 0, s, d_src, d_perm,
                           n, row_len, d_dst);
    SLK_LAUNCH_CHECK(ctx, "k_gather_rows");
    return SLK_OK;
}

SLK_EXPORT int slk_pack_id_pairs(slk_ctx *ctx, const int64_t *d_users, const int64_t *d_items, int64_t n, uint32_t *d_pairs,
                                 void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (n < 0 || (n > 0 && (!d_users || !d_items || !d_pairs))) return slk_fail(ctx, SLK_EINVAL, "slk_pack_id_pairs: bad arguments");
    if (n == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    hipLaunchKernelGGL(k_pack_id_pairs, dim3(fy_grid(ctx, (size_t)n)), dim3(256), 0, s, d_users, d_items, n, (uint2 *)d_pairs);
    SLK_LAUNCH_CHECK(ctx, "k_pack_id_pairs");
    return SLK_OK;
}

SLK_EXPORT int slk_gather_id_pairs(slk_ctx *ctx, const uint32_t *d_pairs, const int64_t *d_perm, int64_t n, int64_t *d_users_out,
                                   int64_t *d_items_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (n < 0 || (n > 0 && (!d_pairs || !d_perm || !d_users_out || !d_items_out)))
        return slk_fail(ctx, SLK_EINVAL, "slk_gather_id_pairs: bad arguments");
    if (n == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    hipLaunchKernelGGL(k_gather_id_pairs, dim3(fy_grid(ctx, (size_t)n)), dim3(256), 0, s, (const uint2 *)d_pairs, d_perm, n,
                       d_users_out, d_items_out);
    SLK_LAUNCH_CHECK(ctx, "k_gather_id_pairs");
    return SLK_OK;
}
