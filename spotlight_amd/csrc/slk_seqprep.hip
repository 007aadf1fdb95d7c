// slk_seqprep.hip -- Interactions.to_sequence on the device (spotlight/interactions.py:170-266).
//
// The reference sorts the interactions with np.lexsort((timestamps, user_ids)) -- by user, ties by
// timestamp, stable -- and then walks every user's history in a Python double loop, emitting
// left-zero-padded windows of max_sequence_length items that END at positions count, count - step,
// count - 2*step, ... of the history (newest window first), optionally dropping windows shorter than
// min_sequence_length.  Here:
//   (1) timestamps -> order-preserving unsigned keys (min / max found on the way, so only the bits in
//       use are sorted); stable radix sort by timestamp, then by user: np.lexsort's order exactly;
//   (2) segment heads compacted (one tile scan), windows per user prefix-summed (a second one):
//       row r of the output belongs to user segment s = upper_bound(rowoff, r) - 1, window
//       w = r - rowoff[s];
//   (3) one 16-lane group per output row writes it (64 B stores): 4 B read + 4 B written per cell.
// Everything is integer work bound by HBM bandwidth; three host synchronisations (key range, number
// of users present, number of rows).
#include "slk_common.h"

enum { TS_T0 = 26, TS_T1, TS_T2, TS_T3, TS_T4, TS_SMALL, TS_SORT, TS_USERS = 33, TS_ITEMS, TS_HEADS, TS_ROWOFF };
// ctx->extra slots: 26..32 are temporaries shared with slk_shuffle.hip (nothing survives a call);
// 33..36 hold the plan (sorted users, sorted items, segment heads, row offsets) until slk_to_sequence_fill.

#define TS_TILE 2048  // elements per workgroup of the scan kernels (256 threads x 8)

static int ts_grid(slk_ctx *ctx, uint64_t n, int per_block = 256) {
    uint64_t b = (n + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)ctx->num_cus * 16;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---- (1) keys -------------------------------------------------------------------------------
// kind 0: int64 timestamps; kind 1: float64 (NaN sorts last as in numpy, -0.0 == +0.0)
__device__ inline uint64_t ts_key(const void *ts, int kind, size_t i) {
    const uint64_t sign = 0x8000000000000000ull;
    uint64_t b = ((const uint64_t *)ts)[i];
    if (kind == 0) return b ^ sign;
    const double x = ((const double *)ts)[i];
    if (x != x) return ~0ull;
    if (x == 0.0) b = 0;
    return (b & sign) ? ~b : (b | sign);
}

__global__ __launch_bounds__(256) void k_ts_range(const void *ts, int kind, size_t n, unsigned long long *minmax) {
    __shared__ unsigned long long lo[256], hi[256];
    unsigned long long mn = ~0ull, mx = 0ull;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned long long k = ts_key(ts, kind, i);
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
    }
    lo[threadIdx.x] = mn;
    hi[threadIdx.x] = mx;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            if (lo[threadIdx.x + d] < lo[threadIdx.x]) lo[threadIdx.x] = lo[threadIdx.x + d];
            if (hi[threadIdx.x + d] > hi[threadIdx.x]) hi[threadIdx.x] = hi[threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMin(&minmax[0], lo[0]);
        atomicMax(&minmax[1], hi[0]);
    }
}

// rebased keys (key - min) in the width the range needs, and the identity payload
__global__ __launch_bounds__(256) void k_ts_rebase(const void *ts, int kind, size_t n, unsigned long long base,
                                                    uint32_t *k32, unsigned long long *k64, uint32_t *idx) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned long long k = ts_key(ts, kind, i) - base;
        if (k32) k32[i] = (uint32_t)k;
        if (k64) k64[i] = k;
        idx[i] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_ts_take_users(const int64_t *users, const uint32_t *order, size_t n, uint32_t *ku) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        ku[i] = (uint32_t)users[order[i]];
}

__global__ __launch_bounds__(256) void k_ts_take_items(const int64_t *items, const uint32_t *order, size_t n, int32_t *it) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        it[i] = (int32_t)items[order[i]];
}

// ---- (2) two tile scans over one template -----------------------------------------------------
// HEADS: value(k) = 1 where sorted user k starts a segment; the pass compacts the head positions.
// ROWS:  value(s) = windows kept for segment s; the pass writes their exclusive prefix sums.
struct ts_scan_args {
    const uint32_t *su;     // HEADS: sorted users [n]
    const uint32_t *heads;  // ROWS: segment heads [n + 1 entries used: nseg + 1]
    uint32_t n;             // elements scanned (HEADS: interactions, ROWS: segments)
    uint32_t step, thr;     // ROWS
    uint32_t *out;          // HEADS: heads[rank] = k ; ROWS: rowoff[s]
    uint32_t total_n;       // HEADS: n, written at heads[nseg]
    uint32_t *segid;        // HEADS, optional: segid[k] = index of the segment position k belongs to
};

template <int ROWS>
__device__ inline uint32_t ts_value(const ts_scan_args &a, uint32_t k) {
    if (k >= a.n) return 0u;
    if (!ROWS) return (k == 0 || a.su[k] != a.su[k - 1]) ? 1u : 0u;
    const uint32_t count = a.heads[k + 1] - a.heads[k];
    // windows end at count - w*step; kept while that end (= the window's length before clipping at
    // max_sequence_length) is >= thr.  thr = 1 without a minimum length: ceil(count / step).
    return count >= a.thr ? (count - a.thr) / a.step + 1u : 0u;
}

template <int ROWS>
__global__ __launch_bounds__(256) void k_ts_tile_count(ts_scan_args a, uint32_t *tilecnt) {
    __shared__ uint32_t s[256];
    const uint32_t base = blockIdx.x * TS_TILE + threadIdx.x * 8;
    uint32_t c = 0;
    for (int j = 0; j < 8; ++j) c += ts_value<ROWS>(a, base + j);
    s[threadIdx.x] = c;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) tilecnt[blockIdx.x] = s[0];
}

// exclusive scan of cnt[nb] -> off[nb], total -> *total (one workgroup)
__global__ __launch_bounds__(256) void k_ts_scan(const uint32_t *cnt, uint32_t *off, uint32_t nb, uint32_t *total) {
    __shared__ uint32_t s[256];
    __shared__ uint32_t carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 256) {
        const uint32_t i = base + t;
        const uint32_t v = (i < nb) ? cnt[i] : 0u;
        s[t] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const uint32_t x = (t >= d) ? s[t - d] : 0u;
            __syncthreads();
            s[t] += x;
            __syncthreads();
        }
        if (i < nb) off[i] = carry + s[t] - v;
        __syncthreads();
        if (t == 255) carry += s[255];
        __syncthreads();
    }
    if (t == 0) *total = carry;
}

template <int ROWS>
__global__ __launch_bounds__(256) void k_ts_tile_apply(ts_scan_args a, const uint32_t *tileoff, const uint32_t *total) {
    __shared__ uint32_t s[256];
    const int t = threadIdx.x;
    const uint32_t base = blockIdx.x * TS_TILE + t * 8;
    uint32_t v[8];
    uint32_t c = 0;
    for (int j = 0; j < 8; ++j) {
        v[j] = ts_value<ROWS>(a, base + j);
        c += v[j];
    }
    s[t] = c;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t x = (t >= d) ? s[t - d] : 0u;
        __syncthreads();
        s[t] += x;
        __syncthreads();
    }
    uint32_t run = tileoff[blockIdx.x] + s[t] - c;
    for (int j = 0; j < 8; ++j) {
        const uint32_t k = base + j;
        if (k < a.n) {
            if (ROWS)
                a.out[k] = run;
            else if (v[j])
                a.out[run] = k;
            if (!ROWS && a.segid) a.segid[k] = v[j] ? run : run - 1;
        }
        run += v[j];
    }
    if (blockIdx.x == 0 && t == 0) {
        if (ROWS)
            a.out[a.n] = *total;
        else
            a.out[*total] = a.total_n;
    }
}

// ---- (3) the windows ------------------------------------------------------------------------
#define TS_G 16     // lanes per output row
#define TS_ROWS 16  // consecutive rows per lane group: one binary search, then a linear walk over the segments
__global__ __launch_bounds__(256) void k_ts_fill(const uint32_t *su, const int32_t *it, const uint32_t *heads,
                                                  const uint32_t *rowoff, uint32_t nseg, uint32_t rows, int L, uint32_t step,
                                                  int32_t *seq, int32_t *seq_users) {
    const int lane = threadIdx.x % TS_G;
    const uint32_t grp = threadIdx.x / TS_G;
    const uint32_t nblk = (rows + TS_ROWS - 1) / TS_ROWS;
    for (uint32_t blk = blockIdx.x * (256 / TS_G) + grp; blk < nblk; blk += gridDim.x * (256 / TS_G)) {
        uint32_t r = blk * TS_ROWS;
        const uint32_t r_end = r + TS_ROWS < rows ? r + TS_ROWS : rows;
        // last segment whose first row is <= r (segments without rows share their successor's offset)
        uint32_t lo = 0, hi = nseg;  // invariant: rowoff[lo] <= r < rowoff[hi]
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (rowoff[mid] <= r)
                lo = mid;
            else
                hi = mid;
        }
        uint32_t next = rowoff[lo + 1];
        for (; r < r_end; ++r) {
            while (next <= r) next = rowoff[++lo + 1];
            const uint32_t start = heads[lo];
            const uint32_t count = heads[lo + 1] - start;
            const int64_t end = (int64_t)count - (int64_t)(r - rowoff[lo]) * step;
            int32_t *dst = seq + (size_t)r * L;
            for (int c = lane; c < L; c += TS_G) {
                const int64_t pos = end - L + c;
                dst[c] = pos >= 0 ? it[start + pos] : 0;
            }
            if (lane == 0) seq_users[r] = (int32_t)su[start];
        }
    }
}

static unsigned bits_for(unsigned long long range) {
    unsigned b = 0;
    while (range) {
        ++b;
        range >>= 1;
    }
    return b;
}

template <int ROWS>
static int ts_scan(slk_ctx *ctx, ts_scan_args a, uint32_t *d_total, uint32_t *total_out, hipStream_t s) {
    const uint32_t nb = (a.n + TS_TILE - 1) / TS_TILE;
    int rc;
    if ((rc = slk_ensure(ctx, ctx->extra[TS_T3], (size_t)(nb + 1) * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[TS_T4], (size_t)(nb + 1) * 4))) return rc;
    uint32_t *cnt = (uint32_t *)ctx->extra[TS_T3].p, *off = (uint32_t *)ctx->extra[TS_T4].p;
    hipLaunchKernelGGL(k_ts_tile_count<ROWS>, dim3(nb), dim3(256), 0, s, a, cnt);
    hipLaunchKernelGGL(k_ts_scan, dim3(1), dim3(256), 0, s, (const uint32_t *)cnt, off, nb, d_total);
    hipLaunchKernelGGL(k_ts_tile_apply<ROWS>, dim3(nb), dim3(256), 0, s, a, (const uint32_t *)off, (const uint32_t *)d_total);
    SLK_LAUNCH_CHECK(ctx, "k_ts_tile_apply");
    SLK_HIP(ctx, hipMemcpyAsync(total_out, d_total, 4, hipMemcpyDeviceToHost, s));
    SLK_HIP(ctx, hipStreamSynchronize(s));
    return SLK_OK;
}

int slk_compact_heads(slk_ctx *ctx, const uint32_t *d_sorted, uint32_t n, uint32_t *d_heads, uint32_t *d_segid,
                      uint32_t *nseg_out, hipStream_t s) {
    int rc;
    if ((rc = slk_ensure(ctx, ctx->extra[TS_SMALL], 64))) return rc;
    ts_scan_args a;
    memset(&a, 0, sizeof(a));
    a.su = d_sorted;
    a.n = n;
    a.out = d_heads;
    a.total_n = n;
    a.segid = d_segid;
    return ts_scan<0>(ctx, a, (uint32_t *)((char *)ctx->extra[TS_SMALL].p + 32), nseg_out, s);
}

SLK_EXPORT int slk_to_sequence_plan(slk_ctx *ctx, const int64_t *d_users, const int64_t *d_items, const void *d_timestamps,
                                    int32_t ts_kind, int64_t n, int64_t num_users, int32_t max_sequence_length,
                                    int32_t step_size, int32_t min_length, int64_t *num_sequences_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    ctx->ts_rows = -1;
    if (!d_users || !d_items || !d_timestamps || !num_sequences_out)
        return slk_fail(ctx, SLK_EINVAL, "slk_to_sequence_plan: NULL argument");
    if (n < 0 || n >= ((int64_t)1 << 31) - TS_TILE)
        return slk_fail(ctx, SLK_EINVAL, "slk_to_sequence_plan: n = %lld out of range", (long long)n);
    if (ts_kind != 0 && ts_kind != 1) return slk_fail(ctx, SLK_EINVAL, "slk_to_sequence_plan: ts_kind must be 0 (int64) or 1 (float64)");
    if (max_sequence_length < 1 || step_size < 1 || min_length < 1 || min_length > max_sequence_length)
        return slk_fail(ctx, SLK_EINVAL, "slk_to_sequence_plan: need max_sequence_length >= 1, step_size >= 1, 1 <= min_length <= max_sequence_length");
    if (num_users > ((int64_t)1 << 32)) return slk_fail(ctx, SLK_ERANGE, "slk_to_sequence_plan: user ids need more than 32 bits");
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    *num_sequences_out = 0;
    ctx->ts_L = max_sequence_length;
    ctx->ts_step = step_size;
    if (n == 0) {
        ctx->ts_rows = 0;
        ctx->ts_nseg = 0;
        return SLK_OK;
    }
    const size_t N = (size_t)n;
    int rc;
    slk_buf *E = ctx->extra;
    if ((rc = slk_ensure(ctx, E[TS_SMALL], 64))) return rc;
    unsigned long long *d_minmax = (unsigned long long *)E[TS_SMALL].p;
    uint32_t *d_total = (uint32_t *)((char *)E[TS_SMALL].p + 32);

    slk_prof_begin(ctx, SLK_K_PREP, s);
    // (1a) key range
    unsigned long long mm[2] = {~0ull, 0ull};
    SLK_HIP(ctx, hipMemcpyAsync(d_minmax, mm, 16, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_ts_range, dim3(ts_grid(ctx, N)), dim3(256), 0, s, d_timestamps, (int)ts_kind, N, d_minmax);
    SLK_LAUNCH_CHECK(ctx, "k_ts_range");
    SLK_HIP(ctx, hipMemcpyAsync(mm, d_minmax, 16, hipMemcpyDeviceToHost, s));
    SLK_HIP(ctx, hipStreamSynchronize(s));
    const unsigned tbits = bits_for(mm[1] - mm[0]);

    // (1b) stable sort by timestamp: order1
    if ((rc = slk_ensure(ctx, E[TS_T0], N * 4))) return rc;  // payload in
    if ((rc = slk_ensure(ctx, E[TS_T1], N * 4))) return rc;  // payload out
    uint32_t *idx = (uint32_t *)E[TS_T0].p, *order1 = (uint32_t *)E[TS_T1].p;
    if (tbits <= 32) {
        if ((rc = slk_ensure(ctx, E[TS_T2], N * 4))) return rc;
        if ((rc = slk_ensure(ctx, E[TS_T3], N * 4))) return rc;
        uint32_t *k32 = (uint32_t *)E[TS_T2].p, *k32o = (uint32_t *)E[TS_T3].p;
        hipLaunchKernelGGL(k_ts_rebase, dim3(ts_grid(ctx, N)), dim3(256), 0, s, d_timestamps, (int)ts_kind, N, mm[0], k32,
                           (unsigned long long *)nullptr, idx);
        SLK_LAUNCH_CHECK(ctx, "k_ts_rebase");
        if (tbits == 0)
            order1 = idx;  // all timestamps equal: lexsort keeps the input order
        else if ((rc = slk_sort_pairs_u32_u32_in(ctx, E[TS_SORT], k32, k32o, idx, order1, N, tbits, s, true)))
            return rc;
    } else {
        if ((rc = slk_ensure(ctx, E[TS_T2], N * 8))) return rc;
        if ((rc = slk_ensure(ctx, E[TS_T3], N * 8))) return rc;
        unsigned long long *k64 = (unsigned long long *)E[TS_T2].p, *k64o = (unsigned long long *)E[TS_T3].p;
        hipLaunchKernelGGL(k_ts_rebase, dim3(ts_grid(ctx, N)), dim3(256), 0, s, d_timestamps, (int)ts_kind, N, mm[0],
                           (uint32_t *)nullptr, k64, idx);
        SLK_LAUNCH_CHECK(ctx, "k_ts_rebase");
        if ((rc = slk_sort_pairs_u64_u32_in(ctx, E[TS_SORT], (const uint64_t *)k64, (uint64_t *)k64o, idx, order1, N, tbits, s, true)))
            return rc;
    }

    // (1c) stable sort by user: np.lexsort((timestamps, user_ids))
    if ((rc = slk_ensure(ctx, E[TS_T2], N * 4))) return rc;
    if ((rc = slk_ensure(ctx, E[TS_USERS], N * 4))) return rc;
    if ((rc = slk_ensure(ctx, E[TS_ITEMS], N * 4))) return rc;
    uint32_t *ku = (uint32_t *)E[TS_T2].p, *su = (uint32_t *)E[TS_USERS].p;
    uint32_t *order = (order1 == idx) ? (uint32_t *)E[TS_T1].p : idx;
    hipLaunchKernelGGL(k_ts_take_users, dim3(ts_grid(ctx, N)), dim3(256), 0, s, d_users, (const uint32_t *)order1, N, ku);
    SLK_LAUNCH_CHECK(ctx, "k_ts_take_users");
    unsigned ubits = num_users > 0 ? bits_for((unsigned long long)(num_users - 1)) : 32;
    if (ubits == 0) ubits = 1;
    if ((rc = slk_sort_pairs_u32_u32_in(ctx, E[TS_SORT], ku, su, order1, order, N, ubits, s))) return rc;
    int32_t *it = (int32_t *)E[TS_ITEMS].p;
    hipLaunchKernelGGL(k_ts_take_items, dim3(ts_grid(ctx, N)), dim3(256), 0, s, d_items, (const uint32_t *)order, N, it);
    SLK_LAUNCH_CHECK(ctx, "k_ts_take_items");

    // (2) segment heads, rows per segment
    if ((rc = slk_ensure(ctx, E[TS_HEADS], (N + 1) * 4))) return rc;
    uint32_t nseg = 0, rows = 0;
    ts_scan_args a;
    memset(&a, 0, sizeof(a));
    a.su = su;
    a.n = (uint32_t)N;
    a.out = (uint32_t *)E[TS_HEADS].p;
    a.total_n = (uint32_t)N;
    if ((rc = ts_scan<0>(ctx, a, d_total, &nseg, s))) return rc;
    if ((rc = slk_ensure(ctx, E[TS_ROWOFF], ((size_t)nseg + 1) * 4))) return rc;
    memset(&a, 0, sizeof(a));
    a.heads = (const uint32_t *)E[TS_HEADS].p;
    a.n = nseg;
    a.step = (uint32_t)step_size;
    a.thr = (uint32_t)min_length;
    a.out = (uint32_t *)E[TS_ROWOFF].p;
    if ((rc = ts_scan<1>(ctx, a, d_total, &rows, s))) return rc;
    slk_prof_end(ctx, s);

    ctx->ts_nseg = nseg;
    ctx->ts_rows = rows;
    *num_sequences_out = rows;
    return SLK_OK;
}

SLK_EXPORT int slk_to_sequence_fill(slk_ctx *ctx, int32_t *d_sequences, int32_t *d_sequence_users, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (ctx->ts_rows < 0) return slk_fail(ctx, SLK_EINVAL, "slk_to_sequence_fill: no plan (call slk_to_sequence_plan first)");
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const int64_t rows = ctx->ts_rows;
    ctx->ts_rows = -1;
    if (rows == 0) return SLK_OK;
    if (!d_sequences || !d_sequence_users) return slk_fail(ctx, SLK_EINVAL, "slk_to_sequence_fill: NULL output");
    slk_buf *E = ctx->extra;
    slk_prof_begin(ctx, SLK_K_PREP, s);
    hipLaunchKernelGGL(k_ts_fill, dim3(ts_grid(ctx, ((uint64_t)rows + TS_ROWS - 1) / TS_ROWS, 256 / TS_G)), dim3(256), 0, s, (const uint32_t *)E[TS_USERS].p,
                       (const int32_t *)E[TS_ITEMS].p, (const uint32_t *)E[TS_HEADS].p, (const uint32_t *)E[TS_ROWOFF].p,
                       (uint32_t)ctx->ts_nseg, (uint32_t)rows, (int)ctx->ts_L, (uint32_t)ctx->ts_step, d_sequences,
                       d_sequence_users);
    SLK_LAUNCH_CHECK(ctx, "k_ts_fill");
    slk_prof_end(ctx, s);
    return SLK_OK;
}
