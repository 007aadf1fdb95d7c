// slk_probe.hip -- bandwidth probes of the device the engine runs on (measurement support, no reference
// counterpart; SURVEY.md 8(d) asks for the measured stream figure next to the nominal 8 TB/s, and
// VERDICT r01 item 4 for "a micro-kernel that performs exactly the algorithmic accesses (no records, no
// sorts)" as the ceiling of the fused step).
//
//   slk_probe_stream        float4 copy / triad over caller-supplied buffers: the achievable streaming rate
//   slk_probe_step_ceiling  the BilinearNet step's ALGORITHMIC row accesses and nothing else, on the real
//                           tables, in the passes' own lane layout (one row per G = D/4 lanes, 16 B per lane):
//                             user side  per interaction: U[u] + state read, V[pos], V[neg] read, dot products,
//                                        U[u] + state written back (users ascending, as the sorted pass sees them)
//                             item side  per touched item row: V[i] + state read and written back (ascending;
//                                        the expected number of distinct items of 2B uniform draws)
//                           No sort, no record, no key/payload stream, no bias traffic, no duplicate handling:
//                           what an exact implementation could reach if grouping and the user->item hand-over
//                           were free.  Values are written back unchanged (x + 0 * dot), so the probe can run
//                           on live tables.
#include "slk_kernels.h"

__global__ __launch_bounds__(256) void k_probe_copy(float4 *__restrict__ a, const float4 *__restrict__ b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) a[i] = b[i];
}

__global__ __launch_bounds__(256) void k_probe_triad(float4 *__restrict__ a, const float4 *__restrict__ b,
                                                     const float4 *__restrict__ c, float s, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}

// streaming variants: 4 independent 16-B accesses per lane per iteration, non-temporal (nothing is re-read)
// kind 2: copy, 3: triad, 4: read only (the sum is kept live through a never-taken store), 5: write only
template <int KIND>
__global__ __launch_bounds__(256) void k_probe_stream4(float *__restrict__ a, const float *__restrict__ b,
                                                       const float *__restrict__ c, float s, size_t n4) {
    const size_t tile = (size_t)gridDim.x * 256;
    slk_vec<4> acc = slk_vzero<4>();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * tile) {
        slk_vec<4> x[4], y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t j = i + k * tile;
            x[k] = y[k] = acc;
            if (KIND != 5 && j < n4) x[k] = slk_vload_nt<4>(b + 4 * j);
            if (KIND == 3 && j < n4) y[k] = slk_vload_nt<4>(c + 4 * j);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t j = i + k * tile;
            if (j >= n4) continue;
            if (KIND == 3) slk_vaxpy<4>(x[k], s, y[k]);
            if (KIND == 4) slk_vaxpy<4>(acc, 1.0f, x[k]);
            if (KIND != 4) slk_vstore_nt<4>(a + 4 * j, x[k]);
        }
    }
    if (KIND == 4 && acc.v[0] + acc.v[1] + acc.v[2] + acc.v[3] == 12345.678f) slk_vstore<4>(a, acc);
}

// copy, chunked: a workgroup moves contiguous chunks of 256 * UNROLL float4 (every load of a chunk issued before its first
// store; a wavefront's 64 lanes cover 1 KB per access), chunks dealt round-robin to the workgroups.  kinds 6..11 of
// slk_probe_stream: (UNROLL, non-temporal loads, non-temporal stores, workgroups per CU)
template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_probe_copy_chunk(float *__restrict__ a, const float *__restrict__ b, size_t n4) {
    const size_t chunk = (size_t)256 * UNROLL;
    for (size_t c0 = (size_t)blockIdx.x * chunk; c0 < n4; c0 += (size_t)gridDim.x * chunk) {
        slk_vec<4> x[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const size_t j = c0 + (size_t)k * 256 + threadIdx.x;
            if (j < n4) x[k] = NTL ? slk_vload_nt<4>(b + 4 * j) : slk_vload<4>(b + 4 * j);
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const size_t j = c0 + (size_t)k * 256 + threadIdx.x;
            if (j < n4) {
                if (NTS) slk_vstore_nt<4>(a + 4 * j, x[k]);
                else slk_vstore<4>(a + 4 * j, x[k]);
            }
        }
    }
}

__device__ __forceinline__ uint32_t slk_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

struct slk_probe_args {
    float *U, *SU, *V, *SV;
    uint32_t n_users, n_items, batch;
    uint32_t ustride;     // users ascend: user(p) = p * ustride + mix(p) % ustride
    uint32_t item_thresh; // item i is "touched" when mix(i ^ salt) < item_thresh
    uint32_t salt;
    int D;
    int nt;
};

// user side: one row group per interaction
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_probe_user_side(slk_probe_args a) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int D = a.D, d0 = lane * VEC;
    const bool on = d0 < D;
    const bool nt = (a.nt & 1) != 0;
    for (uint32_t p = blockIdx.x * GPB + grp; p < a.batch; p += gridDim.x * GPB) {
        const uint32_t h = slk_mix32(p ^ a.salt);
        uint32_t user = p * a.ustride + h % a.ustride;
        if (user >= a.n_users) user = a.n_users - 1;
        const uint32_t ip = slk_mix32(h + 0x9e3779b9u) % a.n_items, in = slk_mix32(h + 0x3c6ef372u) % a.n_items;
        if (!on) continue;
        const size_t uoff = (size_t)user * D + d0;
        slk_vec<VEC> u = slk_vload_if_nt<VEC>(a.U + uoff, nt);
        slk_vec<VEC> s = slk_vload_if_nt<VEC>(a.SU + uoff, nt);
        const slk_vec<VEC> vi = slk_vload<VEC>(a.V + (size_t)ip * D + d0);
        const slk_vec<VEC> vj = slk_vload<VEC>(a.V + (size_t)in * D + d0);
        const float dp = slk_group_sum<G>(slk_vdot<VEC>(u, vi)), dn = slk_group_sum<G>(slk_vdot<VEC>(u, vj));
        const float z = (dp - dn) * 0.0f;  // not foldable (NaN / inf semantics): the stores below stay
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            u.v[i] += z;
            s.v[i] += z;
        }
        slk_vstore_if_nt<VEC>(a.SU + uoff, s, nt);
        slk_vstore_if_nt<VEC>(a.U + uoff, u, nt);
    }
}

// item side: one row group per candidate item, touched ones only
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_probe_item_side(slk_probe_args a) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int D = a.D, d0 = lane * VEC;
    const bool on = d0 < D;
    const bool nt = (a.nt & 2) != 0;
    for (uint32_t i = blockIdx.x * GPB + grp; i < a.n_items; i += gridDim.x * GPB) {
        if (slk_mix32(i ^ a.salt) >= a.item_thresh || !on) continue;
        const size_t off = (size_t)i * D + d0;
        slk_vec<VEC> v = slk_vload_if_nt<VEC>(a.V + off, nt);
        slk_vec<VEC> s = slk_vload_if_nt<VEC>(a.SV + off, nt);
        const float z = slk_group_sum<G>(slk_vdot<VEC>(v, s)) * 0.0f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v.v[e] += z;
            s.v[e] += z;
        }
        slk_vstore_if_nt<VEC>(a.SV + off, s, nt);
        slk_vstore_if_nt<VEC>(a.V + off, v, nt);
    }
}

// Random-row regime (the item table of the 1B-item configuration: 125M rows per GPU, every touched row in its own DRAM
// page): row + optimizer-state read-modify-write with the two kept in SEPARATE tables (the layout torch's parameter /
// state tensors have) or INTERLEAVED as [row | state] records of 2 * D floats (one location per update instead of two).
struct slk_rows_args {
    float *buf;
    uint64_t rows;
    uint64_t n_access;
    uint64_t stride;  // ascending order: row(k) = k * stride + mix(k) % stride
    int D, layout, order, rmw, nt;
    uint32_t salt;
};

template <int VEC, int G>
__global__ __launch_bounds__(256) void k_probe_random_rows(slk_rows_args a) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int D = a.D, d0 = lane * VEC;
    const bool on = d0 < D;
    const bool nt = a.nt != 0;
    float acc = 0.0f;
    for (uint64_t k = (uint64_t)blockIdx.x * GPB + grp; k < a.n_access; k += (uint64_t)gridDim.x * GPB) {
        const uint32_t h = slk_mix32((uint32_t)k ^ a.salt), h2 = slk_mix32(h + 0x9e3779b9u);
        uint64_t row = a.order == 0 ? k * a.stride + h % a.stride : (((uint64_t)h << 32) | h2) % a.rows;
        if (row >= a.rows) row = a.rows - 1;
        if (!on) continue;
        float *pp = a.layout ? a.buf + row * 2 * D + d0 : a.buf + row * D + d0;
        float *ps = a.layout ? pp + D : a.buf + (a.rows + row) * D + d0;
        slk_vec<VEC> p = slk_vload_if_nt<VEC>(pp, nt);
        slk_vec<VEC> s = slk_vload_if_nt<VEC>(ps, nt);
        const float z = slk_group_sum<G>(slk_vdot<VEC>(p, s)) * 0.0f;
        if (a.rmw) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                p.v[i] += z;
                s.v[i] += z;
            }
            slk_vstore_if_nt<VEC>(ps, s, nt);
            slk_vstore_if_nt<VEC>(pp, p, nt);
        } else {
            acc += z;
        }
    }
    if (!a.rmw && acc == 12345.678f) a.buf[0] = acc;
}

// average duration of `iters` back-to-back launches of fn(i) on stream s (one untimed launch first)
template <typename F>
static int probe_time(slk_ctx *ctx, int iters, hipStream_t s, double *avg_ms, F launch) {
    hipEvent_t e0, e1;
    SLK_HIP(ctx, hipEventCreate(&e0));
    SLK_HIP(ctx, hipEventCreate(&e1));
    launch();
    SLK_HIP(ctx, hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch();
    SLK_HIP(ctx, hipEventRecord(e1, s));
    SLK_HIP(ctx, hipEventSynchronize(e1));
    hipError_t e = hipGetLastError();
    float ms = 0.0f;
    SLK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return slk_fail(ctx, SLK_EIO, "probe launch failed: %s", hipGetErrorString(e));
    *avg_ms = (double)ms / iters;
    return SLK_OK;
}

SLK_EXPORT int slk_probe_stream(slk_ctx *ctx, int32_t kind, float *d_a, const float *d_b, const float *d_c,
                                int64_t n_floats, int32_t iters, double *avg_ms, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (!d_a || !d_b || ((kind == 1 || kind == 3) && !d_c) || n_floats < 4 || (n_floats & 3) || iters < 1 || !avg_ms || kind < 0 ||
        kind > 11)
        return slk_fail(ctx, SLK_EINVAL, "slk_probe_stream: bad arguments");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = (size_t)n_floats / 4;
    const unsigned grid = (unsigned)ctx->num_cus * 8u;
    const unsigned grid4 = (unsigned)ctx->num_cus * 16u;
    return probe_time(ctx, iters, s, avg_ms, [&]() {
        float *a4 = d_a;
        const float *b4 = d_b, *c4 = d_c;
        switch (kind) {
            case 0: hipLaunchKernelGGL(k_probe_copy, dim3(grid), dim3(256), 0, s, (float4 *)d_a, (const float4 *)d_b, n4); break;
            case 1:
                hipLaunchKernelGGL(k_probe_triad, dim3(grid), dim3(256), 0, s, (float4 *)d_a, (const float4 *)d_b,
                                   (const float4 *)d_c, 0.5f, n4);
                break;
            case 2: hipLaunchKernelGGL(k_probe_stream4<2>, dim3(grid4), dim3(256), 0, s, a4, b4, c4, 0.5f, n4); break;
            case 3: hipLaunchKernelGGL(k_probe_stream4<3>, dim3(grid4), dim3(256), 0, s, a4, b4, c4, 0.5f, n4); break;
            case 4: hipLaunchKernelGGL(k_probe_stream4<4>, dim3(grid4), dim3(256), 0, s, a4, b4, c4, 0.5f, n4); break;
            case 5: hipLaunchKernelGGL(k_probe_stream4<5>, dim3(grid4), dim3(256), 0, s, a4, b4, c4, 0.5f, n4); break;
            case 6: hipLaunchKernelGGL((k_probe_copy_chunk<8, false, false>), dim3(grid), dim3(256), 0, s, a4, b4, n4); break;
            case 7: hipLaunchKernelGGL((k_probe_copy_chunk<8, true, true>), dim3(grid), dim3(256), 0, s, a4, b4, n4); break;
            case 8: hipLaunchKernelGGL((k_probe_copy_chunk<4, false, false>), dim3(grid4), dim3(256), 0, s, a4, b4, n4); break;
            case 9: hipLaunchKernelGGL((k_probe_copy_chunk<16, false, false>), dim3(grid / 2), dim3(256), 0, s, a4, b4, n4); break;
            case 10: hipLaunchKernelGGL((k_probe_copy_chunk<8, true, false>), dim3(grid), dim3(256), 0, s, a4, b4, n4); break;
            default: hipLaunchKernelGGL((k_probe_copy_chunk<4, true, true>), dim3(grid4 * 2), dim3(256), 0, s, a4, b4, n4); break;
        }
    });
}

SLK_EXPORT int slk_probe_step_ceiling(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, int64_t batch,
                                      int32_t iters, double *user_ms, double *item_ms, int64_t *items_touched,
                                      void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, tables, 3u, &vec, &g))) return rc;
    if (!optim || !optim->d_state1[0] || !optim->d_state1[1] || batch < 1 || batch >= ((int64_t)1 << 31) || iters < 1 ||
        !user_ms || !item_ms || tables->user_bloom || tables->item_bloom)
        return slk_fail(ctx, SLK_EINVAL, "slk_probe_step_ceiling: bad arguments (plain tables + first optimizer state)");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    slk_probe_args a;
    memset(&a, 0, sizeof(a));
    a.U = tables->d_param[0];
    a.V = tables->d_param[1];
    a.SU = optim->d_state1[0];
    a.SV = optim->d_state1[1];
    a.n_users = (uint32_t)tables->num_users;
    a.n_items = (uint32_t)tables->num_items;
    a.batch = (uint32_t)batch;
    a.ustride = (uint32_t)(tables->num_users / batch);
    if (a.ustride < 1) a.ustride = 1;
    // expected distinct items among 2B uniform draws over I: I * (1 - exp(-2B / I))
    const double frac = 1.0 - exp(-2.0 * (double)batch / (double)tables->num_items);
    a.item_thresh = frac >= 1.0 ? 0xffffffffu : (uint32_t)(frac * 4294967296.0);
    a.D = tables->dim;
    a.nt = ctx->opt_nt;
    if (items_touched) *items_touched = (int64_t)(frac * (double)tables->num_items);
    const unsigned gpb = 256u / (unsigned)g;
    const unsigned ugrid = slk_grid_for(ctx, (size_t)batch, gpb), igrid = slk_grid_for(ctx, (size_t)tables->num_items, gpb);
    unsigned salt = 1;
#define SLK_PROBE_U(V_, G_) hipLaunchKernelGGL((k_probe_user_side<V_, G_>), dim3(ugrid), dim3(256), 0, s, a)
#define SLK_PROBE_I(V_, G_) hipLaunchKernelGGL((k_probe_item_side<V_, G_>), dim3(igrid), dim3(256), 0, s, a)
    if ((rc = probe_time(ctx, iters, s, user_ms, [&]() {
            a.salt = salt++ * 0x9e3779b9u;
            SLK_FOR_LAYOUT(vec, g, SLK_PROBE_U);
        })))
        return rc;
    return probe_time(ctx, iters, s, item_ms, [&]() {
        a.salt = salt++ * 0x9e3779b9u;
        SLK_FOR_LAYOUT(vec, g, SLK_PROBE_I);
    });
#undef SLK_PROBE_U
#undef SLK_PROBE_I
}

SLK_EXPORT int slk_probe_random_rows(slk_ctx *ctx, float *d_buf, int64_t rows, int32_t dim, int32_t layout, int32_t order,
                                     int32_t rmw, int64_t n_access, int32_t iters, double *avg_ms, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g;
    if (!d_buf || rows < 1 || n_access < 1 || iters < 1 || !avg_ms || !slk_pick_layout(dim, &vec, &g))
        return slk_fail(ctx, SLK_EINVAL, "slk_probe_random_rows: bad arguments");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    slk_rows_args a;
    memset(&a, 0, sizeof(a));
    a.buf = d_buf;
    a.rows = (uint64_t)rows;
    a.n_access = (uint64_t)n_access;
    a.stride = (uint64_t)(rows / n_access) ? (uint64_t)(rows / n_access) : 1;
    a.D = dim;
    a.layout = layout;
    a.order = order;
    a.rmw = rmw;
    a.nt = (ctx->opt_nt & 2) != 0;
    const unsigned grid = slk_grid_for(ctx, (size_t)n_access, 256u / (unsigned)g, ctx->opt_item_grid_mult);
    unsigned salt = 7;
#define SLK_PROBE_R(V_, G_) hipLaunchKernelGGL((k_probe_random_rows<V_, G_>), dim3(grid), dim3(256), 0, s, a)
    return probe_time(ctx, iters, s, avg_ms, [&]() {
        a.salt = salt++ * 0x9e3779b9u;
        SLK_FOR_LAYOUT(vec, g, SLK_PROBE_R);
    });
#undef SLK_PROBE_R
}

SLK_EXPORT int slk_probe_sort(slk_ctx *ctx, int32_t kind, const void *d_keys_in, void *d_keys_out, const void *d_vals_in,
                              void *d_vals_out, int64_t n, int64_t seg_len, int32_t bits, int32_t iters, double *avg_ms,
                              void *stream) {
    if (!ctx) return SLK_EINVAL;
    const bool clobber = (kind & 8) != 0;  // the inputs may be overwritten (they serve as the second buffer pair): iters must be 0
    kind &= 7;
    if (clobber && iters > 0) return slk_fail(ctx, SLK_EINVAL, "slk_probe_sort: a clobbering sort cannot be repeated");
    if (n < 0 || seg_len < 0 || bits < 1 || kind < 0 || kind > 2 || (n > 0 && (!d_keys_in || !d_keys_out || !d_vals_in || !d_vals_out)))
        return slk_fail(ctx, SLK_EINVAL, "slk_probe_sort: bad arguments");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    int rc;
    if ((rc = slk_sort_pairs_any(ctx, kind, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, (size_t)seg_len, (unsigned)bits,
                                 s, clobber)))
        return rc;
    if (iters > 0 && avg_ms) {
        hipEvent_t e0, e1;
        SLK_HIP(ctx, hipEventCreate(&e0));
        SLK_HIP(ctx, hipEventCreate(&e1));
        SLK_HIP(ctx, hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i)
            if ((rc = slk_sort_pairs_any(ctx, kind, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, (size_t)seg_len,
                                         (unsigned)bits, s, false)))
                return rc;
        SLK_HIP(ctx, hipEventRecord(e1, s));
        SLK_HIP(ctx, hipEventSynchronize(e1));
        float ms = 0.0f;
        SLK_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = (double)ms / iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return SLK_OK;
}
