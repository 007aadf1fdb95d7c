// slk_shard.hip -- row-sharded BilinearNet training step for G = world GPUs.
//
// The reference has no multi-GPU code; the north star fixes the design: user and item tables
// (with their biases and optimizer state) are row-sharded cyclically (owner(row) = row mod G,
// local row = row div G), every rank processes the interactions of the global minibatch whose
// USER it owns (so user rows are always local), and item rows move over xGMI in three RCCL
// all-to-all phases per minibatch, issued by the host between the calls below:
//
//   slk_shard_begin      negatives; sort the local interactions by user; bucket the 2n item
//                        lookups by owner -> send_ids (owner-local rows, grouped by owner),
//                        send_counts                                  [a2a #1: ids -> owners]
//   slk_shard_gather     owner side: requested rows (+ bias) -> row records
//                                                               [a2a #2: rows -> requesters]
//   slk_shard_user_pass  USER PASS of slk_bilinear.hip reading item rows from the received
//                        records; writes one gradient record g * u_old (+ g) per lookup,
//                        updates the local user rows in place      [a2a #3: grads -> owners]
//   slk_shard_item_pass  owner side: sort the received lookups by item, ITEM PASS (ROW mode):
//                        one owner group per unique row sums the records and applies the
//                        optimizer -- duplicate semantics stay exact (sum, then one update).
//
// Every forward still reads pre-step parameters and every row still gets the sum of its
// contributions before one optimizer update, exactly as on one GPU; only the summation order
// of an item row's contributions differs (by source rank).
#include <math.h>

#include "slk_kernels.h"

#define SLK_MAX_WORLD 64

// scratch slots in ctx->extra
enum { SH_OKEY0 = 0, SH_OKEY1, SH_OVAL0, SH_OVAL1, SH_VSLOT, SH_HIST };

static inline int shard_rsv(int D) { return ((D + 1 + 3) / 4) * 4; }

SLK_EXPORT int slk_shard_row_floats(int32_t dim) { return shard_rsv(dim); }

// key = user (one minibatch); value = (neg << 32) | pos so the sorted values are the item pairs
__global__ __launch_bounds__(256) void k_shard_user_keys(const int64_t *users, const int64_t *items,
                                                         const uint32_t *neg32, uint32_t n, uint32_t *key,
                                                         uint64_t *val) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        key[k] = (uint32_t)users[k];
        val[k] = ((uint64_t)neg32[k] << 32) | (uint64_t)(uint32_t)items[k];
    }
}

// lookup l = 2*q + s (sorted position q, pair s): key = owner of the item, value = l
__global__ __launch_bounds__(256) void k_shard_owner_keys(const uint32_t *uit, uint32_t nl, uint32_t world,
                                                          uint32_t *okey, uint32_t *oval,
                                                          unsigned long long *hist) {
    __shared__ unsigned h[SLK_MAX_WORLD];
    if (threadIdx.x < SLK_MAX_WORLD) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t l = blockIdx.x * 256 + threadIdx.x; l < nl; l += gridDim.x * 256) {
        const uint32_t o = uit[l] % world;
        okey[l] = o;
        oval[l] = l;
        atomicAdd(&h[o], 1u);
    }
    __syncthreads();
    if (threadIdx.x < world && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_shard_slots(const uint32_t *uit, const uint32_t *oval_sorted, uint32_t nl,
                                                     uint32_t world, int64_t *send_ids, uint32_t *vslot) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < nl; j += gridDim.x * 256) {
        const uint32_t l = oval_sorted[j];
        send_ids[j] = (int64_t)(uit[l] / world);
        vslot[l] = j;
    }
}

__global__ void k_shard_counts(const unsigned long long *hist, uint32_t world, int64_t *out) {
    if (threadIdx.x < world) out[threadIdx.x] = (int64_t)hist[threadIdx.x];
}

// owner side: record(j) = [V[id_j] (D) | bias[id_j] | pad]
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_shard_gather(const float *V, const float *bi, int D, int RSV,
                                                      const int64_t *ids, int64_t n, float *out) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    for (int64_t j = (int64_t)blockIdx.x * GPB + grp; j < n; j += (int64_t)gridDim.x * GPB) {
        const int64_t i = ids[j];
        float *rec = out + (size_t)j * RSV;
        if (on) slk_vstore<VEC>(rec + d0, slk_vload<VEC>(V + (size_t)i * D + d0));
        if (lane == 0) rec[D] = bi[i];
    }
}

// USER PASS over the exchange buffers (see k_user_pass in slk_bilinear.hip)
template <int VEC, int G, int UPD>
__global__ __launch_bounds__(256) void k_shard_user_pass(slk_pass_args a) {
    __shared__ double red[256];
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const uint32_t stride = gridDim.x * GPB;
    float loss_acc = 0.0f;

    for (uint32_t p = a.begin + blockIdx.x * GPB + grp; p < a.end; p += stride) {
        const uint32_t key = a.ukey[p];
        if (p > a.begin && a.ukey[p - 1] == key) continue;  // not the head of its user segment
        const uint32_t user = key & a.umask;
        const size_t uoff = (size_t)user * D + d0;
        slk_vec<VEC> u = on ? slk_vload<VEC>(a.P[0] + uoff) : slk_vzero<VEC>();
        const float bu = a.P[2][user];
        slk_vec<VEC> gu = slk_vzero<VEC>();
        float gbu = 0.0f;
        uint32_t q = p;
        do {
            const size_t sp_ = (size_t)a.vslot[2 * (size_t)q] * a.RSV, sn_ = (size_t)a.vslot[2 * (size_t)q + 1] * a.RSV;
            const slk_vec<VEC> vi = on ? slk_vload<VEC>(a.vrows + sp_ + d0) : slk_vzero<VEC>();
            const slk_vec<VEC> vj = on ? slk_vload<VEC>(a.vrows + sn_ + d0) : slk_vzero<VEC>();
            const float sp = slk_group_sum<G>(slk_vdot<VEC>(u, vi)) + bu + a.vrows[sp_ + D];
            const float sn = slk_group_sum<G>(slk_vdot<VEC>(u, vj)) + bu + a.vrows[sn_ + D];
            float l, gp, gn;
            slk_pair_loss(a.loss_kind, sp, sn, a.inv_b, l, gp, gn);
            slk_vec<VEC> cp, cn;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                gu.v[i] += gp * vi.v[i] + gn * vj.v[i];
                cp.v[i] = gp * u.v[i];
                cn.v[i] = gn * u.v[i];
            }
            gbu += gp + gn;
            if (on) {
                slk_vstore<VEC>(a.grows + sp_ + d0, cp);
                slk_vstore<VEC>(a.grows + sn_ + d0, cn);
            }
            if (lane == 0) {
                a.grows[sp_ + D] = gp;
                a.grows[sn_ + D] = gn;
                loss_acc += l;
            }
            ++q;
        } while (q < a.end && a.ukey[q] == key);
        if (on) slk_apply_vec<VEC, UPD>(a, 0, uoff, u, gu);
        if (lane == 0) slk_apply_bias<UPD>(a, 2, user, gbu);
    }
    const double tot = slk_block_sum_256((double)loss_acc, red);
    if (threadIdx.x == 0) a.loss_partial[blockIdx.x] = tot;
}

// this rank's share of the minibatch loss: (sum of its per-interaction losses) / global batch
__global__ __launch_bounds__(256) void k_shard_loss(const double *partial, int n, float inv_b, float *out) {
    __shared__ double red[256];
    double x = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) x += partial[i];
    const double tot = slk_block_sum_256(x, red);
    if (threadIdx.x == 0) *out = (float)(tot * (double)inv_b);
}

__global__ __launch_bounds__(256) void k_shard_item_keys(const int64_t *ids, uint32_t n, uint32_t *key,
                                                         uint32_t *val) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        key[j] = (uint32_t)ids[j];
        val[j] = j;
    }
}

static int check_plain(slk_ctx *ctx, const slk_tables *t) {
    if (t->user_bloom || t->item_bloom)
        return slk_fail(ctx, SLK_EINVAL, "BloomEmbedding tables are not supported by the row-sharded path");
    return SLK_OK;
}

static int check_shard(slk_ctx *ctx, const slk_shard *sh) {
    if (!sh) return slk_fail(ctx, SLK_EINVAL, "shard descriptor is NULL");
    if (sh->world < 1 || sh->world > SLK_MAX_WORLD || sh->rank < 0 || sh->rank >= sh->world)
        return slk_fail(ctx, SLK_EINVAL, "shard world %d / rank %d out of range (world <= %d)", sh->world,
                        sh->rank, SLK_MAX_WORLD);
    if (sh->num_items_global < 1 || sh->num_items_global > ((int64_t)1 << 32))
        return slk_fail(ctx, SLK_EINVAL, "num_items_global %lld outside [1, 2^32]", (long long)sh->num_items_global);
    if (sh->global_batch < 1) return slk_fail(ctx, SLK_EINVAL, "global_batch must be >= 1");
    return SLK_OK;
}

SLK_EXPORT int slk_shard_begin(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh,
                               const int64_t *d_users_local, const int64_t *d_items, int64_t n,
                               const int64_t *d_neg_in, int64_t *d_neg_out, int64_t *d_send_ids,
                               int64_t *d_send_counts, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if ((rc = check_shard(ctx, sh))) return rc;
    if (n < 0 || n >= ((int64_t)1 << 30)) return slk_fail(ctx, SLK_EINVAL, "slk_shard_begin: n %lld outside [0, 2^30)", (long long)n);
    if (!d_send_counts || (n > 0 && (!d_users_local || !d_items || !d_send_ids)))
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_begin: NULL pointer");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const uint32_t world = (uint32_t)sh->world;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_HIST], SLK_MAX_WORLD * 8))) return rc;
    unsigned long long *hist = (unsigned long long *)ctx->extra[SH_HIST].p;
    SLK_HIP(ctx, hipMemsetAsync(hist, 0, SLK_MAX_WORLD * 8, s));
    ctx->shard_n = 0;
    if (n > 0) {
        const uint32_t nn = (uint32_t)n, nl = 2 * nn;
        if ((rc = slk_ensure(ctx, ctx->neg32, (size_t)nn * 4))) return rc;
        for (int b = 0; b < 2; ++b) {
            if ((rc = slk_ensure(ctx, ctx->ukey[b], (size_t)nn * 4))) return rc;
            if ((rc = slk_ensure(ctx, ctx->uval[b], (size_t)nn * 8))) return rc;
            if ((rc = slk_ensure(ctx, ctx->extra[SH_OKEY0 + b], (size_t)nl * 4))) return rc;
            if ((rc = slk_ensure(ctx, ctx->extra[SH_OVAL0 + b], (size_t)nl * 4))) return rc;
        }
        if ((rc = slk_ensure(ctx, ctx->extra[SH_VSLOT], (size_t)nl * 4))) return rc;
        uint32_t *neg32 = (uint32_t *)ctx->neg32.p;
        // ---- negatives over the GLOBAL item range (sampling.py:34)
        if (d_neg_in) {
            slk_prof_begin(ctx, SLK_K_SAMPLE, s);
            if ((rc = slk_launch_i64_to_u32(ctx, d_neg_in, neg32, nn, s))) return rc;
            if (d_neg_out)
                SLK_HIP(ctx, hipMemcpyAsync(d_neg_out, d_neg_in, (size_t)nn * 8, hipMemcpyDeviceToDevice, s));
            slk_prof_end(ctx, s);
        } else {
            if ((rc = slk_sample_u32(ctx, sh->num_items_global, n, neg32, d_neg_out, s))) return rc;
        }
        // ---- sort by user; bucket the lookups by owner
        slk_prof_begin(ctx, SLK_K_PREP, s);
        const unsigned ubits = slk_bits_for((uint64_t)local->num_users - 1);
        hipLaunchKernelGGL(k_shard_user_keys, dim3(slk_grid_for(ctx, nn, 256)), dim3(256), 0, s, d_users_local,
                           d_items, (const uint32_t *)neg32, nn, (uint32_t *)ctx->ukey[0].p,
                           (uint64_t *)ctx->uval[0].p);
        SLK_LAUNCH_CHECK(ctx, "k_shard_user_keys");
        if ((rc = slk_sort_pairs_u32_u64(ctx, (const uint32_t *)ctx->ukey[0].p, (uint32_t *)ctx->ukey[1].p,
                                         (const uint64_t *)ctx->uval[0].p, (uint64_t *)ctx->uval[1].p, nn, ubits,
                                         s)))
            return rc;
        const uint32_t *uit = (const uint32_t *)ctx->uval[1].p;
        hipLaunchKernelGGL(k_shard_owner_keys, dim3(slk_grid_for(ctx, nl, 256)), dim3(256), 0, s, uit, nl, world,
                           (uint32_t *)ctx->extra[SH_OKEY0].p, (uint32_t *)ctx->extra[SH_OVAL0].p, hist);
        SLK_LAUNCH_CHECK(ctx, "k_shard_owner_keys");
        if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)ctx->extra[SH_OKEY0].p,
                                         (uint32_t *)ctx->extra[SH_OKEY1].p,
                                         (const uint32_t *)ctx->extra[SH_OVAL0].p,
                                         (uint32_t *)ctx->extra[SH_OVAL1].p, nl, slk_bits_for(world - 1), s)))
            return rc;
        hipLaunchKernelGGL(k_shard_slots, dim3(slk_grid_for(ctx, nl, 256)), dim3(256), 0, s, uit,
                           (const uint32_t *)ctx->extra[SH_OVAL1].p, nl, world, d_send_ids,
                           (uint32_t *)ctx->extra[SH_VSLOT].p);
        SLK_LAUNCH_CHECK(ctx, "k_shard_slots");
        slk_prof_end(ctx, s);
        ctx->shard_n = n;
    }
    hipLaunchKernelGGL(k_shard_counts, dim3(1), dim3(SLK_MAX_WORLD), 0, s, (const unsigned long long *)hist, world,
                       d_send_counts);
    SLK_LAUNCH_CHECK(ctx, "k_shard_counts");
    return SLK_OK;
}

SLK_EXPORT int slk_shard_gather(slk_ctx *ctx, const slk_tables *local, const int64_t *d_ids, int64_t n_ids,
                                float *d_rows_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if (n_ids < 0 || (n_ids > 0 && (!d_ids || !d_rows_out))) return slk_fail(ctx, SLK_EINVAL, "slk_shard_gather: bad arguments");
    if (n_ids == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    slk_prof_begin(ctx, SLK_K_EXCHANGE, s);
#define SLK_GATHER(V_, G_)                                                                                  \
    hipLaunchKernelGGL((k_shard_gather<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n_ids, 256 / G_)), dim3(256), 0, s, \
                       (const float *)local->d_param[1], (const float *)local->d_param[3], (int)local->dim,  \
                       shard_rsv(local->dim), d_ids, n_ids, d_rows_out)
    SLK_FOR_LAYOUT(vec, g, SLK_GATHER);
#undef SLK_GATHER
    SLK_LAUNCH_CHECK(ctx, "k_shard_gather");
    slk_prof_end(ctx, s);
    return SLK_OK;
}

template <int VEC, int G>
static slk_pass_fn shard_user_pass_fn(int upd) {
    if (upd == SLK_UPD_ADAGRAD) return k_shard_user_pass<VEC, G, SLK_UPD_ADAGRAD>;
    if (upd == SLK_UPD_SPARSE_ADAM) return k_shard_user_pass<VEC, G, SLK_UPD_SPARSE_ADAM>;
    return k_shard_user_pass<VEC, G, SLK_UPD_GRAD_ONLY>;
}

static void fill_tables(slk_pass_args &a, slk_ctx *ctx, const slk_tables *local, const slk_optim *optim,
                        bool dense) {
    for (int t = 0; t < 4; ++t) {
        a.P[t] = local->d_param[t];
        a.S1[t] = dense ? (float *)ctx->dgrad[t].p : optim->d_state1[t];
        a.S2[t] = optim->d_state2[t];
    }
    a.D = local->dim;
    a.NP = 2;
    a.pad_item = a.pad_item2 = 0xffffffffu;
    slk_set_opt_coeffs(a, optim);
    a.nt = ctx->opt_nt;
}

SLK_EXPORT int slk_shard_user_pass(slk_ctx *ctx, const slk_tables *local, const slk_optim *optim,
                                   const slk_shard *sh, int64_t n, int32_t loss, const float *d_rows_in,
                                   float *d_grad_out, float *d_loss_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if ((rc = slk_check_optim(ctx, optim, 15u))) return rc;
    if ((rc = check_shard(ctx, sh))) return rc;
    if (loss < SLK_LOSS_POINTWISE || loss > SLK_LOSS_HINGE)
        return slk_fail(ctx, SLK_EINVAL, "row-sharded path supports pointwise/bpr/hinge (loss kind %d)", loss);
    if (n != ctx->shard_n) return slk_fail(ctx, SLK_EINVAL, "slk_shard_user_pass: n %lld does not match slk_shard_begin (%lld)", (long long)n, (long long)ctx->shard_n);
    if (!d_loss_out || (n > 0 && (!d_rows_in || !d_grad_out))) return slk_fail(ctx, SLK_EINVAL, "slk_shard_user_pass: NULL pointer");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const bool dense = optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE;
    if (dense) {
        const size_t elems[4] = {(size_t)local->num_users * local->dim, (size_t)local->num_items * local->dim,
                                 (size_t)local->num_users, (size_t)local->num_items};
        if ((rc = slk_ensure_dgrad(ctx, elems, 15u, s))) return rc;
    }
    const unsigned max_grid = (unsigned)ctx->num_cus * (unsigned)(ctx->opt_user_grid_mult > 8 ? ctx->opt_user_grid_mult : 8);
    if ((rc = slk_ensure(ctx, ctx->losspart, (size_t)max_grid * 8))) return rc;
    slk_pass_args a;
    memset(&a, 0, sizeof(a));
    fill_tables(a, ctx, local, optim, dense);
    a.begin = 0;
    a.end = (uint32_t)n;
    a.ukey = (const uint32_t *)ctx->ukey[1].p;
    a.umask = 0xffffffffu;
    a.vslot = (const uint32_t *)ctx->extra[SH_VSLOT].p;
    a.vrows = d_rows_in;
    a.grows = d_grad_out;
    a.RSV = shard_rsv(local->dim);
    a.loss_partial = (double *)ctx->losspart.p;
    a.loss_kind = loss;
    a.inv_b = 1.0f / (float)sh->global_batch;
    const unsigned gpb = 256u / (unsigned)g;
    const unsigned ugrid = n > 0 ? slk_grid_for(ctx, (size_t)n, gpb) : 0;
    if (n > 0) {
        slk_pass_fn upass = nullptr;
        const int upd = slk_upd_for(optim->kind);
#define SLK_PICK(V_, G_) upass = shard_user_pass_fn<V_, G_>(upd)
        SLK_FOR_LAYOUT(vec, g, SLK_PICK);
#undef SLK_PICK
        slk_prof_begin(ctx, SLK_K_USER_PASS, s);
        hipLaunchKernelGGL(upass, dim3(ugrid), dim3(256), 0, s, a);
        SLK_LAUNCH_CHECK(ctx, "k_shard_user_pass");
        slk_prof_end(ctx, s);
    }
    hipLaunchKernelGGL(k_shard_loss, dim3(1), dim3(256), 0, s, (const double *)ctx->losspart.p, (int)ugrid, a.inv_b,
                       d_loss_out);
    SLK_LAUNCH_CHECK(ctx, "k_shard_loss");
    return SLK_OK;
}

SLK_EXPORT int slk_shard_item_pass(slk_ctx *ctx, const slk_tables *local, slk_optim *optim, const int64_t *d_ids,
                                   const float *d_grad_in, int64_t n_ids, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if ((rc = slk_check_optim(ctx, optim, 15u))) return rc;
    if (n_ids < 0 || n_ids >= ((int64_t)1 << 31) || (n_ids > 0 && (!d_ids || !d_grad_in)))
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_item_pass: bad arguments");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const bool dense = optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE;
    if (dense) {
        const size_t elems[4] = {(size_t)local->num_users * local->dim, (size_t)local->num_items * local->dim,
                                 (size_t)local->num_users, (size_t)local->num_items};
        if ((rc = slk_ensure_dgrad(ctx, elems, 15u, s))) return rc;
    }
    if (n_ids > 0) {
        const uint32_t nr = (uint32_t)n_ids;
        for (int b = 0; b < 2; ++b) {
            if ((rc = slk_ensure(ctx, ctx->ikey[b], (size_t)nr * 4))) return rc;
            if ((rc = slk_ensure(ctx, ctx->ipay[b], (size_t)nr * 4))) return rc;
        }
        const unsigned ibits = slk_bits_for((uint64_t)local->num_items - 1);
        slk_prof_begin(ctx, SLK_K_PREP, s);
        hipLaunchKernelGGL(k_shard_item_keys, dim3(slk_grid_for(ctx, nr, 256)), dim3(256), 0, s, d_ids, nr,
                           (uint32_t *)ctx->ikey[0].p, (uint32_t *)ctx->ipay[0].p);
        SLK_LAUNCH_CHECK(ctx, "k_shard_item_keys");
        if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)ctx->ikey[0].p, (uint32_t *)ctx->ikey[1].p,
                                         (const uint32_t *)ctx->ipay[0].p, (uint32_t *)ctx->ipay[1].p, nr, ibits, s)))
            return rc;
        slk_prof_end(ctx, s);
        slk_pass_args a;
        memset(&a, 0, sizeof(a));
        fill_tables(a, ctx, local, optim, dense);
        a.snap = const_cast<float *>(d_grad_in);
        a.RS = shard_rsv(local->dim);
        a.ibegin = 0;
        a.iend = nr;
        a.ikey = (const uint32_t *)ctx->ikey[1].p;
        a.imask = 0xffffffffu;
        a.ipay = (const uint32_t *)ctx->ipay[1].p;
        a.mb_loss_out = nullptr;
        slk_pass_fn ipass = nullptr;
        const int upd = slk_upd_for(optim->kind);
#define SLK_PICK(V_, G_) ipass = slk_item_pass_fn<V_, G_, SLK_ITEM_ROW>(upd)
        SLK_FOR_LAYOUT(vec, g, SLK_PICK);
#undef SLK_PICK
        const unsigned gpb = 256u / (unsigned)g;
        slk_prof_begin(ctx, SLK_K_ITEM_PASS, s);
        hipLaunchKernelGGL(ipass, dim3(slk_grid_for(ctx, nr, 4 * gpb, ctx->opt_item_grid_mult)), dim3(256), 0, s, a);
        SLK_LAUNCH_CHECK(ctx, "k_item_pass<ROW>");
        slk_prof_end(ctx, s);
    }
    if (dense && (rc = slk_dense_sweeps(ctx, local->d_param, optim, 15u, s))) return rc;
    optim->step += 1;
    return SLK_OK;
}
