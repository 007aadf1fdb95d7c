// slk_shard.hip -- row-sharded BilinearNet training for G = world GPUs.
//
// The reference has no multi-GPU code; the north star fixes the design: user and item tables
// (with their biases and optimizer state) are row-sharded cyclically (owner(row) = row mod G,
// local row = row div G), every rank processes the interactions of the global minibatch whose
// USER it owns (so user rows are always local), and item rows move over xGMI in RCCL
// all-to-all exchanges issued by the host (torch.distributed) between the calls below.
//
// Everything that depends only on IDS is done once per CHUNK of M minibatches, so that the
// minibatch loop needs no host synchronisation (the split sizes of every exchange of the chunk
// are known after one count exchange):
//
//   slk_shard_chunk_begin   negatives; sort the chunk's local interactions by (unit, user), a
//                           unit being one of the S user-slices of one minibatch; bucket the 2n
//                           item lookups by (owner, unit) -> send_ids (owner-local rows),
//                           send_counts[owner][unit]; every lookup gets a SLOT of its unit's
//                           exchange buffers (slk_blk_row in slk_kernels.h: blocks of 64 rows + their
//                           64 scalars; a peer's slots start on a block boundary)
//        [a2a: counts; ONE host sync; a2a: ids to the owners]
//   slk_shard_chunk_commit  both count matrices come back from the host; owner side: received
//                           ids regrouped by (unit, source) and sorted by (minibatch, row)
//   per unit t:
//     slk_shard_gather      owner: row records of unit t's requests   [a2a: rows -> requesters]
//     slk_shard_user_pass   USER PASS of slk_bilinear.hip over unit t reading item rows from the
//                           received records; one gradient record g * u_old (+ g) per lookup;
//                           local user rows updated in place            [a2a: grads -> owners]
//   per minibatch:
//     slk_shard_item_pass   owner: ITEM PASS (ROW mode) over the records of the minibatch's S
//                           units: one owner group per unique row sums them and applies the
//                           optimizer -- duplicate semantics stay exact (sum, then one update).
//
// Slices exist so that the exchanges of one unit overlap the compute of another (xGMI and HBM
// work in parallel); users of different slices are disjoint, so their in-place updates commute.
// Every forward still reads pre-step parameters and every row still gets the sum of its
// contributions before one optimizer update, exactly as on one GPU; only the summation order
// of an item row's contributions differs (by slice and source rank).
#include <math.h>

#include "slk_kernels.h"

#define SLK_MAX_WORLD 64
#define SLK_SHARD_MAX_BINS 2048  // world * units per chunk

// scratch slots in ctx->extra
enum { SH_OKEY0 = 0, SH_OKEY1, SH_OVAL0, SH_OVAL1, SH_VSLOT, SH_HIST, SH_SEGSTART, SH_UNITBASE, SH_MBOFF, SH_RID,
       SH_SEGTAB, SH_GSLOT, SH_UIT = 48, SH_GPOS = 49 };  // (48, 49: adaptive hinge -- the packed 1 + n item lists, global positions)

static inline int64_t pad_slots(int64_t lookups) { return (lookups + SLK_SHARD_BLOCK - 1) / SLK_SHARD_BLOCK * SLK_SHARD_BLOCK; }

SLK_EXPORT int64_t slk_shard_buffer_floats(int32_t dim, int64_t slots) { return pad_slots(slots) * (int64_t)(dim + 1); }

// key = (unit << ubits) | user with unit = minibatch * S + user % S; value = (neg << 32) | pos so
// the sorted values are the item pairs.  mb_off[0..M]: minibatch boundaries in the chunk.
__global__ __launch_bounds__(256) void k_shard_user_keys(const int64_t *users, const int64_t *items,
                                                         const uint32_t *neg32, uint32_t n,
                                                         const uint32_t *mb_off, uint32_t M, uint32_t S,
                                                         unsigned ubits, uint32_t *key, uint64_t *val) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        uint32_t lo = 0, hi = M;  // largest mb with mb_off[mb] <= k
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (mb_off[mid] <= k) lo = mid;
            else hi = mid;
        }
        const uint32_t u = (uint32_t)users[k];
        key[k] = ((lo * S + u % S) << ubits) | u;
        val[k] = ((uint64_t)neg32[k] << 32) | (uint64_t)(uint32_t)items[k];
    }
}

// adaptive hinge (1 + nn lookups per interaction, no room for them in a 64-bit payload): the same key, payload = the
// interaction's index in the chunk; k_shard_pack then lays the item lists out in sorted order
__global__ __launch_bounds__(256) void k_shard_user_keys_idx(const int64_t *users, uint32_t n, const uint32_t *mb_off,
                                                             uint32_t M, uint32_t S, unsigned ubits, uint32_t *key,
                                                             uint32_t *val) {
    for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        uint32_t lo = 0, hi = M;  // largest mb with mb_off[mb] <= k
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (mb_off[mid] <= k) lo = mid;
            else hi = mid;
        }
        const uint32_t u = (uint32_t)users[k];
        key[k] = ((lo * S + u % S) << ubits) | u;
        val[k] = k;
    }
}

// uit[q * NP + s]: the positive (s = 0) and the nn draws of the interaction at sorted position q (its draws are the flat
// entries [k * nn, (k + 1) * nn) of the minibatch's ONE randint call: users repeated user-major, implicit.py:266-275);
// gpos[q]: its position inside its GLOBAL minibatch (the row of the minibatch's score matrix it fills)
__global__ __launch_bounds__(256) void k_shard_pack(const uint32_t *uk, const int64_t *items, const uint32_t *neg32,
                                                    const int64_t *mb_pos, uint32_t n, int nn, uint32_t *uit, uint32_t *gpos) {
    const int NP = nn + 1;
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) {
        const uint32_t k = uk[q];
        uit[(size_t)q * NP] = (uint32_t)items[k];
        for (int r = 0; r < nn; ++r) uit[(size_t)q * NP + 1 + r] = neg32[(size_t)k * nn + r];
        gpos[q] = (uint32_t)mb_pos[k];
    }
}

// lookup l = NP*q + s (sorted position q, pair s): key = owner * T + unit, value = l
__global__ __launch_bounds__(256) void k_shard_owner_keys(const uint32_t *uit, const uint32_t *ukey, unsigned ubits,
                                                          uint32_t nl, uint32_t NP, uint32_t world, uint32_t T, uint32_t *okey,
                                                          uint32_t *oval, unsigned long long *hist) {
    __shared__ unsigned h[SLK_SHARD_MAX_BINS];
    const uint32_t bins = world * T;
    for (uint32_t i = threadIdx.x; i < bins; i += 256) h[i] = 0;
    __syncthreads();
    for (uint32_t l = blockIdx.x * 256 + threadIdx.x; l < nl; l += gridDim.x * 256) {
        const uint32_t b = (uit[l] % world) * T + (ukey[l / NP] >> ubits);
        okey[l] = b;
        oval[l] = l;
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// one block: seg_start = exclusive scan of the (owner, unit) counts in sorted order;
// unit_base[o][t] = first slot of owner o inside unit t's buffers: the slots of the owners before it, each owner's
// count rounded up to whole blocks
__global__ __launch_bounds__(256) void k_shard_scan(const unsigned long long *hist, uint32_t world, uint32_t T,
                                                    uint32_t *seg_start, uint32_t *unit_base, int64_t *counts_out) {
    const uint32_t bins = world * T;
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t b = 0; b < bins; ++b) {
            seg_start[b] = run;
            run += (uint32_t)hist[b];
        }
    }
    for (uint32_t t = threadIdx.x; t < T; t += 256) {
        uint32_t run = 0;
        for (uint32_t o = 0; o < world; ++o) {
            unit_base[o * T + t] = run;
            run += ((uint32_t)hist[o * T + t] + (SLK_SHARD_BLOCK - 1u)) & ~(SLK_SHARD_BLOCK - 1u);
        }
    }
    for (uint32_t b = threadIdx.x; b < bins; b += 256) counts_out[b] = (int64_t)hist[b];
}

__global__ __launch_bounds__(256) void k_shard_slots(const uint32_t *uit, const uint32_t *okey_sorted,
                                                     const uint32_t *oval_sorted, uint32_t nl, uint32_t world,
                                                     const uint32_t *seg_start, const uint32_t *unit_base,
                                                     int32_t *send_ids, uint32_t *vslot) {
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < nl; j += gridDim.x * 256) {
        const uint32_t l = oval_sorted[j], b = okey_sorted[j];
        send_ids[j] = (int32_t)(uit[l] / world);
        vslot[l] = unit_base[b] + (j - seg_start[b]);
    }
}

// owner side: blockIdx.x = received segment (source, unit), blockIdx.y strides inside it (a single-rank run has two
// segments of a million ids each: one block per segment took 3.5 ms, profiles/r02_a_kernel_stats.md): ids move from [source][unit] order
// to [unit][source] order, with the item-pass key (minibatch, row) and payload (slot inside the
// minibatch's record buffer).  segtab[seg] = {from, to, len, minibatch, first slot of the segment in its unit's
// buffers, first slot of the segment in its minibatch's gradient buffer}
__global__ __launch_bounds__(256) void k_shard_regroup(const int32_t *recv_ids, const uint32_t *segtab, unsigned ibits,
                                                       int32_t *rid, uint32_t *gslot, uint32_t *ikey, uint32_t *ipay) {
    const uint32_t *sg = segtab + 6 * (size_t)blockIdx.x;
    const uint32_t from = sg[0], to = sg[1], len = sg[2], mb = sg[3], unit_slot = sg[4], mb_slot = sg[5];
    for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < len; i += gridDim.y * 256) {
        const int32_t id = recv_ids[from + i];
        rid[to + i] = id;
        gslot[to + i] = unit_slot + i;
        ikey[to + i] = (mb << ibits) | (uint32_t)id;
        ipay[to + i] = mb_slot + i;
    }
}

// owner side: slot(j) of the unit's row buffer = V[id_j] (D), bias[id_j]
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_shard_gather(const float *V, const float *bi, uint32_t bsh, int D, const int32_t *ids,
                                                      const uint32_t *gslot, int64_t n, float *out) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    for (int64_t j = (int64_t)blockIdx.x * GPB + grp; j < n; j += (int64_t)gridDim.x * GPB) {
        const int64_t i = (int64_t)ids[j];
        const uint32_t slot = gslot[j];
        if (on) slk_vstore<VEC>(out + slk_blk_row(slot, D) + d0, slk_vload<VEC>(V + (size_t)i * D + d0));
        if (lane == 0) out[slk_blk_scalar(slot, D)] = bi[(size_t)i << bsh];  // (bsh = 1: the {bias, accumulator} shadow of a training scope)
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
        slk_vec<VEC> u = on ? slk_vload_if_nt<VEC>(a.P[0] + uoff, (SLK_NT_OF(a) & 1) != 0) : slk_vzero<VEC>();
        const float bu = a.ubz ? 0.0f : a.P[2][user];  // (SLK_TABLES_USER_BIAS_ZERO, as in slk_bilinear.hip's user pass)
        slk_vec<VEC> gu = slk_vzero<VEC>();
        float gbu = 0.0f;
        uint32_t q = p;
        do {
            const uint32_t slp = a.vslot[2 * (size_t)q], sln = a.vslot[2 * (size_t)q + 1];
            const size_t sp_ = slk_blk_row(slp, D), sn_ = slk_blk_row(sln, D);
            const size_t bp_ = slk_blk_scalar(slp, D), bn_ = slk_blk_scalar(sln, D);
            // exchange buffers are read / written exactly once: streaming hints (ctx option "nt" bit 0)
            const bool nt = (SLK_NT_OF(a) & 1) != 0;
            const slk_vec<VEC> vi = on ? slk_vload_if_nt<VEC>(a.vrows + sp_ + d0, nt) : slk_vzero<VEC>();
            const slk_vec<VEC> vj = on ? slk_vload_if_nt<VEC>(a.vrows + sn_ + d0, nt) : slk_vzero<VEC>();
            const float sp = slk_group_sum<G>(slk_vdot<VEC>(u, vi)) + bu + a.vrows[bp_];
            const float sn = slk_group_sum<G>(slk_vdot<VEC>(u, vj)) + bu + a.vrows[bn_];
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
                slk_vstore_if_nt<VEC>(a.grows + sp_ + d0, cp, nt);
                slk_vstore_if_nt<VEC>(a.grows + sn_ + d0, cn, nt);
            }
            if (lane == 0) {
                a.grows[bp_] = gp;
                a.grows[bn_] = gn;
                loss_acc += l;
            }
            ++q;
        } while (q < a.end && a.ukey[q] == key);
        if (on) slk_apply_vec<VEC, UPD>(a, 0, uoff, u, gu, (SLK_NT_OF(a) & 1) != 0);
        if (lane == 0) slk_apply_bias<UPD>(a, 2, user, gbu);
    }
    const double tot = slk_block_sum_256((double)loss_acc, red);
    if (threadIdx.x == 0) a.loss_partial[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------------------------------------
// adaptive hinge on the row-sharded path (implicit.py:266-275 + losses.py:127-166).  Column c of the minibatch's [n, B]
// candidate matrix holds the flat draws {r B + c}, scored with the users of OTHER interactions (flat entry k uses user
// k / n) -- interactions that live on other ranks.  So the step has a score phase in front: every rank scores the 1 + n
// pairs of ITS interactions (all 1 + n rows travel), the B x (1 + n) score matrix is summed over the ranks (each entry has
// one non-zero contributor: exact), every rank runs the same selection over the whole matrix (k_adaptive_select,
// slk_kernels.h: the one-GPU path's kernel), and the user pass reads dL/dscore of its pairs from the result.
// ---------------------------------------------------------------------------------------------------------------------
// one row group per sorted position of the unit: sk[gpos * NP + s] = u . v_s + bu + bi_s, the dot as k_score_pass forms it
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_shard_score_pass(slk_pass_args a) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const uint32_t stride = gridDim.x * GPB;
    for (uint32_t q = a.begin + blockIdx.x * GPB + grp; q < a.end; q += stride) {
        const uint32_t user = a.ukey[q] & a.umask;
        const slk_vec<VEC> u = on ? slk_vload<VEC>(a.P[0] + (size_t)user * D + d0) : slk_vzero<VEC>();
        const float bu = a.P[2][user];
        const size_t kb = (size_t)a.uk[q] * a.NP, qb = (size_t)q * a.NP;
        for (int s = 0; s < a.NP; ++s) {
            const uint32_t slot = a.vslot[qb + s];
            const slk_vec<VEC> v = on ? slk_vload<VEC>(a.vrows + slk_blk_row(slot, D) + d0) : slk_vzero<VEC>();
            const float bi = a.vrows[slk_blk_scalar(slot, D)];
            const float sc = slk_group_sum<G>(slk_vdot<VEC>(u, v)) + bu + bi;
            if (lane == 0) a.sk[kb + s] = sc;
        }
    }
}

// USER PASS with dL/dscore given (k_user_pass<..., UMODE 1> over the exchange buffers): the user gradient is the sum over the
// user's positions, and per position over its LIVE pairs in pair order, of g * v; every lookup's slot of the gradient buffer
// gets g * u_old and g -- zeros for the pairs the selection left out (the owner adds them: SparseAdam's "touched" rows
// decay, Adagrad's zero sum is an exact no-op, as on one GPU)
template <int VEC, int G, int UPD>
__global__ __launch_bounds__(256) void k_shard_user_pass_pre(slk_pass_args a) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const uint32_t stride = gridDim.x * GPB;
    for (uint32_t p = a.begin + blockIdx.x * GPB + grp; p < a.end; p += stride) {
        const uint32_t key = a.ukey[p];
        if (p > a.begin && a.ukey[p - 1] == key) continue;  // not the head of its user segment
        const uint32_t user = key & a.umask;
        const size_t uoff = (size_t)user * D + d0;
        slk_vec<VEC> u = on ? slk_vload<VEC>(a.P[0] + uoff) : slk_vzero<VEC>();
        slk_vec<VEC> gu = slk_vzero<VEC>();
        float gbu = 0.0f;
        uint32_t q = p;
        do {
            const size_t kb = (size_t)a.uk[q] * a.NP, qb = (size_t)q * a.NP;
            for (int s = 0; s < a.NP; ++s) {
                const float g = a.gk[kb + s];
                const uint32_t slot = a.vslot[qb + s];
                const size_t ro = slk_blk_row(slot, D);
                if (g != 0.0f) {
                    const slk_vec<VEC> v = on ? slk_vload<VEC>(a.vrows + ro + d0) : slk_vzero<VEC>();
                    slk_vaxpy<VEC>(gu, g, v);
                    gbu += g;
                }
                slk_vec<VEC> c;
#pragma unroll
                for (int i = 0; i < VEC; ++i) c.v[i] = g * u.v[i];
                if (on) slk_vstore<VEC>(a.grows + ro + d0, c);
                if (lane == 0) a.grows[slk_blk_scalar(slot, D)] = g;
            }
            ++q;
        } while (q < a.end && a.ukey[q] == key);
        if (on) slk_apply_vec<VEC, UPD>(a, 0, uoff, u, gu, false);
        if (lane == 0) slk_apply_bias<UPD>(a, 2, user, gbu);
    }
}

// this rank's share of the minibatch loss: (sum of its per-interaction losses) / global batch
__global__ __launch_bounds__(256) void k_shard_loss(const double *partial, int n, float inv_b, float *out,
                                                     int accumulate) {
    __shared__ double red[256];
    double x = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) x += partial[i];
    const double tot = slk_block_sum_256(x, red);
    if (threadIdx.x == 0) *out = (accumulate ? *out : 0.0f) + (float)(tot * (double)inv_b);
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
    return SLK_OK;
}


static int check_chunk(slk_ctx *ctx, const slk_shard *sh, int32_t M, int32_t S) {
    if (M < 1 || S < 1 || (int64_t)M * S * sh->world > SLK_SHARD_MAX_BINS)
        return slk_fail(ctx, SLK_EINVAL, "shard chunk: minibatches %d x slices %d x world %d must be in [1, %d]", M, S,
                        sh->world, SLK_SHARD_MAX_BINS);
    return SLK_OK;
}

// Scratch of chunks of up to n local interactions whose owner side receives up to n_recv lookups, allocated now: a
// timed loop that starts with a larger chunk than its warm-up would otherwise pay hipFree + hipMalloc (device-wide
// synchronisations) inside its first chunk.
SLK_EXPORT int slk_shard_reserve(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh, int64_t n, int64_t n_recv) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g, /*shadow_ok=*/true))) return rc;
    if ((rc = check_shard(ctx, sh))) return rc;
    if (n < 0 || n >= ((int64_t)1 << 30) || n_recv < 0 || n_recv >= ((int64_t)1 << 31))
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_reserve: n %lld / n_recv %lld out of range", (long long)n, (long long)n_recv);
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)n, nl = 2 * nn, nr = (size_t)n_recv;  // (an adaptive-hinge chunk grows its scratch on demand)
    if ((rc = slk_ensure(ctx, ctx->extra[SH_HIST], (size_t)SLK_SHARD_MAX_BINS * 8))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_SEGSTART], (size_t)SLK_SHARD_MAX_BINS * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_UNITBASE], (size_t)SLK_SHARD_MAX_BINS * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_MBOFF], (size_t)(SLK_SHARD_MAX_BINS + 1) * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_SEGTAB], (size_t)SLK_SHARD_MAX_BINS * 6 * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->neg32, nn * 4))) return rc;
    for (int b = 0; b < 2; ++b) {
        if ((rc = slk_ensure(ctx, ctx->ukey[b], nn * 4))) return rc;
        if ((rc = slk_ensure(ctx, ctx->uval[b], nn * 8))) return rc;
        if ((rc = slk_ensure(ctx, ctx->extra[SH_OKEY0 + b], nl * 4))) return rc;
        if ((rc = slk_ensure(ctx, ctx->extra[SH_OVAL0 + b], nl * 4))) return rc;
        if ((rc = slk_ensure(ctx, ctx->ikey[b], nr * 4))) return rc;
        if ((rc = slk_ensure(ctx, ctx->ipay[b], nr * 4))) return rc;
    }
    if ((rc = slk_ensure(ctx, ctx->extra[SH_VSLOT], nl * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_RID], nr * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_GSLOT], nr * 4))) return rc;
    const unsigned max_grid = (unsigned)ctx->num_cus * (unsigned)(ctx->opt_user_grid_mult > 8 ? ctx->opt_user_grid_mult : 8);
    if ((rc = slk_ensure(ctx, ctx->losspart, (size_t)max_grid * 8))) return rc;
    if ((rc = slk_sample_reserve(ctx, sh->num_items_global, n))) return rc;
    return slk_sort_reserve(ctx, nl > nr ? nl : nr);
}

// n_neg == 0: one negative per interaction, the (positive, negative) pair rides as the sort's 64-bit payload;
// n_neg >= 1 (adaptive hinge): 1 + n_neg lookups per interaction, d_mb_pos gives every interaction's row of its
// minibatch's score matrix
static int shard_chunk_begin_impl(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh,
                                  const int64_t *d_users_local, const int64_t *d_items, int64_t n,
                                  const int64_t *h_mb_off, int32_t M, int32_t S, int32_t n_neg, const int64_t *d_neg_in,
                                  int64_t *d_neg_out, const int64_t *d_mb_pos, int32_t *d_send_ids, int64_t *d_send_counts,
                                  void *stream);

SLK_EXPORT int slk_shard_chunk_begin(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh,
                                     const int64_t *d_users_local, const int64_t *d_items, int64_t n,
                                     const int64_t *h_mb_off, int32_t M, int32_t S, const int64_t *d_neg_in,
                                     int64_t *d_neg_out, int32_t *d_send_ids, int64_t *d_send_counts,
                                     void *stream) {
    return shard_chunk_begin_impl(ctx, local, sh, d_users_local, d_items, n, h_mb_off, M, S, 0, d_neg_in, d_neg_out, nullptr,
                                  d_send_ids, d_send_counts, stream);
}

SLK_EXPORT int slk_shard_chunk_begin_adaptive(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh,
                                              const int64_t *d_users_local, const int64_t *d_items, int64_t n,
                                              const int64_t *h_mb_off, int32_t M, int32_t S, int32_t n_neg,
                                              const int64_t *d_neg_in, int64_t *d_neg_out, const int64_t *d_mb_pos,
                                              int32_t *d_send_ids, int64_t *d_send_counts, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (n_neg < 1 || n_neg > 1024) return slk_fail(ctx, SLK_EINVAL, "num_negative_samples %d outside [1, 1024]", n_neg);
    if (n > 0 && !d_mb_pos) return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_begin_adaptive: d_mb_pos is NULL");
    return shard_chunk_begin_impl(ctx, local, sh, d_users_local, d_items, n, h_mb_off, M, S, n_neg, d_neg_in, d_neg_out, d_mb_pos,
                                  d_send_ids, d_send_counts, stream);
}

static int shard_chunk_begin_impl(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh,
                                  const int64_t *d_users_local, const int64_t *d_items, int64_t n,
                                  const int64_t *h_mb_off, int32_t M, int32_t S, int32_t n_neg, const int64_t *d_neg_in,
                                  int64_t *d_neg_out, const int64_t *d_mb_pos, int32_t *d_send_ids, int64_t *d_send_counts,
                                  void *stream) {
    if (!ctx) return SLK_EINVAL;
    const bool multi = n_neg > 0;
    const int nng = multi ? n_neg : 1;  // negatives per interaction
    const uint32_t NP = (uint32_t)nng + 1u;
    if (n * (int64_t)NP >= ((int64_t)1 << 31)) return slk_fail(ctx, SLK_EINVAL, "shard chunk: %lld lookups >= 2^31", (long long)(n * NP));
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g, /*shadow_ok=*/true))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if ((rc = check_shard(ctx, sh))) return rc;
    if ((rc = check_chunk(ctx, sh, M, S))) return rc;
    if (n < 0 || n >= ((int64_t)1 << 30)) return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_begin: n %lld outside [0, 2^30)", (long long)n);
    if (!d_send_counts || !h_mb_off || (n > 0 && (!d_users_local || !d_items || !d_send_ids)))
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_begin: NULL pointer");
    if (h_mb_off[0] != 0 || h_mb_off[M] != n) return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_begin: minibatch offsets must span [0, n]");
    for (int m = 0; m < M; ++m)
        if (h_mb_off[m + 1] < h_mb_off[m]) return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_begin: minibatch offsets must ascend");
    const unsigned ubits = slk_bits_for((uint64_t)local->num_users - 1);
    const uint32_t T = (uint32_t)M * (uint32_t)S, world = (uint32_t)sh->world, bins = world * T;
    if (ubits + slk_bits_for((uint64_t)T - 1) > 32)
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_begin: %u units do not fit beside %u user-id bits in a 32-bit key (use fewer minibatches per chunk)", T, ubits);
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_HIST], (size_t)SLK_SHARD_MAX_BINS * 8))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_SEGSTART], (size_t)SLK_SHARD_MAX_BINS * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_UNITBASE], (size_t)SLK_SHARD_MAX_BINS * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SH_MBOFF], (size_t)(SLK_SHARD_MAX_BINS + 1) * 4))) return rc;
    unsigned long long *hist = (unsigned long long *)ctx->extra[SH_HIST].p;
    SLK_HIP(ctx, hipMemsetAsync(hist, 0, (size_t)bins * 8, s));
    ctx->shard_n = -1;  // no committed chunk
    ctx->sh_M = M;
    ctx->sh_S = S;
    ctx->sh_world = (int)world;
    ctx->sh_ubits = ubits;
    ctx->sh_n = n;
    ctx->sh_NP = (int)NP;
    ctx->sh_adaptive = multi;  // (an adaptive chunk with n_neg = 1 also has NP == 2: the flag, not NP, tells the layouts apart)
    if (n > 0) {
        const uint32_t nn = (uint32_t)n, nl = NP * nn;
        if ((rc = slk_ensure(ctx, ctx->neg32, (size_t)nn * nng * 4))) return rc;
        if (multi) {
            if ((rc = slk_ensure(ctx, ctx->extra[SH_UIT], (size_t)nl * 4))) return rc;
            if ((rc = slk_ensure(ctx, ctx->extra[SH_GPOS], (size_t)nn * 4))) return rc;
        }
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
            if ((rc = slk_launch_i64_to_u32(ctx, d_neg_in, neg32, (size_t)nn * nng, s))) return rc;
            if (d_neg_out)
                SLK_HIP(ctx, hipMemcpyAsync(d_neg_out, d_neg_in, (size_t)nn * nng * 8, hipMemcpyDeviceToDevice, s));
            slk_prof_end(ctx, s);
        } else {
            if ((rc = slk_sample_u32(ctx, sh->num_items_global, n * (int64_t)nng, neg32, d_neg_out, s))) return rc;
        }
        // ---- sort by (unit, user); bucket the lookups by (owner, unit)
        slk_prof_begin(ctx, SLK_K_PREP, s);
        ctx->sh_host.resize((size_t)(M + 1) / 2 + 1);
        uint32_t *h32 = (uint32_t *)ctx->sh_host.data();
        for (int m = 0; m <= M; ++m) h32[m] = (uint32_t)h_mb_off[m];
        SLK_HIP(ctx, hipMemcpyAsync(ctx->extra[SH_MBOFF].p, h32, (size_t)(M + 1) * 4, hipMemcpyHostToDevice, s));
        const uint32_t *uit;
        if (!multi) {
            hipLaunchKernelGGL(k_shard_user_keys, dim3(slk_grid_for(ctx, nn, 256)), dim3(256), 0, s, d_users_local,
                               d_items, (const uint32_t *)neg32, nn, (const uint32_t *)ctx->extra[SH_MBOFF].p,
                               (uint32_t)M, (uint32_t)S, ubits, (uint32_t *)ctx->ukey[0].p, (uint64_t *)ctx->uval[0].p);
            SLK_LAUNCH_CHECK(ctx, "k_shard_user_keys");
            if ((rc = slk_sort_pairs_u32_u64(ctx, (const uint32_t *)ctx->ukey[0].p, (uint32_t *)ctx->ukey[1].p,
                                             (const uint64_t *)ctx->uval[0].p, (uint64_t *)ctx->uval[1].p, nn,
                                             ubits + slk_bits_for((uint64_t)T - 1), s, true)))
                return rc;
            uit = (const uint32_t *)ctx->uval[1].p;
        } else {
            // the payload is the interaction's index; the 1 + n item lists and the global positions follow it into sorted order
            hipLaunchKernelGGL(k_shard_user_keys_idx, dim3(slk_grid_for(ctx, nn, 256)), dim3(256), 0, s, d_users_local, nn,
                               (const uint32_t *)ctx->extra[SH_MBOFF].p, (uint32_t)M, (uint32_t)S, ubits,
                               (uint32_t *)ctx->ukey[0].p, (uint32_t *)ctx->uval[0].p);
            SLK_LAUNCH_CHECK(ctx, "k_shard_user_keys_idx");
            if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)ctx->ukey[0].p, (uint32_t *)ctx->ukey[1].p,
                                             (const uint32_t *)ctx->uval[0].p, (uint32_t *)ctx->uval[1].p, nn,
                                             ubits + slk_bits_for((uint64_t)T - 1), s, true)))
                return rc;
            hipLaunchKernelGGL(k_shard_pack, dim3(slk_grid_for(ctx, nn, 256)), dim3(256), 0, s, (const uint32_t *)ctx->uval[1].p,
                               d_items, (const uint32_t *)neg32, d_mb_pos, nn, nng, (uint32_t *)ctx->extra[SH_UIT].p,
                               (uint32_t *)ctx->extra[SH_GPOS].p);
            SLK_LAUNCH_CHECK(ctx, "k_shard_pack");
            uit = (const uint32_t *)ctx->extra[SH_UIT].p;
        }
        hipLaunchKernelGGL(k_shard_owner_keys, dim3(slk_grid_for(ctx, nl, 256)), dim3(256), 0, s, uit,
                           (const uint32_t *)ctx->ukey[1].p, ubits, nl, NP, world, T, (uint32_t *)ctx->extra[SH_OKEY0].p,
                           (uint32_t *)ctx->extra[SH_OVAL0].p, hist);
        SLK_LAUNCH_CHECK(ctx, "k_shard_owner_keys");
        if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)ctx->extra[SH_OKEY0].p,
                                         (uint32_t *)ctx->extra[SH_OKEY1].p,
                                         (const uint32_t *)ctx->extra[SH_OVAL0].p,
                                         (uint32_t *)ctx->extra[SH_OVAL1].p, nl, slk_bits_for((uint64_t)bins - 1), s, true)))
            return rc;
        slk_prof_end(ctx, s);
    }
    slk_prof_begin(ctx, SLK_K_PREP, s);
    hipLaunchKernelGGL(k_shard_scan, dim3(1), dim3(256), 0, s, (const unsigned long long *)hist, world, T,
                       (uint32_t *)ctx->extra[SH_SEGSTART].p, (uint32_t *)ctx->extra[SH_UNITBASE].p, d_send_counts);
    SLK_LAUNCH_CHECK(ctx, "k_shard_scan");
    if (n > 0) {
        const uint32_t nl = NP * (uint32_t)n;
        hipLaunchKernelGGL(k_shard_slots, dim3(slk_grid_for(ctx, nl, 256)), dim3(256), 0, s,
                           multi ? (const uint32_t *)ctx->extra[SH_UIT].p : (const uint32_t *)ctx->uval[1].p,
                           (const uint32_t *)ctx->extra[SH_OKEY1].p,
                           (const uint32_t *)ctx->extra[SH_OVAL1].p, nl, world,
                           (const uint32_t *)ctx->extra[SH_SEGSTART].p, (const uint32_t *)ctx->extra[SH_UNITBASE].p,
                           d_send_ids, (uint32_t *)ctx->extra[SH_VSLOT].p);
        SLK_LAUNCH_CHECK(ctx, "k_shard_slots");
    }
    slk_prof_end(ctx, s);
    return SLK_OK;
}

SLK_EXPORT int slk_shard_chunk_commit(slk_ctx *ctx, const slk_tables *local, const slk_shard *sh,
                                      const int64_t *h_send_counts, const int64_t *h_recv_counts,
                                      const int32_t *d_recv_ids, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g, /*shadow_ok=*/true))) return rc;
    if ((rc = check_shard(ctx, sh))) return rc;
    if (ctx->sh_M < 1 || ctx->sh_world != sh->world) return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_commit: no chunk begun for this world size");
    if (!h_send_counts || !h_recv_counts) return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_commit: NULL count matrix");
    const int M = ctx->sh_M, S = ctx->sh_S, W = ctx->sh_world, T = M * S;
    // ---- unit windows: user-sorted positions (requester) and received-lookup positions (owner); slots of the unit's
    //      exchange buffers on either side (per-peer counts rounded up to whole blocks)
    ctx->sh_ustart.assign((size_t)T + 1, 0);
    ctx->sh_rstart.assign((size_t)T + 1, 0);
    ctx->sh_sslots.assign((size_t)T, 0);
    ctx->sh_rslots.assign((size_t)T + 1, 0);  // exclusive prefix over the units: first slot of unit t's received region
    for (int t = 0; t < T; ++t) {
        int64_t sent = 0, recv = 0, sslots = 0, rslots = 0;
        for (int r = 0; r < W; ++r) {
            if (h_send_counts[(size_t)r * T + t] < 0 || h_recv_counts[(size_t)r * T + t] < 0)
                return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_commit: negative count");
            sent += h_send_counts[(size_t)r * T + t];
            recv += h_recv_counts[(size_t)r * T + t];
            sslots += pad_slots(h_send_counts[(size_t)r * T + t]);
            rslots += pad_slots(h_recv_counts[(size_t)r * T + t]);
        }
        if (sent % ctx->sh_NP)
            return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_commit: unit %d sends %lld lookups, not a multiple of %d per interaction", t,
                            (long long)sent, ctx->sh_NP);
        ctx->sh_ustart[t + 1] = ctx->sh_ustart[t] + sent / ctx->sh_NP;
        ctx->sh_rstart[t + 1] = ctx->sh_rstart[t] + recv;
        ctx->sh_sslots[t] = sslots;
        ctx->sh_rslots[t + 1] = ctx->sh_rslots[t] + rslots;
    }
    if (ctx->sh_ustart[T] != ctx->sh_n)
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_commit: send counts cover %lld interactions, chunk has %lld",
                        (long long)ctx->sh_ustart[T], (long long)ctx->sh_n);
    if (ctx->sh_rslots[T] >= ((int64_t)1 << 32))
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_commit: %lld received slots >= 2^32", (long long)ctx->sh_rslots[T]);
    const int64_t nr = ctx->sh_rstart[T];
    if (nr >= ((int64_t)1 << 31)) return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_commit: %lld received lookups >= 2^31", (long long)nr);
    if (nr > 0 && !d_recv_ids) return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_commit: d_recv_ids is NULL");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    if (nr > 0) {
        const unsigned ibits = slk_bits_for((uint64_t)local->num_items - 1);
        if (ibits + slk_bits_for((uint64_t)M - 1) > 32)
            return slk_fail(ctx, SLK_EINVAL, "slk_shard_chunk_commit: %d minibatches do not fit beside %u item-row bits in a 32-bit key", M, ibits);
        for (int b = 0; b < 2; ++b) {
            if ((rc = slk_ensure(ctx, ctx->ikey[b], (size_t)nr * 4))) return rc;
            if ((rc = slk_ensure(ctx, ctx->ipay[b], (size_t)nr * 4))) return rc;
        }
        if ((rc = slk_ensure(ctx, ctx->extra[SH_RID], (size_t)nr * 4))) return rc;
        if ((rc = slk_ensure(ctx, ctx->extra[SH_GSLOT], (size_t)nr * 4))) return rc;
        const int nseg = W * T;
        if ((rc = slk_ensure(ctx, ctx->extra[SH_SEGTAB], (size_t)nseg * 6 * 4))) return rc;
        // received buffer order: [source][unit]; regrouped order: [unit][source]
        ctx->sh_host2.resize(((size_t)nseg * 6 + 1) / 2 + 1);
        uint32_t *tab = (uint32_t *)ctx->sh_host2.data();
        int64_t from = 0;
        for (int r = 0; r < W; ++r)
            for (int t = 0; t < T; ++t) {
                int64_t to = ctx->sh_rstart[t], unit_slot = 0;
                for (int r2 = 0; r2 < r; ++r2) {
                    to += h_recv_counts[(size_t)r2 * T + t];
                    unit_slot += pad_slots(h_recv_counts[(size_t)r2 * T + t]);
                }
                uint32_t *sg = tab + 6 * ((size_t)r * T + t);
                sg[0] = (uint32_t)from;
                sg[1] = (uint32_t)to;
                sg[2] = (uint32_t)h_recv_counts[(size_t)r * T + t];
                sg[3] = (uint32_t)(t / S);
                sg[4] = (uint32_t)unit_slot;
                sg[5] = (uint32_t)(ctx->sh_rslots[t] - ctx->sh_rslots[(size_t)(t / S) * S] + unit_slot);
                from += h_recv_counts[(size_t)r * T + t];
            }
        slk_prof_begin(ctx, SLK_K_PREP, s);
        SLK_HIP(ctx, hipMemcpyAsync(ctx->extra[SH_SEGTAB].p, tab, (size_t)nseg * 6 * 4, hipMemcpyHostToDevice, s));
        // enough blocks per segment to fill the chip whatever the segment count is
        unsigned per_seg = (unsigned)((8 * ctx->num_cus + nseg - 1) / nseg);
        const unsigned seg_blocks = (unsigned)((nr / nseg + 255) / 256) + 1u;
        if (per_seg > seg_blocks) per_seg = seg_blocks;
        if (per_seg < 1u) per_seg = 1u;
        if (per_seg > 65535u) per_seg = 65535u;
        hipLaunchKernelGGL(k_shard_regroup, dim3((unsigned)nseg, per_seg), dim3(256), 0, s, d_recv_ids,
                           (const uint32_t *)ctx->extra[SH_SEGTAB].p, ibits, (int32_t *)ctx->extra[SH_RID].p,
                           (uint32_t *)ctx->extra[SH_GSLOT].p, (uint32_t *)ctx->ikey[0].p, (uint32_t *)ctx->ipay[0].p);
        SLK_LAUNCH_CHECK(ctx, "k_shard_regroup");
        if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)ctx->ikey[0].p, (uint32_t *)ctx->ikey[1].p,
                                         (const uint32_t *)ctx->ipay[0].p, (uint32_t *)ctx->ipay[1].p, (size_t)nr,
                                         ibits + slk_bits_for((uint64_t)M - 1), s, true)))
            return rc;
        slk_prof_end(ctx, s);
    }
    ctx->shard_n = ctx->sh_n;  // committed
    return SLK_OK;
}

static int check_unit(slk_ctx *ctx, int32_t unit, const char *who) {
    if (ctx->shard_n < 0 || ctx->sh_M < 1) return slk_fail(ctx, SLK_EINVAL, "%s: no committed chunk", who);
    if (unit < 0 || unit >= ctx->sh_M * ctx->sh_S) return slk_fail(ctx, SLK_EINVAL, "%s: unit %d outside [0, %d)", who, unit, ctx->sh_M * ctx->sh_S);
    return SLK_OK;
}

SLK_EXPORT int slk_shard_gather(slk_ctx *ctx, const slk_tables *local, int32_t unit, float *d_rows_out,
                                void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g, /*shadow_ok=*/true))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if ((rc = check_unit(ctx, unit, "slk_shard_gather"))) return rc;
    const int64_t r0 = ctx->sh_rstart[unit], n_ids = ctx->sh_rstart[unit + 1] - r0;
    if (n_ids == 0) return SLK_OK;
    if (!d_rows_out) return slk_fail(ctx, SLK_EINVAL, "slk_shard_gather: d_rows_out is NULL");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const int32_t *d_ids = (const int32_t *)ctx->extra[SH_RID].p + r0;
    // inside a slk_bias_shadow_begin scope the trained item biases live in the ctx's interleaved copy
    const bool shadowed = ctx->shadow_active && ctx->shadow_src_p == local->d_param[3];
    const float *bias = shadowed ? (const float *)ctx->bias_shadow.p : (const float *)local->d_param[3];
    const uint32_t bsh = shadowed ? 1u : 0u;
    slk_prof_begin(ctx, SLK_K_EXCHANGE, s);
#define SLK_GATHER(V_, G_)                                                                                  \
    hipLaunchKernelGGL((k_shard_gather<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n_ids, 256 / G_)), dim3(256), 0, s, \
                       (const float *)local->d_param[1], bias, bsh, (int)local->dim,                         \
                       d_ids, (const uint32_t *)ctx->extra[SH_GSLOT].p + r0, n_ids, d_rows_out)
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
    if (upd == SLK_UPD_SGD) return k_shard_user_pass<VEC, G, SLK_UPD_SGD>;
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
                                   const slk_shard *sh, int32_t unit, int64_t global_batch, int32_t loss,
                                   const float *d_rows_in, float *d_grad_out, float *d_loss_out, int32_t accumulate,
                                   void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g, /*shadow_ok=*/true))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if ((rc = slk_check_optim(ctx, optim, 15u))) return rc;
    if ((rc = check_shard(ctx, sh))) return rc;
    if ((rc = check_unit(ctx, unit, "slk_shard_user_pass"))) return rc;
    if (loss < SLK_LOSS_POINTWISE || loss > SLK_LOSS_HINGE)
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_user_pass: pointwise/bpr/hinge (loss kind %d); adaptive hinge: slk_shard_score_pass, "
                                         "slk_shard_adaptive_select, slk_shard_user_pass_adaptive", loss);
    if (ctx->sh_adaptive || ctx->sh_NP != 2)
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_user_pass: the staged chunk is an adaptive-hinge one (%d lookups per interaction)", ctx->sh_NP);
    if (global_batch < 1) return slk_fail(ctx, SLK_EINVAL, "global_batch must be >= 1");
    const int64_t p0 = ctx->sh_ustart[unit], n = ctx->sh_ustart[unit + 1] - p0;
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
    a.begin = (uint32_t)p0;
    a.end = (uint32_t)(p0 + n);
    a.ukey = (const uint32_t *)ctx->ukey[1].p;
    a.umask = (uint32_t)((1ull << ctx->sh_ubits) - 1);
    a.vslot = (const uint32_t *)ctx->extra[SH_VSLOT].p;
    a.vrows = d_rows_in;
    a.grows = d_grad_out;
    a.loss_partial = (double *)ctx->losspart.p;
    a.loss_kind = loss;
    a.inv_b = 1.0f / (float)global_batch;
    a.ubz = (local->flags & SLK_TABLES_USER_BIAS_ZERO) && (loss == SLK_LOSS_BPR || loss == SLK_LOSS_HINGE) &&
            (optim->kind == SLK_OPT_ADAGRAD || optim->kind == SLK_OPT_SGD) && ctx->opt_user_bias_zero_hint;
    const unsigned gpb = 256u / (unsigned)g;
    slk_pass_fn upass = nullptr;
    const int upd = slk_upd_for(optim->kind);
#define SLK_PICK(V_, G_) upass = shard_user_pass_fn<V_, G_>(upd)
    SLK_FOR_LAYOUT(vec, g, SLK_PICK);
#undef SLK_PICK
    const unsigned ugrid = n > 0 ? slk_grid_for_fn(ctx, upass, (size_t)n, gpb) : 0;  // (capped at the kernel's occupancy)
    if (n > 0) {
        slk_prof_begin(ctx, SLK_K_USER_PASS, s);
        hipLaunchKernelGGL(upass, dim3(ugrid), dim3(256), 0, s, a);
        SLK_LAUNCH_CHECK(ctx, "k_shard_user_pass");
        slk_prof_end(ctx, s);
    }
    hipLaunchKernelGGL(k_shard_loss, dim3(1), dim3(256), 0, s, (const double *)ctx->losspart.p, (int)ugrid, a.inv_b,
                       d_loss_out, (int)accumulate);
    SLK_LAUNCH_CHECK(ctx, "k_shard_loss");
    return SLK_OK;
}

// ---- adaptive hinge: score phase, selection, user pass with dL/dscore given ---------------------------------------------------
template <int VEC, int G>
static slk_pass_fn shard_user_pass_pre_fn(int upd) {
    if (upd == SLK_UPD_ADAGRAD) return k_shard_user_pass_pre<VEC, G, SLK_UPD_ADAGRAD>;
    if (upd == SLK_UPD_SPARSE_ADAM) return k_shard_user_pass_pre<VEC, G, SLK_UPD_SPARSE_ADAM>;
    if (upd == SLK_UPD_SGD) return k_shard_user_pass_pre<VEC, G, SLK_UPD_SGD>;
    return k_shard_user_pass_pre<VEC, G, SLK_UPD_GRAD_ONLY>;
}

static int check_adaptive_unit(slk_ctx *ctx, int32_t unit, const char *who) {
    int rc;
    if ((rc = check_unit(ctx, unit, who))) return rc;
    if (!ctx->sh_adaptive || ctx->sh_NP < 2 || (ctx->sh_n > 0 && !ctx->extra[SH_GPOS].p))
        return slk_fail(ctx, SLK_EINVAL, "%s: the staged chunk was not begun by slk_shard_chunk_begin_adaptive", who);
    return SLK_OK;
}

static void fill_unit(slk_pass_args &a, slk_ctx *ctx, int32_t unit) {
    const int64_t p0 = ctx->sh_ustart[unit], n = ctx->sh_ustart[unit + 1] - p0;
    a.NP = ctx->sh_NP;
    a.begin = (uint32_t)p0;
    a.end = (uint32_t)(p0 + n);
    a.ukey = (const uint32_t *)ctx->ukey[1].p;
    a.umask = (uint32_t)((1ull << ctx->sh_ubits) - 1);
    a.vslot = (const uint32_t *)ctx->extra[SH_VSLOT].p;
    a.uk = (const uint32_t *)ctx->extra[SH_GPOS].p;  // sorted position -> row of the minibatch's score matrix
}

SLK_EXPORT int slk_shard_score_pass(slk_ctx *ctx, const slk_tables *local, int32_t unit, const float *d_rows_in,
                                    float *d_scores, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g, /*shadow_ok=*/true))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if ((rc = check_adaptive_unit(ctx, unit, "slk_shard_score_pass"))) return rc;
    const int64_t n = ctx->sh_ustart[unit + 1] - ctx->sh_ustart[unit];
    if (n == 0) return SLK_OK;
    if (!d_rows_in || !d_scores) return slk_fail(ctx, SLK_EINVAL, "slk_shard_score_pass: NULL pointer");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    slk_pass_args a;
    memset(&a, 0, sizeof(a));
    for (int t = 0; t < 4; ++t) a.P[t] = local->d_param[t];
    a.D = local->dim;
    fill_unit(a, ctx, unit);
    a.vrows = d_rows_in;
    a.sk = d_scores;
    slk_prof_begin(ctx, SLK_K_SCORE, s);
#define SLK_SCORE(V_, G_) \
    hipLaunchKernelGGL((k_shard_score_pass<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n, 256 / G_)), dim3(256), 0, s, a)
    SLK_FOR_LAYOUT(vec, g, SLK_SCORE);
#undef SLK_SCORE
    SLK_LAUNCH_CHECK(ctx, "k_shard_score_pass");
    slk_prof_end(ctx, s);
    return SLK_OK;
}

SLK_EXPORT int slk_shard_adaptive_select(slk_ctx *ctx, int64_t global_batch, int32_t n_neg, const float *d_scores,
                                         float *d_gk, float *d_loss_out, int32_t report_loss, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (global_batch < 1 || n_neg < 1 || global_batch * (int64_t)(n_neg + 1) >= ((int64_t)1 << 31))
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_adaptive_select: global_batch %lld x (1 + %d) pairs out of range", (long long)global_batch, n_neg);
    if (!d_scores || !d_gk || !d_loss_out) return slk_fail(ctx, SLK_EINVAL, "slk_shard_adaptive_select: NULL pointer");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    int rc;
    const unsigned max_grid = (unsigned)ctx->num_cus * (unsigned)(ctx->opt_user_grid_mult > 8 ? ctx->opt_user_grid_mult : 8);
    if ((rc = slk_ensure(ctx, ctx->losspart, (size_t)max_grid * 8))) return rc;
    const float inv_b = 1.0f / (float)global_batch;
    const unsigned sgrid = slk_grid_for(ctx, (size_t)global_batch, 256);
    slk_prof_begin(ctx, SLK_K_SCORE, s);
    hipLaunchKernelGGL(k_adaptive_select<0>, dim3(sgrid), dim3(256), 0, s, d_scores, d_gk, 0u, (uint32_t)global_batch, (int)n_neg,
                       inv_b, (double *)ctx->losspart.p, (const uint32_t *)nullptr, (uint32_t *)nullptr);
    SLK_LAUNCH_CHECK(ctx, "k_adaptive_select");
    slk_prof_end(ctx, s);
    // every rank has computed the whole minibatch's loss; the ranks' shares must ADD UP to it: one rank reports it
    hipLaunchKernelGGL(k_shard_loss, dim3(1), dim3(256), 0, s, (const double *)ctx->losspart.p, (int)(report_loss ? sgrid : 0), inv_b,
                       d_loss_out, 0);
    SLK_LAUNCH_CHECK(ctx, "k_shard_loss");
    return SLK_OK;
}

SLK_EXPORT int slk_shard_user_pass_adaptive(slk_ctx *ctx, const slk_tables *local, const slk_optim *optim, int32_t unit,
                                            const float *d_gk, const float *d_rows_in, float *d_grad_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g, /*shadow_ok=*/true))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if ((rc = slk_check_optim(ctx, optim, 15u))) return rc;
    if ((rc = check_adaptive_unit(ctx, unit, "slk_shard_user_pass_adaptive"))) return rc;
    const int64_t n = ctx->sh_ustart[unit + 1] - ctx->sh_ustart[unit];
    if (n == 0) return SLK_OK;
    if (!d_gk || !d_rows_in || !d_grad_out) return slk_fail(ctx, SLK_EINVAL, "slk_shard_user_pass_adaptive: NULL pointer");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const bool dense = optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE;
    if (dense) {
        const size_t elems[4] = {(size_t)local->num_users * local->dim, (size_t)local->num_items * local->dim,
                                 (size_t)local->num_users, (size_t)local->num_items};
        if ((rc = slk_ensure_dgrad(ctx, elems, 15u, s))) return rc;
    }
    slk_pass_args a;
    memset(&a, 0, sizeof(a));
    fill_tables(a, ctx, local, optim, dense);
    fill_unit(a, ctx, unit);
    a.gk = d_gk;
    a.vrows = d_rows_in;
    a.grows = d_grad_out;
    slk_pass_fn upass = nullptr;
    const int upd = slk_upd_for(optim->kind);
#define SLK_PICK(V_, G_) upass = shard_user_pass_pre_fn<V_, G_>(upd)
    SLK_FOR_LAYOUT(vec, g, SLK_PICK);
#undef SLK_PICK
    slk_prof_begin(ctx, SLK_K_USER_PASS, s);
    hipLaunchKernelGGL(upass, dim3(slk_grid_for_fn(ctx, upass, (size_t)n, 256u / (unsigned)g)), dim3(256), 0, s, a);
    SLK_LAUNCH_CHECK(ctx, "k_shard_user_pass_pre");
    slk_prof_end(ctx, s);
    return SLK_OK;
}

SLK_EXPORT int slk_shard_item_pass(slk_ctx *ctx, const slk_tables *local, slk_optim *optim, int32_t minibatch,
                                   const float *d_grad_in, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, local, 15u, &vec, &g, /*shadow_ok=*/true))) return rc;
    if ((rc = check_plain(ctx, local))) return rc;
    if ((rc = slk_check_optim(ctx, optim, 15u))) return rc;
    if (ctx->shard_n < 0 || minibatch < 0 || minibatch >= ctx->sh_M)
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_item_pass: no committed chunk / minibatch %d out of range", minibatch);
    const int S = ctx->sh_S;
    const int64_t r0 = ctx->sh_rstart[(size_t)minibatch * S], nr = ctx->sh_rstart[(size_t)(minibatch + 1) * S] - r0;
    if (nr > 0 && !d_grad_in) return slk_fail(ctx, SLK_EINVAL, "slk_shard_item_pass: d_grad_in is NULL");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const bool dense = optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE;
    const bool shadowed = ctx->shadow_active && ctx->shadow_src_p == local->d_param[3];
    if (shadowed && (optim->kind != SLK_OPT_ADAGRAD || optim->d_state1[3] != ctx->shadow_src_s))
        return slk_fail(ctx, SLK_EINVAL, "slk_shard_item_pass: the item biases are shadowed (slk_bias_shadow_begin) for another optimizer state");
    if (dense) {
        const size_t elems[4] = {(size_t)local->num_users * local->dim, (size_t)local->num_items * local->dim,
                                 (size_t)local->num_users, (size_t)local->num_items};
        if ((rc = slk_ensure_dgrad(ctx, elems, 15u, s))) return rc;
    }
    if (shadowed) ++ctx->stat_shadowed;
    if (nr > 0) {
        const unsigned ibits = slk_bits_for((uint64_t)local->num_items - 1);
        slk_pass_args a;
        memset(&a, 0, sizeof(a));
        fill_tables(a, ctx, local, optim, dense);
        if (shadowed) {  // {bias, Adagrad accumulator} interleaved: slk_bilinear.hip, slk_bias_shadow_begin
            a.P[3] = (float *)ctx->bias_shadow.p;
            a.S1[3] = (float *)ctx->bias_shadow.p + 1;
            a.bsh3 = 1;
        }
        a.snap = const_cast<float *>(d_grad_in);  // slots inside this minibatch's gradient buffer
        a.begin = 0;
        a.ibegin = (uint32_t)r0;
        a.iend = (uint32_t)(r0 + nr);
        a.ikey = (const uint32_t *)ctx->ikey[1].p;
        a.imask = (uint32_t)((1ull << ibits) - 1);
        a.ipay = (const uint32_t *)ctx->ipay[1].p;
        a.mb_loss_out = nullptr;
        slk_item_fns ipass = {nullptr, nullptr};
        const int upd = slk_upd_for(optim->kind);
#define SLK_PICK(V_, G_) ipass = slk_item_pass_fn<V_, G_, SLK_ITEM_BLK>(upd)
        SLK_FOR_LAYOUT(vec, g, SLK_PICK);
#undef SLK_PICK
        slk_prof_begin(ctx, SLK_K_ITEM_PASS, s);
        if ((rc = slk_launch_item_pass(ctx, ipass, a, g, s, "k_item_pass<BLK>"))) return rc;
        slk_prof_end(ctx, s);
    }
    if (dense && (rc = slk_dense_sweeps(ctx, local->d_param, optim, 15u, s))) return rc;
    optim->step += 1;
    return SLK_OK;
}
