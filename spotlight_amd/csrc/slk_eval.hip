// slk_eval.hip -- the evaluation side of the path: batched predict and on-GPU ranking.
//
// spotlight/evaluation.py:9-109 (mrr_score, sequence_mrr_score) calls model.predict() once per
// user / sequence -- one pass over the whole item table each (num_items * (4D + 4) bytes) -- and
// ranks the scores on the host with scipy.stats.rankdata.  Here a TILE of users (sequences) is
// scored per pass over the item table (the item rows are read once per 16 representations), and
// the rank of every held-out item is counted on the GPU:
//
//   slk_bilinear_scores  out[r][i] = <U[user_r], V[i]> + bu[user_r] + bi[i]      (BilinearNet.forward)
//   slk_poolnet_scores   out[r][i] = bi[i] + <final representation of sequence r, V[i]>  (PoolNet)
//   slk_rank_targets     rankdata(-out[row])[item] ('average' ties) for each (row, item) target, after
//                        out[row][excluded] = -FLT_MAX (the reference sets predictions[...] = FLOAT_MAX)
//
// Scores are formed exactly as k_predict / k_seq_predict form them (same lane layout, same
// summation order), so the batched path returns bit-identical numbers.
#include <float.h>
#include <math.h>

#include "slk_kernels.h"

enum { EV_REP = 24, EV_RBIAS };  // ctx->extra slots

#define SLK_EVAL_TILE 16  // representations per pass over the item table

// representation of row r for BilinearNet: the user's embedding vector and bias
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_eval_user_rows(const float *U, const float *bu, slk_bloom_dev ub, int D,
                                                        const int64_t *users, int64_t n, float *rep, float *rbias) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    for (int64_t r = (int64_t)blockIdx.x * GPB + grp; r < n; r += (int64_t)gridDim.x * GPB) {
        const int64_t u = users[r];
        const slk_vec<VEC> v = slk_emb_vec<VEC>(U, ub, (uint32_t)u, D, d0, on);
        if (on) slk_vstore<VEC>(rep + (size_t)r * D + d0, v);
        if (lane == 0) rbias[r] = bu[u];
    }
}

// representation of row r for PoolNet: user_representation's final state (sequence/implicit.py:331-335,
// sequence/representations.py:92-114): sum of the L item vectors / (per-dimension non-zero count + 1)
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_eval_seq_rows(const float *E, slk_bloom_dev ib, int D, const int64_t *seqs,
                                                       int64_t n, int L, float *rep) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    for (int64_t r = (int64_t)blockIdx.x * GPB + grp; r < n; r += (int64_t)gridDim.x * GPB) {
        const int64_t *seq = seqs + (size_t)r * L;
        slk_vec<VEC> S = slk_vzero<VEC>(), Cn = slk_vzero<VEC>();
        for (int t = 0; t < L; ++t) {
            const slk_vec<VEC> e = slk_emb_vec<VEC>(E, ib, (uint32_t)seq[t], D, d0, on);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                S.v[i] += e.v[i];
                Cn.v[i] += (e.v[i] != 0.0f) ? 1.0f : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) S.v[i] = S.v[i] / (Cn.v[i] + 1.0f);
        if (on) slk_vstore<VEC>(rep + (size_t)r * D + d0, S);
    }
}

// out[r][i] for a tile of SLK_EVAL_TILE representations (blockIdx.y) and all items: the tile's
// representations are staged in LDS, every row group streams item rows and scores them against
// the whole tile.  rbias == nullptr: no row bias (PoolNet).
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_eval_scores(const float *rep, const float *rbias, const float *V,
                                                     const float *bi, slk_bloom_dev ib, int D, int64_t n_rows,
                                                     int64_t n_items, float *out) {
    constexpr int GPB = 256 / G;
    constexpr int DL = G * VEC;
    __shared__ __attribute__((aligned(16))) float s_rep[SLK_EVAL_TILE * DL];
    __shared__ float s_rb[SLK_EVAL_TILE];
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const int64_t r0 = (int64_t)blockIdx.y * SLK_EVAL_TILE;
    const int nr = (n_rows - r0 < SLK_EVAL_TILE) ? (int)(n_rows - r0) : SLK_EVAL_TILE;
    for (int k = grp; k < SLK_EVAL_TILE; k += GPB) {
        const slk_vec<VEC> v = (on && k < nr) ? slk_vload<VEC>(rep + (size_t)(r0 + k) * D + d0) : slk_vzero<VEC>();
        slk_vstore<VEC>(s_rep + k * DL + d0, v);
        if (lane == 0) s_rb[k] = (rbias && k < nr) ? rbias[r0 + k] : 0.0f;
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * GPB + grp; i < n_items; i += (int64_t)gridDim.x * GPB) {
        const slk_vec<VEC> b = slk_emb_vec<VEC>(V, ib, (uint32_t)i, D, d0, on);
        const float bias = bi[i];
        for (int k = 0; k < nr; ++k) {
            const slk_vec<VEC> a = slk_vload<VEC>(s_rep + k * DL + d0);
            const float dot = slk_group_sum<G>(slk_vdot<VEC>(a, b));
            // k_predict: dot + bu + bi;  k_seq_predict: bi + dot
            const float s = rbias ? (dot + s_rb[k]) + bias : bias + dot;
            if (lane == 0) out[(size_t)(r0 + k) * n_items + i] = s;
        }
    }
}

__global__ __launch_bounds__(256) void k_eval_exclude(float *scores, int64_t n_items, const int64_t *rows,
                                                      const int64_t *items, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        scores[(size_t)rows[e] * n_items + items[e]] = -FLT_MAX;
}

// one workgroup per target: rank = #{j: s_j > s_t} + (#{j: s_j == s_t} + 1) / 2
// (= scipy.stats.rankdata(-scores)[t], method 'average': ties share the mean of their ranks)
__global__ __launch_bounds__(256) void k_eval_rank(const float *scores, int64_t n_items, const int64_t *rows,
                                                   const int64_t *items, int64_t n, double *rank) {
    __shared__ unsigned long long s_gt[256], s_eq[256];
    for (int64_t t = blockIdx.x; t < n; t += gridDim.x) {
        const float *row = scores + (size_t)rows[t] * n_items;
        const float st = row[items[t]];
        unsigned long long gt = 0, eq = 0;
        for (int64_t j = threadIdx.x; j < n_items; j += 256) {
            const float s = row[j];
            gt += s > st;
            eq += s == st;
        }
        s_gt[threadIdx.x] = gt;
        s_eq[threadIdx.x] = eq;
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off) {
                s_gt[threadIdx.x] += s_gt[threadIdx.x + off];
                s_eq[threadIdx.x] += s_eq[threadIdx.x + off];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) rank[t] = (double)s_gt[0] + ((double)s_eq[0] + 1.0) * 0.5;
        __syncthreads();
    }
}

static int eval_scores(slk_ctx *ctx, const slk_tables *tables, int vec, int g, const float *rep, const float *rbias,
                       int64_t n_rows, float *d_out, hipStream_t s) {
    slk_bloom_dev ibd;
    slk_bloom_to_dev(tables->item_bloom, &ibd);
    const unsigned tiles = (unsigned)((n_rows + SLK_EVAL_TILE - 1) / SLK_EVAL_TILE);
#define SLK_SCORES(V_, G_)                                                                                       \
    hipLaunchKernelGGL((k_eval_scores<V_, G_>),                                                                  \
                       dim3(slk_grid_for(ctx, (size_t)tables->num_items, 256 / G_, tiles >= 8 ? 2 : 8), tiles),   \
                       dim3(256), 0, s, rep, rbias, (const float *)tables->d_param[1],                           \
                       (const float *)tables->d_param[3], ibd, (int)tables->dim, n_rows, tables->num_items, d_out)
    SLK_FOR_LAYOUT(vec, g, SLK_SCORES);
#undef SLK_SCORES
    SLK_LAUNCH_CHECK(ctx, "k_eval_scores");
    return SLK_OK;
}

SLK_EXPORT int slk_bilinear_scores(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_users, int64_t n_users,
                                   float *d_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, tables, 15u, &vec, &g))) return rc;
    if (n_users < 0 || (n_users > 0 && (!d_users || !d_out))) return slk_fail(ctx, SLK_EINVAL, "slk_bilinear_scores: bad arguments");
    if (n_users == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const int D = tables->dim;
    if ((rc = slk_ensure(ctx, ctx->extra[EV_REP], (size_t)n_users * D * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[EV_RBIAS], (size_t)n_users * 4))) return rc;
    float *rep = (float *)ctx->extra[EV_REP].p, *rbias = (float *)ctx->extra[EV_RBIAS].p;
    slk_bloom_dev ubd;
    slk_bloom_to_dev(tables->user_bloom, &ubd);
    slk_prof_begin(ctx, SLK_K_SCORE, s);
#define SLK_ROWS(V_, G_)                                                                                          \
    hipLaunchKernelGGL((k_eval_user_rows<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n_users, 256 / G_)), dim3(256), 0, s, \
                       (const float *)tables->d_param[0], (const float *)tables->d_param[2], ubd, D, d_users,      \
                       n_users, rep, rbias)
    SLK_FOR_LAYOUT(vec, g, SLK_ROWS);
#undef SLK_ROWS
    SLK_LAUNCH_CHECK(ctx, "k_eval_user_rows");
    rc = eval_scores(ctx, tables, vec, g, rep, rbias, n_users, d_out, s);
    slk_prof_end(ctx, s);
    return rc;
}

SLK_EXPORT int slk_poolnet_scores(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_sequences, int64_t n_seq,
                                  int64_t seq_len, float *d_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, tables, 10u, &vec, &g))) return rc;
    if (n_seq < 0 || seq_len < 1 || (n_seq > 0 && (!d_sequences || !d_out)))
        return slk_fail(ctx, SLK_EINVAL, "slk_poolnet_scores: bad arguments");
    if (n_seq == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const int D = tables->dim;
    if ((rc = slk_ensure(ctx, ctx->extra[EV_REP], (size_t)n_seq * D * 4))) return rc;
    float *rep = (float *)ctx->extra[EV_REP].p;
    slk_bloom_dev ibd;
    slk_bloom_to_dev(tables->item_bloom, &ibd);
    slk_prof_begin(ctx, SLK_K_SCORE, s);
#define SLK_ROWS(V_, G_)                                                                                         \
    hipLaunchKernelGGL((k_eval_seq_rows<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n_seq, 256 / G_)), dim3(256), 0, s, \
                       (const float *)tables->d_param[1], ibd, D, d_sequences, n_seq, (int)seq_len, rep)
    SLK_FOR_LAYOUT(vec, g, SLK_ROWS);
#undef SLK_ROWS
    SLK_LAUNCH_CHECK(ctx, "k_eval_seq_rows");
    rc = eval_scores(ctx, tables, vec, g, rep, nullptr, n_seq, d_out, s);
    slk_prof_end(ctx, s);
    return rc;
}

SLK_EXPORT int slk_rank_targets(slk_ctx *ctx, float *d_scores, int64_t n_rows, int64_t num_items,
                                const int64_t *d_exc_rows, const int64_t *d_exc_items, int64_t n_exc,
                                const int64_t *d_tgt_rows, const int64_t *d_tgt_items, int64_t n_tgt,
                                double *d_rank_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (n_rows < 0 || num_items < 1 || n_exc < 0 || n_tgt < 0 || (n_rows > 0 && !d_scores) ||
        (n_exc > 0 && (!d_exc_rows || !d_exc_items)) || (n_tgt > 0 && (!d_tgt_rows || !d_tgt_items || !d_rank_out)))
        return slk_fail(ctx, SLK_EINVAL, "slk_rank_targets: bad arguments");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    slk_prof_begin(ctx, SLK_K_SCORE, s);
    if (n_exc > 0) {
        hipLaunchKernelGGL(k_eval_exclude, dim3(slk_grid_for(ctx, (size_t)n_exc, 256)), dim3(256), 0, s, d_scores,
                           num_items, d_exc_rows, d_exc_items, n_exc);
        SLK_LAUNCH_CHECK(ctx, "k_eval_exclude");
    }
    if (n_tgt > 0) {
        hipLaunchKernelGGL(k_eval_rank, dim3(slk_grid_for(ctx, (size_t)n_tgt, 1, 32)), dim3(256), 0, s,
                           (const float *)d_scores, num_items, d_tgt_rows, d_tgt_items, n_tgt, d_rank_out);
        SLK_LAUNCH_CHECK(ctx, "k_eval_rank");
    }
    slk_prof_end(ctx, s);
    return SLK_OK;
}
