// slk_kernels.h -- device templates shared by the engine's training paths
// (slk_bilinear.hip: one-GPU BilinearNet; slk_shard.hip: row-sharded BilinearNet;
// slk_seq.hip: PoolNet): kernel argument block, row optimizer updates, pair losses and the
// ITEM PASS (one owner group per unique item row: sum the row's gradient contributions, apply
// the optimizer once).
#pragma once
#include <math.h>

#include "slk_common.h"

// ---------------------------------------------------------------------------------------
// BloomEmbedding (spotlight/layers.py:74-244): the vector of id x is the sum of n_hash rows
//   row_h(x) = murmurhash3_32(int32 x, seed_h) mod rows   (signed hash, floor-mod; x == pad_id -> 0)
// of a compressed table.  The hash is ~15 integer ops: computed in-kernel instead of reading
// the reference's [num_embeddings, n_hash] int64 cache (8*H bytes per lookup).
// ---------------------------------------------------------------------------------------
struct slk_bloom_dev {
    uint32_t rows;    // compressed rows; n_hash == 0: plain table (row = id)
    int32_t n_hash;
    uint32_t pad_id;  // ~0u: none
    uint32_t seeds[8];
};

__device__ __forceinline__ uint32_t slk_murmur3_32(uint32_t k, uint32_t seed) {
    // MurmurHash3_x86_32 of the 4 little-endian bytes of one int32 key (sklearn.utils.murmurhash3_32)
    uint32_t h = seed;
    k *= 0xcc9e2d51u;
    k = (k << 15) | (k >> 17);
    k *= 0x1b873593u;
    h ^= k;
    h = (h << 13) | (h >> 19);
    h = h * 5u + 0xe6546b64u;
    h ^= 4u;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

__device__ __forceinline__ uint32_t slk_bloom_row(const slk_bloom_dev &b, uint32_t id, int h) {
    if (id == b.pad_id) return 0u;
    const int32_t hv = (int32_t)slk_murmur3_32(id, b.seeds[h]);
    int32_t v = hv % (int32_t)b.rows;  // rows < 2^31
    if (v < 0) v += (int32_t)b.rows;
    return (uint32_t)v;
}

// embedding vector of `id`: one row of a plain table, or embeddings(hashed).sum(1) (layers.py:236-242)
template <int VEC>
__device__ __forceinline__ slk_vec<VEC> slk_emb_vec(const float *T, const slk_bloom_dev &b, uint32_t id, int D,
                                                    int d0, bool on) {
    if (!on) return slk_vzero<VEC>();
    if (b.n_hash == 0) return slk_vload<VEC>(T + (size_t)id * D + d0);
    slk_vec<VEC> v = slk_vload<VEC>(T + (size_t)slk_bloom_row(b, id, 0) * D + d0);
    for (int h = 1; h < b.n_hash; ++h) {
        const slk_vec<VEC> x = slk_vload<VEC>(T + (size_t)slk_bloom_row(b, id, h) * D + d0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v.v[i] += x.v[i];
    }
    return v;
}

static inline void slk_bloom_to_dev(const slk_bloom *b, slk_bloom_dev *out) {
    memset(out, 0, sizeof(*out));
    out->pad_id = 0xffffffffu;
    if (!b) return;
    out->rows = (uint32_t)b->rows;
    out->n_hash = b->n_hash;
    out->pad_id = b->padding_idx < 0 ? 0xffffffffu : (uint32_t)b->padding_idx;
    for (int h = 0; h < 8; ++h) out->seeds[h] = b->seeds[h];
}

// ---------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------
struct slk_pass_args {
    float *P[4];   // tables (user_emb, item_emb, user_bias, item_bias)
    float *S1[4];  // optimizer state 1 / dense gradient buffer in *_DENSE modes
    float *S2[4];
    int D;
    int NP;                  // score pairs per interaction: 1 positive + nn negatives
    uint32_t begin, end;     // this minibatch's window in the user-sorted arrays
    const uint32_t *ukey;    // (minibatch << ubits) | user, sorted
    uint32_t umask;
    const uint32_t *uit;     // [pos*NP + s] item of pair s at sorted position pos
    const uint32_t *uk;      // sorted position -> chunk-local interaction index (PRE mode)
    const float *gk;         // PRE mode: dL/dscore per (interaction, pair)
    const float *ratings;    // explicit feedback, fused route: ratings[chunk-local interaction]; the user pass forms
                             // the score, the loss and dL/dscore itself (NP == 1) instead of reading gk
    float *sk;               // PRE mode: scores per (interaction, pair)
    float *snap;             // records, see slk_item_mode
    int RS;                  // record stride in floats
    float *gsn;              // SNAP / SEQ: dL/dscore of the minibatch's pairs, [payload - begin * NP] (kept out of the
                             // records so that a D = 64 record is exactly two aligned 128-B lines)
    uint32_t ibegin, iend;   // this minibatch's window in the item-sorted occurrence arrays
    const uint32_t *ikey;    // (minibatch << ibits) | item, sorted
    uint32_t imask;
    const uint32_t *ipay;    // occurrence -> record reference (see slk_item_mode)
    uint32_t pad_item;       // occurrences of this item row are never updated (padding_idx); ~0u = none
    uint32_t pad_item2;      // a second never-updated key (sentinel of non-head positions); ~0u = none
    slk_bloom_dev ub, ib;    // BloomEmbedding user / item layers (n_hash == 0: plain)
    float *urec;             // user-bloom: [(p - begin) * RSU] summed user-vector gradient of the
    int RSU;                 //   segment headed at sorted position p (applied by a ROW item pass)
    double *loss_partial;    // per-block partial loss sums
    int n_loss_partial;
    float *mb_loss_out;      // this minibatch's loss.item()
    int loss_kind;
    float inv_b;             // 1 / (minibatch size)   [sequences: 1 / mask.sum()]
    // row-sharded path: item rows arrive in / gradient rows leave through exchange buffers (slot layout: slk_blk_row)
    const float *vrows;      // slot -> item row (D floats) + bias
    float *grows;            // slot -> g * u_old (D floats) + g
    const uint32_t *vslot;   // [pos*NP + s] -> slot
    // optimizer coefficients, rounded from double on the host exactly as torch does
    float c_lr;    // Adagrad: clr.  SparseAdam: step_size.
    float c_eps;
    float c_omb1, c_omb2;  // 1-beta1, 1-beta2
    int nt;                // cache-policy bits (ctx option "nt")
};

enum { SLK_UPD_ADAGRAD = 0, SLK_UPD_SPARSE_ADAM = 1, SLK_UPD_GRAD_ONLY = 2 };

// How the item pass turns an occurrence payload r into a gradient contribution:
//   SNAP  r = pos*NP + s; record(pos) = [u_old (D)], g_s = gsn[r - begin*NP];  vec = g_s * u_old, bias g_s
//   SEQ   r = pos*NP + s; record(pos) = [repr (D) | hist (D)], g_s likewise;
//         vec = g_s * repr (+ hist when s == 0), bias g_s                       (PoolNet)
//   ROW   r = slot; record(slot) = [vec (D) | bias grad];                        (user-bloom rows)
//   BLK   r = slot of an exchange buffer (slk_blk_row / slk_blk_scalar): vec, bias grad  (row-sharded)
enum slk_item_mode { SLK_ITEM_SNAP = 0, SLK_ITEM_SEQ = 1, SLK_ITEM_ROW = 2, SLK_ITEM_BLK = 3 };

// Exchange buffers of the row-sharded path (slk_shard.hip): slots come in blocks of 64, a block is 64 rows of D floats
// followed by the 64 scalars (bias / bias gradient) of those rows.  Every row of a D = 64 table is then one aligned
// 256-B line on both sides of the wire and the scalars of 64 consecutive slots share one line (a [row | scalar]
// record of 260 B straddles three lines and makes every store a partial one); a block is 64 * (D + 1) floats, so a
// peer's segment is contiguous -- one all-to-all moves rows and scalars -- as long as segments start on block
// boundaries (the per-peer slot counts are rounded up to 64).
__device__ __forceinline__ size_t slk_blk_row(uint32_t slot, int D) {
    return (size_t)(slot >> 6) * (size_t)(SLK_SHARD_BLOCK * (D + 1)) + (size_t)(slot & 63u) * (size_t)D;
}
__device__ __forceinline__ size_t slk_blk_scalar(uint32_t slot, int D) {
    return (size_t)(slot >> 6) * (size_t)(SLK_SHARD_BLOCK * (D + 1)) + (size_t)SLK_SHARD_BLOCK * (size_t)D + (slot & 63u);
}

// Row update for the elements one lane owns.  GRAD_ONLY stores the summed gradient into the
// dense gradient buffer (aliased on S1) for the full-table sweep.
template <int VEC, int UPD>
__device__ __forceinline__ void slk_apply_vec(const slk_pass_args &a, int t, size_t off, slk_vec<VEC> &p,
                                              const slk_vec<VEC> &g, bool nt = false) {
    if (UPD == SLK_UPD_ADAGRAD) {
        // torch/optim/adagrad.py:360-385: sum += g^2; p += -clr * (g / (sqrt(sum) + eps))
        slk_vec<VEC> s = slk_vload_if_nt<VEC>(a.S1[t] + off, nt);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s.v[i] += g.v[i] * g.v[i];
            p.v[i] += -a.c_lr * (g.v[i] / (sqrtf(s.v[i]) + a.c_eps));
        }
        slk_vstore_if_nt<VEC>(a.S1[t] + off, s, nt);
        slk_vstore_if_nt<VEC>(a.P[t] + off, p, nt);
    } else if (UPD == SLK_UPD_SPARSE_ADAM) {
        // torch/optim/_functional.py:61-84
        slk_vec<VEC> m = slk_vload_if_nt<VEC>(a.S1[t] + off, nt);
        slk_vec<VEC> v = slk_vload_if_nt<VEC>(a.S2[t] + off, nt);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float mu = (g.v[i] - m.v[i]) * a.c_omb1;
            const float vu = (g.v[i] * g.v[i] - v.v[i]) * a.c_omb2;
            m.v[i] = mu + m.v[i];
            v.v[i] = vu + v.v[i];
            p.v[i] += -a.c_lr * (m.v[i] / (sqrtf(v.v[i]) + a.c_eps));
        }
        slk_vstore_if_nt<VEC>(a.S1[t] + off, m, nt);
        slk_vstore_if_nt<VEC>(a.S2[t] + off, v, nt);
        slk_vstore_if_nt<VEC>(a.P[t] + off, p, nt);
    } else {
        slk_vstore<VEC>(a.S1[t] + off, g);
    }
}

template <int UPD>
__device__ __forceinline__ void slk_apply_bias(const slk_pass_args &a, int t, size_t row, float g) {
    slk_vec<1> gv;
    gv.v[0] = g;
    if (UPD == SLK_UPD_ADAGRAD && g == 0.0f) return;  // exact no-op: sum += 0, p -= 0
    slk_vec<1> p = slk_vload<1>(a.P[t] + row);
    slk_apply_vec<1, UPD>(a, t, row, p, gv);
}

__device__ __forceinline__ float slk_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Loss of one (positive score, negative score) pair and its derivatives, already scaled by
// inv_b (spotlight/losses.py: pointwise :40-50, bpr :82-90, hinge :115-124).
__device__ __forceinline__ void slk_pair_loss(int loss_kind, float sp, float sn, float inv_b, float &l,
                                              float &gp, float &gn) {
    if (loss_kind == SLK_LOSS_BPR) {
        const float s = slk_sigmoid(sp - sn);
        l = 1.0f - s;
        gp = -(s * (1.0f - s)) * inv_b;
        gn = -gp;
    } else if (loss_kind == SLK_LOSS_HINGE || loss_kind == SLK_LOSS_ADAPTIVE_HINGE) {
        const float x = sn - sp + 1.0f;
        l = x > 0.0f ? x : 0.0f;
        gn = x >= 0.0f ? inv_b : 0.0f;  // clamp backward is inclusive at 0
        gp = -gn;
    } else {
        const float sa = slk_sigmoid(sp), sb = slk_sigmoid(sn);
        l = (1.0f - sa) + sb;
        gp = -(sa * (1.0f - sa)) * inv_b;
        gn = (sb * (1.0f - sb)) * inv_b;
    }
}

// loss of ONE predicted score against the observed rating and dL/dscore (losses.py:169-244), formed in fp32
// operation by operation as autograd forms them; bm = minibatch size, inv_b = 1 / bm.
__device__ __forceinline__ void slk_explicit_loss(int loss_kind, float sc, float r, float inv_b, uint32_t bm, float &l,
                                                  float &g) {
    if (loss_kind == SLK_LOSS_REGRESSION) {  // ((r - p) ** 2).mean()
        const float diff = r - sc;
        l = diff * diff;
        g = -(inv_b * (2.0f * diff));
    } else if (loss_kind == SLK_LOSS_POISSON) {  // p = exp(score); (p - r * log(p)).mean()
        const float p = expf(sc);
        l = p - r * logf(p);
        g = (inv_b + ((-inv_b) * r) / p) * p;
    } else {  // binary_cross_entropy_with_logits(score, clamp(r, 0, 1)), mean reduction
        const float t = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
        const float mx = -sc > 0.0f ? -sc : 0.0f;
        l = (1.0f - t) * sc + (mx + logf(expf(-mx) + expf(-sc - mx)));
        g = (slk_sigmoid(sc) - t) / (float)bm;
    }
}

// ---------------------------------------------------------------------------------------
// ITEM PASS
// ---------------------------------------------------------------------------------------
// One occurrence's contribution, read from its record.
template <int VEC, int MODE>
__device__ __forceinline__ void slk_item_contrib(const slk_pass_args &a, uint32_t r, int D, int d0, bool on,
                                                 slk_vec<VEC> &c, float &gb) {
    if (MODE == SLK_ITEM_ROW) {
        const float *rec = a.snap + (size_t)(r - a.begin) * a.RS;
        gb = rec[D];
        c = on ? slk_vload<VEC>(rec + d0) : slk_vzero<VEC>();
    } else if (MODE == SLK_ITEM_BLK) {
        gb = a.snap[slk_blk_scalar(r, D)];
        c = on ? slk_vload<VEC>(a.snap + slk_blk_row(r, D) + d0) : slk_vzero<VEC>();
    } else {
        const uint32_t NP = (uint32_t)a.NP;
        const uint32_t pos = (NP == 2) ? (r >> 1) : (r / NP);
        const uint32_t s = r - pos * NP;
        const float *rec = a.snap + (size_t)(pos - a.begin) * a.RS;
        if (MODE == SLK_ITEM_SNAP) {
            gb = a.gsn[r - a.begin * NP];
            // several negatives per interaction (adaptive hinge): only the positive and the selected
            // negative carry a gradient, so the row is fetched only when dL/dscore != 0 (a dependent
            // load; with one negative every occurrence is live and both loads are issued at once)
            const slk_vec<VEC> u = (on && (NP <= 2 || gb != 0.0f)) ? slk_vload<VEC>(rec + d0) : slk_vzero<VEC>();
#pragma unroll
            for (int i = 0; i < VEC; ++i) c.v[i] = gb * u.v[i];
        } else {
            gb = a.gsn[r - a.begin * NP];
            const slk_vec<VEC> u = on ? slk_vload<VEC>(rec + d0) : slk_vzero<VEC>();
#pragma unroll
            for (int i = 0; i < VEC; ++i) c.v[i] = gb * u.v[i];
            if (s == 0 && on) {
                const slk_vec<VEC> h = slk_vload<VEC>(rec + D + d0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) c.v[i] += h.v[i];
            }
        }
    }
}

// A block walks tiles of T = 4 * (256/G) consecutive positions of the item-sorted occurrence
// list.  Per tile: (1) keys + payloads -> LDS; wave 0 compacts the heads of the runs of equal keys
// that START in the tile into a list, head r belongs to row group r mod GPB (an even 1-2 heads per
// group); (2) a group issues the loads of its first head's item row + optimizer state TOGETHER
// with its four record gathers (every load independent: one HBM round trip covers both) and parks
// the contributions in LDS; (3) each run is summed from LDS by its owner group (runs that spill
// past the tile end are finished from global memory; rows of a run that started in an earlier
// tile are skipped -- its owner already took them) and the optimizer is applied to that item's
// row and bias.  Block 0 also reduces the loss partials of the preceding pass into loss.item().
// Measured on MI355X (profiles/README.md, r01_e): 0.415 -> 0.351 ms against the first version,
// which waited for the records, then made one dependent row round trip per head, back to back.
// PART: which of the run owner's two updates are applied -- the embedding row (table slot 1), the
// bias (slot 3), or both.  They separate when the embedding rows are BloomEmbedding rows (keys =
// hashed rows) while the bias table is indexed by the item id itself.
enum { SLK_PART_BOTH = 0, SLK_PART_ROWS = 1, SLK_PART_BIAS = 2 };
#ifndef SLK_ITEM_WAVES
#define SLK_ITEM_WAVES 7  // occupancy target of the item pass (waves per SIMD): 72 VGPRs, no spills with NPRE 1
#endif
#ifndef SLK_ITEM_NPRE
#define SLK_ITEM_NPRE 1  // heads per group whose row + state loads ride along with the record gather; measured
                         // (profiles/sweeps/r01_x): 7 waves x 1 head beats 6 x 2 (C2 0.340 -> 0.332, C5 0.651 -> 0.601 ms)
#endif
#ifndef SLK_SPILL_BATCH
#define SLK_SPILL_BATCH 2  // occurrences of a spilled run in flight per row group (4 spills VGPRs at 6 waves/SIMD)
#endif

// Row update with the parameter / first-state elements already in registers (loaded early, see
// k_item_pass).  Same arithmetic, in the same order, as slk_apply_vec.
template <int VEC, int UPD, bool S2PRE = false>
__device__ __forceinline__ void slk_apply_vec_pre(const slk_pass_args &a, int t, size_t off, slk_vec<VEC> &p,
                                                  slk_vec<VEC> &s, const slk_vec<VEC> &g,
                                                  const slk_vec<VEC> *s2 = nullptr, bool nt = false) {
    if (UPD == SLK_UPD_ADAGRAD) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s.v[i] += g.v[i] * g.v[i];
            p.v[i] += -a.c_lr * (g.v[i] / (sqrtf(s.v[i]) + a.c_eps));
        }
        slk_vstore_if_nt<VEC>(a.S1[t] + off, s, nt);
        slk_vstore_if_nt<VEC>(a.P[t] + off, p, nt);
    } else if (UPD == SLK_UPD_SPARSE_ADAM) {
        slk_vec<VEC> v = S2PRE ? *s2 : slk_vload<VEC>(a.S2[t] + off);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float mu = (g.v[i] - s.v[i]) * a.c_omb1;
            const float vu = (g.v[i] * g.v[i] - v.v[i]) * a.c_omb2;
            s.v[i] = mu + s.v[i];
            v.v[i] = vu + v.v[i];
            p.v[i] += -a.c_lr * (s.v[i] / (sqrtf(v.v[i]) + a.c_eps));
        }
        slk_vstore_if_nt<VEC>(a.S1[t] + off, s, nt);
        slk_vstore_if_nt<VEC>(a.S2[t] + off, v, nt);
        slk_vstore_if_nt<VEC>(a.P[t] + off, p, nt);
    } else {
        slk_vstore<VEC>(a.S1[t] + off, g);
    }
}

template <int VEC, int G, int UPD, int MODE, int PART = SLK_PART_BOTH>
__global__ __launch_bounds__(256) SLK_WAVES_PER_EU(SLK_ITEM_WAVES) void k_item_pass(slk_pass_args a) {
    constexpr int GPB = 256 / G;
    constexpr int T = 4 * GPB;
    constexpr int DL = G * VEC;  // LDS row length (>= D)
    constexpr int NPRE = SLK_ITEM_NPRE;  // heads per group whose rows are loaded early
    __shared__ double red[256];
    __shared__ uint32_t s_key[T + 1];  // s_key[i] = key of position tb - 1 + i
    __shared__ uint32_t s_pay[T];
    __shared__ float s_g[T];
    __shared__ uint8_t s_live[T];
    __shared__ uint16_t s_head[T];
    __shared__ int s_nheads;
    __shared__ __attribute__((aligned(16))) float s_row[T * DL];
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const bool rows_on = on && PART != SLK_PART_BIAS;
    const bool nt_rows = (SLK_NT_OF(a) & 2) != 0, nt_keys = (SLK_NT_OF(a) & 8) != 0;
    const uint32_t ibegin = a.ibegin, iend = a.iend;

    if (blockIdx.x == 0 && a.mb_loss_out) {
        double x = 0.0;
        for (int i = threadIdx.x; i < a.n_loss_partial; i += 256) x += a.loss_partial[i];
        const double tot = slk_block_sum_256(x, red);
        if (threadIdx.x == 0) *a.mb_loss_out = (float)(tot * (double)a.inv_b);
    }

    const uint32_t ntiles = (iend - ibegin + T - 1) / T;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t tb = ibegin + tile * T;
        const int tn = (iend - tb < (uint32_t)T) ? (int)(iend - tb) : T;
        const bool first_tile = tb == ibegin;
        __syncthreads();  // LDS of the previous tile no longer in use
        for (int i = threadIdx.x; i <= tn; i += 256)
            s_key[i] = (i == 0 && first_tile) ? 0u : slk_ld_u32(a.ikey + (tb - 1 + i), nt_keys);
        for (int i = threadIdx.x; i < tn; i += 256) s_pay[i] = slk_ld_u32(a.ipay + (tb + i), nt_keys);
        __syncthreads();

        // (1b) wave 0 compacts the heads of the runs that start in this tile
        if (threadIdx.x < 64) {
            int base = 0;
            for (int c = 0; c < T; c += 64) {
                const int j = c + (int)threadIdx.x;
                int f = 0;
                if (j < tn) {
                    const uint32_t key = s_key[j + 1];
                    const uint32_t item = key & a.imask;
                    // padding_idx rows receive no gradient: they never become heads
                    f = (((j == 0 && first_tile) || key != s_key[j]) && item != a.pad_item && item != a.pad_item2) ? 1 : 0;
                }
                int incl = f;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int up = __shfl_up(incl, d, 64);
                    if ((int)threadIdx.x >= d) incl += up;
                }
                if (f) s_head[base + incl - 1] = (uint16_t)j;
                base += __shfl(incl, 63, 64);
            }
            if (threadIdx.x == 0) s_nheads = base;
        }
        __syncthreads();
        const int nheads = s_nheads;

        // (2a) early loads: row + state (+ bias) of this group's first NPRE heads
        slk_vec<VEC> pv[NPRE], sv[NPRE];
        float pb[NPRE], sb[NPRE];
#pragma unroll
        for (int h = 0; h < NPRE; ++h) {
            pv[h] = slk_vzero<VEC>();
            sv[h] = slk_vzero<VEC>();
            pb[h] = sb[h] = 0.0f;
            const int r = grp + h * GPB;
            if (r < nheads) {
                const uint32_t item = s_key[(int)s_head[r] + 1] & a.imask;
                if (rows_on && UPD != SLK_UPD_GRAD_ONLY) {
                    const size_t voff = (size_t)item * D + d0;
                    pv[h] = slk_vload_if_nt<VEC>(a.P[1] + voff, nt_rows);
                    sv[h] = slk_vload_if_nt<VEC>(a.S1[1] + voff, nt_rows);
                }
                if (PART != SLK_PART_ROWS && UPD != SLK_UPD_GRAD_ONLY) {
                    pb[h] = a.P[3][item];
                    sb[h] = a.S1[3][item];
                }
            }
        }

        // (2b) gather: position j = grp + it * GPB
        slk_vec<VEC> c[4];
        float g[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int j = grp + it * GPB;
            g[it] = 0.0f;
            c[it] = slk_vzero<VEC>();
            // rows of the run inherited from the previous tile belong to that tile's owner
            bool mine = j < tn && (first_tile || s_key[j + 1] != s_key[0]);
            if (mine) {
                // never-updated keys (padding_idx rows, the dead entries of a live list) have no
                // record worth reading -- a dead entry's payload is not even a valid reference
                const uint32_t item = s_key[j + 1] & a.imask;
                mine = item != a.pad_item && item != a.pad_item2;
            }
            if (mine) slk_item_contrib<VEC, MODE>(a, s_pay[j], D, d0, rows_on, c[it], g[it]);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int j = grp + it * GPB;
            if (j < tn) {
                slk_vstore<VEC>(s_row + j * DL + d0, c[it]);
                if (lane == 0) {
                    s_g[j] = g[it];
                    s_live[j] = (MODE == SLK_ITEM_SNAP) ? (g[it] != 0.0f) : 1;
                }
            }
        }
        __syncthreads();

        // (3) head r -> group r mod GPB
        auto finish = [&](int j, bool pre, slk_vec<VEC> &p, slk_vec<VEC> &s, float bp, float bs) {
            const uint32_t key = s_key[j + 1];
            const uint32_t item = key & a.imask;
            slk_vec<VEC> gv = slk_vzero<VEC>();
            float gb = 0.0f;
            bool any = false;
            int k = j;
            do {
                if (s_live[k]) {
                    const slk_vec<VEC> cc = slk_vload<VEC>(s_row + k * DL + d0);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) gv.v[i] += cc.v[i];
                    gb += s_g[k];
                    any = true;
                }
                ++k;
            } while (k < tn && s_key[k + 1] == key);
            if (k == tn) {
                // The run may continue in the following tiles (long runs: BloomEmbedding rows shared
                // by many ids, skewed items).  The group reads G keys + payloads at once (the keys are
                // sorted, so the lanes that still match are a prefix), then takes the occurrences
                // SLK_SPILL_BATCH at a time: independent loads in flight, summed in occurrence order.
                uint32_t q = tb + tn;
                bool more = q < iend;
                while (more) {
                    const uint32_t qi = q + (uint32_t)lane;
                    uint32_t pq = 0u;
                    int cnt = 0;
                    if (qi < iend && a.ikey[qi] == key) {
                        pq = a.ipay[qi];
                        cnt = 1;
                    }
#pragma unroll
                    for (int m = G / 2; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, G);
                    for (int j0 = 0; j0 < cnt; j0 += SLK_SPILL_BATCH) {
                        slk_vec<VEC> cc[SLK_SPILL_BATCH];
                        float gq[SLK_SPILL_BATCH];
#pragma unroll
                        for (int e = 0; e < SLK_SPILL_BATCH; ++e) {
                            gq[e] = 0.0f;
                            cc[e] = slk_vzero<VEC>();
                            const uint32_t pj = __shfl(pq, (j0 + e) & (G - 1), G);
                            if (j0 + e < cnt) slk_item_contrib<VEC, MODE>(a, pj, D, d0, rows_on, cc[e], gq[e]);
                        }
#pragma unroll
                        for (int e = 0; e < SLK_SPILL_BATCH; ++e) {
                            if (j0 + e < cnt && (MODE != SLK_ITEM_SNAP || gq[e] != 0.0f)) {
#pragma unroll
                                for (int i = 0; i < VEC; ++i) gv.v[i] += cc[e].v[i];
                                gb += gq[e];
                                any = true;
                            }
                        }
                    }
                    more = cnt == G;
                    q += (uint32_t)G;
                }
            }
            if (UPD != SLK_UPD_SPARSE_ADAM && !any) return;
            const size_t voff = (size_t)item * D + d0;
            if (rows_on) {
                if (!pre && UPD != SLK_UPD_GRAD_ONLY) {
                    p = slk_vload_if_nt<VEC>(a.P[1] + voff, nt_rows);
                    s = slk_vload_if_nt<VEC>(a.S1[1] + voff, nt_rows);
                }
                slk_apply_vec_pre<VEC, UPD>(a, 1, voff, p, s, gv, nullptr, nt_rows);
            }
            if (lane == 0 && PART != SLK_PART_ROWS) {
                if (UPD == SLK_UPD_ADAGRAD && gb == 0.0f) return;  // exact no-op
                slk_vec<1> bpv, bsv, gbv;
                if (pre || UPD == SLK_UPD_GRAD_ONLY) {
                    bpv.v[0] = bp;
                    bsv.v[0] = bs;
                } else {
                    bpv.v[0] = a.P[3][item];
                    bsv.v[0] = a.S1[3][item];
                }
                gbv.v[0] = gb;
                slk_apply_vec_pre<1, UPD>(a, 3, item, bpv, bsv, gbv);
            }
        };
#pragma unroll
        for (int h = 0; h < NPRE; ++h) {
            const int r = grp + h * GPB;
            if (r < nheads) finish((int)s_head[r], true, pv[h], sv[h], pb[h], sb[h]);
        }
        for (int r = grp + NPRE * GPB; r < nheads; r += GPB) {
            slk_vec<VEC> p = slk_vzero<VEC>(), s = slk_vzero<VEC>();
            finish((int)s_head[r], false, p, s, 0.0f, 0.0f);
        }
    }
}

// ---------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------
// Row passes keep 8 workgroups (32 waves) per CU busy with grid-stride loops.  The item pass uses
// many more, smaller workgroups instead (mult = ctx->opt_item_grid_mult): at 69 VGPRs only 7 of 8
// fit a CU at once, and with equal work per workgroup the 8th ran alone in a second round.
static inline unsigned slk_grid_for(const slk_ctx *ctx, size_t work_items, unsigned per_block, int mult = 0) {
    size_t blocks = (work_items + per_block - 1) / per_block;
    const size_t cap = (size_t)ctx->num_cus * (size_t)(mult > 0 ? mult : ctx->opt_user_grid_mult);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// (VEC, G) layout for an embedding dim: 16 B per lane when dim % 4 == 0.
static inline bool slk_pick_layout(int D, int *vec, int *g) {
    if (D <= 0) return false;
    if (D % 4 == 0 && D <= 256) {
        *vec = 4;
        int need = D / 4, G = 1;
        while (G < need) G <<= 1;
        *g = G;
        return true;
    }
    if (D <= 64) {
        *vec = 1;
        int G = 1;
        while (G < D) G <<= 1;
        *g = G;
        return true;
    }
    return false;
}

#define SLK_FOR_LAYOUT(vec, g, MACRO)                                                 \
    do {                                                                              \
        if ((vec) == 4) {                                                             \
            switch (g) {                                                              \
                case 1: MACRO(4, 1); break;                                           \
                case 2: MACRO(4, 2); break;                                           \
                case 4: MACRO(4, 4); break;                                           \
                case 8: MACRO(4, 8); break;                                           \
                case 16: MACRO(4, 16); break;                                         \
                case 32: MACRO(4, 32); break;                                         \
                default: MACRO(4, 64); break;                                         \
            }                                                                         \
        } else {                                                                      \
            switch (g) {                                                              \
                case 1: MACRO(1, 1); break;                                           \
                case 2: MACRO(1, 2); break;                                           \
                case 4: MACRO(1, 4); break;                                           \
                case 8: MACRO(1, 8); break;                                           \
                case 16: MACRO(1, 16); break;                                         \
                case 32: MACRO(1, 32); break;                                         \
                default: MACRO(1, 64); break;                                         \
            }                                                                         \
        }                                                                             \
    } while (0)

typedef void (*slk_pass_fn)(slk_pass_args);

template <int VEC, int G, int MODE, int PART = SLK_PART_BOTH>
static slk_pass_fn slk_item_pass_fn(int upd) {
    if (upd == SLK_UPD_ADAGRAD) return k_item_pass<VEC, G, SLK_UPD_ADAGRAD, MODE, PART>;
    if (upd == SLK_UPD_SPARSE_ADAM) return k_item_pass<VEC, G, SLK_UPD_SPARSE_ADAM, MODE, PART>;
    return k_item_pass<VEC, G, SLK_UPD_GRAD_ONLY, MODE, PART>;
}

// Row-update mode of the fused passes for an optimizer kind.
static inline int slk_upd_for(int opt_kind) {
    if (opt_kind == SLK_OPT_ADAGRAD) return SLK_UPD_ADAGRAD;
    if (opt_kind == SLK_OPT_SPARSE_ADAM) return SLK_UPD_SPARSE_ADAM;
    return SLK_UPD_GRAD_ONLY;
}

// Per-step optimizer coefficients, formed in double and rounded to fp32 as torch does.
static inline void slk_set_opt_coeffs(slk_pass_args &a, const slk_optim *optim) {
    const double step = (double)(optim->step + 1);
    a.c_eps = (float)optim->eps;
    if (optim->kind == SLK_OPT_ADAGRAD) {
        a.c_lr = (float)(optim->lr / (1.0 + (step - 1.0) * optim->lr_decay));
    } else if (optim->kind == SLK_OPT_SPARSE_ADAM) {
        const double bc1 = 1.0 - pow(optim->beta1, step), bc2 = 1.0 - pow(optim->beta2, step);
        a.c_lr = (float)(optim->lr * sqrt(bc2) / bc1);
        a.c_omb1 = (float)(1.0 - optim->beta1);
        a.c_omb2 = (float)(1.0 - optim->beta2);
    }
}

// slk_bilinear.hip: validates the optimizer block (kinds, state pointers for tables in `mask`)
int slk_check_optim(slk_ctx *ctx, const slk_optim *optim, unsigned table_mask);
// slk_bilinear.hip: (re)allocates + zeroes the dense gradient buffers of the tables in `mask`
int slk_ensure_dgrad(slk_ctx *ctx, const size_t elems[4], unsigned table_mask, hipStream_t s);
// slk_bilinear.hip: full-table sweeps of the *_DENSE optimizers over the tables in `mask`
int slk_dense_sweeps(slk_ctx *ctx, float *const params[4], const slk_optim *optim, unsigned table_mask,
                     hipStream_t s);
