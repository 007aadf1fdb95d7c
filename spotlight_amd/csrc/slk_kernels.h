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

// THE score's dot product (predict / ranking; slk_eval.hip): the d-ordered chain acc = fmaf(a_d, b_d, acc) from acc = 0, which
// is bit for bit what v_mfma_f32_32x32x2_f32 computes when the scores are formed as a GEMM on the matrix cores.  Here on
// the vector unit for one (row, item) pair held by a row group: lane s continues the chain of lane s - 1 over its own VEC
// elements (G dependent steps: this form serves explicit pairs, whose cost is the row gathers).  Every lane gets the result.
template <int VEC, int G>
__device__ __forceinline__ float slk_chain_dot(const slk_vec<VEC> &a, const slk_vec<VEC> &b) {
    const int lane = threadIdx.x % G;
    float acc = 0.0f;
#pragma unroll
    for (int s = 0; s < G; ++s) {
        const float in = __shfl(acc, s > 0 ? s - 1 : 0, G);
        if (lane == s) {
            float c = s > 0 ? in : 0.0f;
#pragma unroll
            for (int i = 0; i < VEC; ++i) c = fmaf(a.v[i], b.v[i], c);
            acc = c;
        }
    }
    return __shfl(acc, G - 1, G);
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
    float *ipart;            // item pass: per-tile partial sums of the runs that cross tile boundaries, [2 * tile + which][IPS]
    uint32_t *ipart_meta;    //   [2 * tile + which][2] = {key, flags}; which 0 = the run inherited from the previous tile, 1 = the
    int IPS;                 //   run that continues into the next tile; IPS = floats per partial (row | bias gradient)
    uint32_t ipart_gen;      //   stamp of this launch (flags of other launches' partials are stale); ipart_count: long runs
    uint32_t *ipart_count;   //   that START in this launch's tiles (k_item_stitch leaves at once when there are none)
    float *upart;            // user pass, long runs (hot users): per-tile partial sums of the user gradient, [2 * tile + which][UPS],
    uint32_t *upart_meta;    //   metas and stamp exactly as for the item pass's partials (k_user_pass<..., ULONG>, k_user_stitch)
    int UPS;
    uint32_t upart_gen;
    uint32_t *upart_count;
    uint32_t bsh3;           // item-bias index shift: 0 = plain arrays; 1 = the interleaved shadow {bias, Adagrad sum} of a
                             // training scope (slk_bias_shadow_begin): P[3] = shadow, S1[3] = shadow + 1, element i at 2 i
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
    int ubz;               // 1: the user biases are identically zero and this pass cannot change them (slk_tables::flags): not fetched
    // user-row ping-pong of a training scope (slk_user_pingpong_begin, pair mode over a plain user table): the user table
    // exists twice.  uflag[u] = the copy that holds user u's current row (0 = P[0], 1 = P0alt).  The user pass reads the row
    // from that copy, writes the UPDATED row to the other one and flips the flag; the copy it read still holds the pre-step
    // row, so the item pass gathers u_old from there and the pass writes NO record.  gsn is then [position][pair] of
    // {dL/dscore, src}: src = user | (copy that holds the pre-step row) << 31.  uflag == nullptr: no ping-pong.
    float *P0alt;
    uint8_t *uflag;
    // single-occurrence fast path inside a ping-pong scope (catalogues far larger than a minibatch; slk_bilinear.hip): an item that
    // occurs ONCE in the minibatch is updated by the user pass, which holds everything the update needs; the item pass skips it.
    // mflag[r] (r = the occurrence's payload pos * 2 + s) / msorted[e] (item-sorted order): 1 = the item occurs more than once
    // in its minibatch and is the item pass's.  nullptr: every occurrence is the item pass's.
    const uint8_t *mflag;
    const uint8_t *msorted;
};

enum { SLK_UPD_ADAGRAD = 0, SLK_UPD_SPARSE_ADAM = 1, SLK_UPD_GRAD_ONLY = 2, SLK_UPD_SGD = 3 };
// a zero summed gradient leaves the row exactly as it is (no write needed): Adagrad (sum += 0, p -= 0) and SGD (p -= 0)
#define SLK_UPD_ZERO_IS_NOOP(UPD) ((UPD) == SLK_UPD_ADAGRAD || (UPD) == SLK_UPD_SGD)
// the row update reads first-state rows (S1): not SGD (stateless), not GRAD_ONLY (S1 is the gradient buffer it writes)
#define SLK_UPD_HAS_STATE(UPD) ((UPD) == SLK_UPD_ADAGRAD || (UPD) == SLK_UPD_SPARSE_ADAM)

// How the item pass turns an occurrence payload r into a gradient contribution:
//   SNAP  r = pos*NP + s; record(pos) = [u_old (D)], g_s = gsn[r - begin*NP];  vec = g_s * u_old, bias g_s
//   SEQ   r = pos*NP + s; record(pos) = [repr (D) | hist (D)], g_s likewise;
//         vec = g_s * repr (+ hist when s == 0), bias g_s                       (PoolNet)
//   ROW   r = slot; record(slot) = [vec (D) | bias grad];                        (user-bloom rows)
//   BLK   r = slot of an exchange buffer (slk_blk_row / slk_blk_scalar): vec, bias grad  (row-sharded)
//   SNAPPP  as SNAP with NP == 2 inside a user-row ping-pong scope: no record; gsn[r - begin*2] = {g_s, src} (8 bytes),
//         u_old = the row of user (src & 0x7fffffff) in the copy (src >> 31) of the user table (slk_pass_args::P0alt)
//   SNAPPPS as SNAPPP; occurrences whose item occurs once in the minibatch (msorted[e] == 0) are skipped: the user pass updated them
enum slk_item_mode { SLK_ITEM_SNAP = 0, SLK_ITEM_SEQ = 1, SLK_ITEM_ROW = 2, SLK_ITEM_BLK = 3, SLK_ITEM_SNAPPP = 4, SLK_ITEM_SNAPPPS = 5 };
#define SLK_ITEM_IS_PP(MODE) ((MODE) == SLK_ITEM_SNAPPP || (MODE) == SLK_ITEM_SNAPPPS)
#define SLK_ITEM_IS_SNAP(MODE) ((MODE) == SLK_ITEM_SNAP || SLK_ITEM_IS_PP(MODE))

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

// index of item i's bias (and of its optimizer state) in P[3] / S1[3] / S2[3]
#define SLK_B3(a_, i_) ((size_t)(i_) << (a_).bsh3)

// Row update for the elements one lane owns.  GRAD_ONLY stores the summed gradient into the
// dense gradient buffer (aliased on S1) for the full-table sweep.
template <int VEC, int UPD>
__device__ __forceinline__ void slk_apply_vec(const slk_pass_args &a, int t, size_t off, slk_vec<VEC> &p,
                                              const slk_vec<VEC> &g, bool nt = false, float *pdst = nullptr) {
    // pdst: where the updated parameter elements go (user-row ping-pong: the other copy of the table); default in place
    float *const pout = pdst ? pdst : a.P[t] + off;
    if (UPD == SLK_UPD_ADAGRAD) {
        // torch/optim/adagrad.py:360-385: sum += g^2; p += -clr * (g / (sqrt(sum) + eps))
        slk_vec<VEC> s = slk_vload_if_nt<VEC>(a.S1[t] + off, nt);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s.v[i] += g.v[i] * g.v[i];
            p.v[i] += -a.c_lr * (g.v[i] / (sqrtf(s.v[i]) + a.c_eps));
        }
        slk_vstore_if_nt<VEC>(a.S1[t] + off, s, nt);
        slk_vstore_if_nt<VEC>(pout, p, nt);
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
        slk_vstore_if_nt<VEC>(pout, p, nt);
    } else if (UPD == SLK_UPD_SGD) {
        // torch/optim/sgd.py (momentum 0, weight_decay 0): param.add_(grad, alpha=-lr)
#pragma unroll
        for (int i = 0; i < VEC; ++i) p.v[i] += -a.c_lr * g.v[i];
        slk_vstore_if_nt<VEC>(pout, p, nt);
    } else {
        slk_vstore<VEC>(a.S1[t] + off, g);
    }
}

template <int UPD>
__device__ __forceinline__ void slk_apply_bias(const slk_pass_args &a, int t, size_t row, float g) {
    slk_vec<1> gv;
    gv.v[0] = g;
    if (SLK_UPD_ZERO_IS_NOOP(UPD) && g == 0.0f) return;  // exact no-op: sum += 0, p -= 0
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
        if (SLK_ITEM_IS_PP(MODE)) {
            // user-row ping-pong: {dL/dscore, src} in one 8-byte load, then the pre-step user row from the copy the user pass
            // left untouched (no record was written)
            const uint2 t = reinterpret_cast<const uint2 *>(a.gsn)[r - a.begin * 2u];
            memcpy(&gb, &t.x, 4);
            const float *urow = ((t.y >> 31) ? a.P0alt : a.P[0]) + (size_t)(t.y & 0x7fffffffu) * D;
            const slk_vec<VEC> u = on ? slk_vload_if_nt<VEC>(urow + d0, (SLK_NT_OF(a) & 32) != 0) : slk_vzero<VEC>();
#pragma unroll
            for (int i = 0; i < VEC; ++i) c.v[i] = gb * u.v[i];
        } else if (MODE == SLK_ITEM_SNAP) {
            gb = a.gsn[r - a.begin * NP];
            // several negatives per interaction (adaptive hinge): only the positive and the selected
            // negative carry a gradient, so the row is fetched only when dL/dscore != 0 (a dependent
            // load; with one negative every occurrence is live and both loads are issued at once)
            const slk_vec<VEC> u = (on && (NP <= 2 || gb != 0.0f)) ? slk_vload_if_nt<VEC>(rec + d0, (SLK_NT_OF(a) & 32) != 0) : slk_vzero<VEC>();
#pragma unroll
            for (int i = 0; i < VEC; ++i) c.v[i] = gb * u.v[i];
        } else {
            // PoolNet (slk_seq.hip): record = [representation | the own item's ready-made contribution g * representation +
            // history gradient].  The own item (s == 0) takes the second half as it is, a sampled item g times the first:
            // ONE 4D-byte read per occurrence (rounds 1-4: both halves for s == 0)
            gb = a.gsn[r - a.begin * NP];
            const slk_vec<VEC> u = on ? slk_vload<VEC>(rec + (s == 0 ? D : 0) + d0) : slk_vzero<VEC>();
#pragma unroll
            for (int i = 0; i < VEC; ++i) c.v[i] = s == 0 ? u.v[i] : gb * u.v[i];
        }
    }
}

// A block walks tiles of T = 4 * (256/G) consecutive positions of the item-sorted occurrence
// list.  Per tile: (1) keys + payloads -> LDS; wave 0 compacts the heads of the runs of equal keys
// that START in the tile into a list (plus position 0 when it continues the previous tile's last run), head r belongs
// to row group r mod GPB (an even 1-2 heads per group); (2) a group issues the loads of its first head's item row +
// optimizer state TOGETHER with its four record gathers (every load independent: one HBM round trip covers both) and
// parks the contributions in LDS; (3) each run is summed from LDS by its owner group (a run that spills into the next
// tile is finished from global memory; rows of a run that started in the previous tile are skipped -- its owner already
// took them) and the optimizer is applied to that item's row and bias.
// LONG runs -- a run that wholly covers at least one tile: a popular item of a skewed dataset, a hashed row of a small
// bloom table -- are not walked by their owner (one row group taking tens of thousands of records a batch of loads at a
// time: Zipf(1.0) positives at C2 sizes cost 40 ms per item pass that way, profiles/README.md).  Every tile sums ITS part of
// such a run from LDS and writes it out as a PARTIAL (row sum, bias-gradient sum, key, flags; at most two per tile: the
// run inherited from the previous tile, the run handed to the next one), and k_item_stitch, launched behind the pass,
// adds a long run's partials in tile order and applies the update.  Whether a run is long is decided the same way by
// every tile that sees a piece of it (the head tile looks at the last position of the next tile, the tile a run ends in
// at the first position of the previous one; a last tile of fewer than T positions never counts).  Short runs are summed in occurrence order exactly as before; a long run
// tile by tile, tiles in order -- the persistent epoch kernel's serial walk applies the same rule, so the two routes stay
// bit-identical.  Block 0 also reduces the loss partials of the preceding pass into loss.item().
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

#ifndef SLK_ITEM_KEYPF
#define SLK_ITEM_KEYPF 1       // 1: the keys + payloads of the workgroup's NEXT tile are fetched into registers behind the
                               // record gathers of the current one (tiles of up to 254 positions, i.e. row groups of >= 4 lanes):
                               // one dependent round trip less per tile.  Same-box A/B (profiles/r03_b_*): C2 item pass
                               // 0.3135 -> 0.2935 ms, minibatch 65 536 45.3 -> 43.1 us, C5 shard unchanged
#endif

#ifndef SLK_ITEM_GSPF
#define SLK_ITEM_GSPF 1        // 1: inside a ping-pong scope a tile's {dL/dscore, src} pairs are fetched at the top of the tile (k_item_pass: GSPF)
#endif

#ifndef SLK_USER_TILE
#define SLK_USER_TILE 32  // user pass: a user run that wholly covers an aligned tile of this many positions is LONG (k_user_pass<ULONG>)
#endif

#ifndef SLK_STITCH_BATCH
#define SLK_STITCH_BATCH 8  // partials of a long run in flight per row group in the stitch kernels (a dynamic-trip-count loop of
                            // dependent load + add round trips otherwise: ~1 us per tile of a run that fills thousands)
#endif

#ifndef SLK_SPILL_BATCH
#define SLK_SPILL_BATCH 2  // occurrences of a spilled run in flight per row group (4 spills VGPRs at 6 waves/SIMD; measured again in round 3
                           // at 7: C3 1.78 -> 1.81 ms, C4 0.55 -> 0.61, C2 item pass 0.296 -> 0.310, profiles/r03_zf_*)
#endif

// Row update with the parameter / first-state elements already in registers (loaded early, see
// k_item_pass).  Same arithmetic, in the same order, as slk_apply_vec.
template <int VEC, int UPD, bool S2PRE = false>
__device__ __forceinline__ void slk_apply_vec_pre(const slk_pass_args &a, int t, size_t off, slk_vec<VEC> &p,
                                                  slk_vec<VEC> &s, const slk_vec<VEC> &g,
                                                  const slk_vec<VEC> *s2 = nullptr, bool nt = false, float *pdst = nullptr) {
    float *const pout = pdst ? pdst : a.P[t] + off;
    if (UPD == SLK_UPD_ADAGRAD) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            s.v[i] += g.v[i] * g.v[i];
            p.v[i] += -a.c_lr * (g.v[i] / (sqrtf(s.v[i]) + a.c_eps));
        }
        slk_vstore_if_nt<VEC>(a.S1[t] + off, s, nt);
        slk_vstore_if_nt<VEC>(pout, p, nt);
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
        slk_vstore_if_nt<VEC>(pout, p, nt);
    } else if (UPD == SLK_UPD_SGD) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) p.v[i] += -a.c_lr * g.v[i];
        slk_vstore_if_nt<VEC>(pout, p, nt);
    } else {
        slk_vstore<VEC>(a.S1[t] + off, g);
    }
}

// flags of a partial (slk_pass_args::ipart_meta[slot][1])
enum { SLK_IPART_STARTS = 2, SLK_IPART_ENDS = 4, SLK_IPART_ANY = 8, SLK_IPART_GEN_SHIFT = 4 };  // flags | launch stamp << 4

// Applies the summed gradient (gv, gb) of one item row: the update both k_item_pass (runs inside one tile) and
// k_item_stitch (runs across tiles) end with.  `pre`: the row / bias and their first state were loaded early.
template <int VEC, int G, int UPD, int PART>
__device__ __forceinline__ void slk_item_apply(const slk_pass_args &a, uint32_t item, int D, int d0, int lane, bool rows_on,
                                               bool nt_rows, bool any, const slk_vec<VEC> &gv, float gb, bool pre,
                                               slk_vec<VEC> &p, slk_vec<VEC> &s, float bp, float bs) {
    if (UPD != SLK_UPD_SPARSE_ADAM && !any) return;
    const size_t voff = (size_t)item * D + d0;
    if (rows_on) {
        if (!pre && UPD != SLK_UPD_GRAD_ONLY) {
            p = slk_vload_if_nt<VEC>(a.P[1] + voff, nt_rows);
            if (SLK_UPD_HAS_STATE(UPD)) s = slk_vload_if_nt<VEC>(a.S1[1] + voff, nt_rows);
        }
        slk_apply_vec_pre<VEC, UPD>(a, 1, voff, p, s, gv, nullptr, nt_rows);
    }
    if (lane == 0 && PART != SLK_PART_ROWS) {
        if (SLK_UPD_ZERO_IS_NOOP(UPD) && gb == 0.0f) return;  // exact no-op
        slk_vec<1> bpv, bsv, gbv;
        if (pre || UPD == SLK_UPD_GRAD_ONLY) {
            bpv.v[0] = bp;
            bsv.v[0] = bs;
        } else {
            bpv.v[0] = a.P[3][SLK_B3(a, item)];
            bsv.v[0] = SLK_UPD_HAS_STATE(UPD) ? a.S1[3][SLK_B3(a, item)] : 0.0f;
        }
        gbv.v[0] = gb;
        slk_apply_vec_pre<1, UPD>(a, 3, SLK_B3(a, item), bpv, bsv, gbv);
    }
}

// LONG = false: for a minibatch in which NO run wholly covers a tile (the host knows: k_item_long_flags) -- no run is long,
// nothing is looked up or written for the partial scheme, no stitch kernel follows; results are those of LONG = true.
// NPRE_: heads per group whose row + state loads ride along with the record gather.  SLK_ITEM_NPRE (1) is the bandwidth-bound
// form (C2 minibatches: occupancy 7 matters more than the dependent round trip of a group's second head); 4 = every head a
// group can own, for launches of so few tiles that each workgroup takes ONE and the pass is that tile's chain of dependent
// round trips (minibatches up to ~10^5 occurrences, PoolNet's default 256 sequences): the occupancy cap is lifted.
template <int VEC, int G, int UPD, int MODE, int PART = SLK_PART_BOTH, bool LONG = true, int NPRE_ = SLK_ITEM_NPRE>
__global__ __launch_bounds__(256) SLK_WAVES_PER_EU(NPRE_ > SLK_ITEM_NPRE ? 4 : SLK_ITEM_WAVES) void k_item_pass(slk_pass_args a) {
    constexpr int GPB = 256 / G;
    constexpr int T = 4 * GPB;
    constexpr int DL = G * VEC;  // LDS row length (>= D)
    constexpr int NPRE = NPRE_;  // heads per group whose rows are loaded early
    __shared__ double red[256];
    __shared__ uint32_t s_key[T + 1 + (LONG ? 1 : 0)];  // s_key[i] = key of position tb - 1 + i (LONG: i = tn + 1 = the next tile's first key)
    __shared__ uint32_t s_far[3];      // key of the previous tile's first position, of the next (full) tile's last position
    __shared__ uint32_t s_pay[T];
    __shared__ float s_g[T];
    __shared__ uint32_t s_src[T];      // SNAPPP: where the position's pre-step user row stands (GSPF below)
    __shared__ uint8_t s_multi[T];     // SNAPPPS: 0 = the position's item occurs once in the minibatch (the user pass updated it)
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
    // KEYPF: thread i holds key i - 1 (i <= tn + 1) and payload i (i < tn) of the NEXT tile, threads 64 / 65 its far keys
    constexpr bool KEYPF = SLK_ITEM_KEYPF != 0 && T + 2 <= 256;
    uint32_t pf_key = 0u, pf_pay = 0u, pf_far = 0u, pf_far2 = 0u;
    constexpr bool SKIP1 = MODE == SLK_ITEM_SNAPPPS;
    uint32_t pf_multi = 1u;
    auto tile_fetch = [&](uint32_t tile) {
        const uint32_t tb = ibegin + tile * T;
        const int tn = (iend - tb < (uint32_t)T) ? (int)(iend - tb) : T;
        const bool first_tile = tb == ibegin;
        const bool has_next = tb + (uint32_t)tn < iend;
        const int i = (int)threadIdx.x;
        pf_key = 0u;
        if (i == 0) pf_key = first_tile ? 0u : slk_ld_u32(a.ikey + (tb - 1), nt_keys);
        else if (i <= tn) pf_key = slk_ld_u32(a.ikey + (tb - 1 + i), nt_keys);
        else if (LONG && i == tn + 1 && has_next) pf_key = slk_ld_u32(a.ikey + (tb + tn), nt_keys);
        if (i < tn) pf_pay = slk_ld_u32(a.ipay + (tb + i), nt_keys);
        if (SKIP1 && i < tn) pf_multi = (uint32_t)a.msorted[tb + i];
        if (LONG && i == 64) pf_far = first_tile ? 0u : slk_ld_u32(a.ikey + (tb - T), nt_keys);
        if (LONG && i == 65) {
            const bool next_full = has_next && iend - (tb + (uint32_t)tn) >= (uint32_t)T;
            pf_far = next_full ? slk_ld_u32(a.ikey + (tb + 2u * (uint32_t)T - 1u), nt_keys) : 0u;
            pf_far2 = next_full ? 1u : 0u;
        }
    };
    // GSPF (ping-pong): a position's {dL/dscore, src} pair is fetched by the thread that holds its payload at the TOP of the tile --
    // in flight while wave 0 compacts the heads -- and handed to the row groups through LDS, so that the gather of the pre-step user
    // rows goes out WITH the head's row + state loads instead of one dependent round trip behind them
    constexpr bool GSPF = SLK_ITEM_IS_PP(MODE) && KEYPF && SLK_ITEM_GSPF != 0;
    if (KEYPF && blockIdx.x < ntiles) tile_fetch(blockIdx.x);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t tb = ibegin + tile * T;
        const int tn = (iend - tb < (uint32_t)T) ? (int)(iend - tb) : T;
        const bool first_tile = tb == ibegin;
        const bool has_next = tb + (uint32_t)tn < iend;
        __syncthreads();  // LDS of the previous tile no longer in use
        uint2 gs_pf = make_uint2(0u, 0u);
        if (GSPF && (int)threadIdx.x < tn && (!SKIP1 || pf_multi)) gs_pf = reinterpret_cast<const uint2 *>(a.gsn)[pf_pay - a.begin * 2u];
        if (KEYPF) {
            const int i = (int)threadIdx.x;
            if (i <= tn + (LONG ? 1 : 0)) s_key[i] = pf_key;
            if (i < tn) s_pay[i] = pf_pay;
            if (SKIP1 && i < tn) s_multi[i] = (uint8_t)pf_multi;
            if (LONG && i == 64) s_far[0] = pf_far;
            if (LONG && i == 65) {
                s_far[1] = pf_far;
                s_far[2] = pf_far2;
            }
        } else {
        for (int i = threadIdx.x; i <= tn + (LONG ? 1 : 0); i += 256) {
            uint32_t kv = 0u;
            if (i == 0) kv = first_tile ? 0u : slk_ld_u32(a.ikey + (tb - 1), nt_keys);
            else if (i <= tn) kv = slk_ld_u32(a.ikey + (tb - 1 + i), nt_keys);
            else if (LONG && has_next) kv = slk_ld_u32(a.ikey + (tb + tn), nt_keys);
            s_key[i] = kv;
        }
        for (int i = threadIdx.x; i < tn; i += 256) s_pay[i] = slk_ld_u32(a.ipay + (tb + i), nt_keys);
        for (int i = threadIdx.x; SKIP1 && i < tn; i += 256) s_multi[i] = a.msorted[tb + i];
        if (LONG && threadIdx.x == 64) s_far[0] = first_tile ? 0u : slk_ld_u32(a.ikey + (tb - T), nt_keys);
        if (LONG && threadIdx.x == 65) {
            // the next tile's last key -- if the next tile is a full one (only full tiles make a run long)
            const bool next_full = has_next && iend - (tb + (uint32_t)tn) >= (uint32_t)T;
            s_far[1] = next_full ? slk_ld_u32(a.ikey + (tb + 2u * (uint32_t)T - 1u), nt_keys) : 0u;
            s_far[2] = next_full ? 1u : 0u;
        }
        }
        __syncthreads();
        // the run the previous tile hands over, the run handed to the next tile; each of them is LONG (summed through
        // partials) iff it wholly covers some tile: this one, the previous one (inherited run), the next one (handed over)
        const bool inherits = !first_tile && s_key[1] == s_key[0];
        // (LONG = false: whether the last run continues is found out by its owner's walk, as before the partial scheme)
        const bool hands_over = LONG && has_next && s_key[tn + (LONG ? 1 : 0)] == s_key[tn];
        const bool all_same = tn == T && s_key[1] == s_key[tn];  // this (full) tile is one run's
        const bool inherits_long = LONG && inherits && (all_same || s_far[0] == s_key[0]);
        const bool hands_over_long = LONG && hands_over && (all_same || (s_far[2] != 0u && s_far[1] == s_key[tn]));
        (void)s_far;

        // (1b) wave 0 compacts the heads of the runs that start in this tile; position 0 is also listed when it
        // continues a LONG run of the previous tile
        if (threadIdx.x < 64) {
            int base = 0;
            for (int c = 0; c < T; c += 64) {
                const int j = c + (int)threadIdx.x;
                int f = 0;
                if (j < tn) {
                    const uint32_t key = s_key[j + 1];
                    const uint32_t item = key & a.imask;
                    // padding_idx rows receive no gradient: they never become heads
                    f = (((j == 0 && (first_tile || inherits_long)) || key != s_key[j]) && item != a.pad_item &&
                         item != a.pad_item2 && (!SKIP1 || s_multi[j])) ? 1 : 0;  // (a once-only item is not a head: nothing to do)
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
        if (GSPF && (int)threadIdx.x < tn) {
            memcpy(&s_g[threadIdx.x], &gs_pf.x, 4);
            s_src[threadIdx.x] = gs_pf.y;
        }
        __syncthreads();
        const int nheads = s_nheads;
        // A head's run gets its update here unless it is a piece of a LONG run: the inherited one (position 0) or the one
        // that reaches the tile's end and is handed over (the LAST head).
        auto completes = [&](int r) -> bool {
            const int j = (int)s_head[r];
            if (j == 0 && inherits_long) return false;
            if (r == nheads - 1 && hands_over_long && s_key[j + 1] == s_key[tn]) return false;
            return true;
        };

        // (2a) early loads: row + state (+ bias) of this group's first NPRE heads
        slk_vec<VEC> pv[NPRE], sv[NPRE];
        float pb[NPRE], sb[NPRE];
#pragma unroll
        for (int h = 0; h < NPRE; ++h) {
            pv[h] = slk_vzero<VEC>();
            sv[h] = slk_vzero<VEC>();
            pb[h] = sb[h] = 0.0f;
            const int r = grp + h * GPB;
            if (r < nheads && completes(r)) {
                const uint32_t item = s_key[(int)s_head[r] + 1] & a.imask;
                if (rows_on && UPD != SLK_UPD_GRAD_ONLY) {
                    const size_t voff = (size_t)item * D + d0;
                    pv[h] = slk_vload_if_nt<VEC>(a.P[1] + voff, nt_rows);
                    if (SLK_UPD_HAS_STATE(UPD)) sv[h] = slk_vload_if_nt<VEC>(a.S1[1] + voff, nt_rows);
                }
                if (PART != SLK_PART_ROWS && UPD != SLK_UPD_GRAD_ONLY) {
                    pb[h] = a.P[3][SLK_B3(a, item)];
                    if (SLK_UPD_HAS_STATE(UPD)) sb[h] = a.S1[3][SLK_B3(a, item)];
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
            // rows of a SHORT run inherited from the previous tile belong to that tile's owner
            bool mine = j < tn && (!inherits || inherits_long || s_key[j + 1] != s_key[0]);
            if (mine) {
                // never-updated keys (padding_idx rows, the dead entries of a live list) have no
                // record worth reading -- a dead entry's payload is not even a valid reference
                const uint32_t item = s_key[j + 1] & a.imask;
                mine = item != a.pad_item && item != a.pad_item2 && (!SKIP1 || s_multi[j]);
            }
            if (mine && GSPF) {
                g[it] = s_g[j];
                const uint32_t src = s_src[j];
                const float *urow = ((src >> 31) ? a.P0alt : a.P[0]) + (size_t)(src & 0x7fffffffu) * D;
                const slk_vec<VEC> u = rows_on ? slk_vload_if_nt<VEC>(urow + d0, (SLK_NT_OF(a) & 32) != 0) : slk_vzero<VEC>();
#pragma unroll
                for (int i = 0; i < VEC; ++i) c[it].v[i] = g[it] * u.v[i];
            } else if (mine) {
                slk_item_contrib<VEC, MODE>(a, s_pay[j], D, d0, rows_on, c[it], g[it]);
            }
        }
        if (KEYPF && tile + gridDim.x < ntiles) tile_fetch(tile + gridDim.x);  // in flight behind the gathers
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int j = grp + it * GPB;
            if (j < tn) {
                slk_vstore<VEC>(s_row + j * DL + d0, c[it]);
                if (lane == 0) {
                    s_g[j] = g[it];
                    s_live[j] = SLK_ITEM_IS_SNAP(MODE) ? (g[it] != 0.0f) : 1;
                }
            }
        }
        __syncthreads();

        // (3) head r -> group r mod GPB: the in-tile part of its run, summed in occurrence order
        auto finish = [&](int r, bool pre, slk_vec<VEC> &p, slk_vec<VEC> &s, float bp, float bs) {
            const int j = (int)s_head[r];
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
            const bool starts = !(j == 0 && inherits_long);
            const bool ends = !(k == tn && hands_over_long);
            if (starts && ends) {
                if (k == tn && (LONG ? hands_over : has_next)) {
                    // A SHORT run that spills into the next tile: the group reads G keys + payloads at once (the keys
                    // are sorted, so the lanes that still match are a prefix), then takes the occurrences
                    // SLK_SPILL_BATCH at a time: independent loads in flight, summed in occurrence order.
                    uint32_t q = tb + tn;
                    bool more = true;
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
                                if (j0 + e < cnt && (!SLK_ITEM_IS_SNAP(MODE) || gq[e] != 0.0f)) {
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
                slk_item_apply<VEC, G, UPD, PART>(a, item, D, d0, lane, rows_on, nt_rows, any, gv, gb, pre, p, s, bp, bs);
                return;
            }
            // a partial: slot 0 = the inherited run's part, slot 1 = the part of the run the next tile continues (a run
            // that fills the whole tile is both: slot 0)
            const size_t slot = 2 * (size_t)tile + (starts ? 1u : 0u);
            float *pp = a.ipart + slot * (size_t)a.IPS;
            if (rows_on) slk_vstore<VEC>(pp + d0, gv);
            if (lane == 0) {
                pp[a.IPS - 1] = gb;
                a.ipart_meta[2 * slot] = key;
                a.ipart_meta[2 * slot + 1] = (a.ipart_gen << SLK_IPART_GEN_SHIFT) | (starts ? (uint32_t)SLK_IPART_STARTS : 0u) |
                                             (ends ? (uint32_t)SLK_IPART_ENDS : 0u) | (any ? (uint32_t)SLK_IPART_ANY : 0u);
                if (starts) atomicAdd(a.ipart_count, 1u);
            }
        };
#pragma unroll
        for (int h = 0; h < NPRE; ++h) {
            const int r = grp + h * GPB;
            if (r < nheads) finish(r, completes(r), pv[h], sv[h], pb[h], sb[h]);
        }
        // (issuing the row + state loads of the group's further heads before its first run is summed -- the registers of the
        // record gather are free by then -- was measured twice and does not pay: one more head 0.315 vs 0.313 ms at C2; all of
        // them (up to three more: +30 VGPRs) 0.443 vs 0.291 ms at C2 and 0.70 vs 0.56-0.66 ms at the C5 shard, profiles/r03_*)
        for (int r = grp + NPRE * GPB; r < nheads; r += GPB) {
            slk_vec<VEC> p = slk_vzero<VEC>(), s = slk_vzero<VEC>();
            finish(r, false, p, s, 0.0f, 0.0f);
        }
    }
}

// Behind k_item_pass: one row group per tile.  The group of the tile in which a tile-crossing run STARTS adds the run's
// partials in tile order -- its own tile's slot 1, then slot 0 of the following tiles up to the one in which the run
// ends -- and applies the update.  The metas of the next G tiles are read at once (the lanes that still belong to the
// run are a prefix) and their partials are loaded G at a time: a run that fills a thousand tiles costs its owner a few
// dozen round trips.
template <int VEC, int G, int UPD, int PART>
__global__ __launch_bounds__(256) void k_item_stitch(slk_pass_args a) {
    constexpr int GPB = 256 / G;
    constexpr int T = 4 * GPB;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const bool rows_on = on && PART != SLK_PART_BIAS;
    const bool nt_rows = (SLK_NT_OF(a) & 2) != 0;
    const uint32_t ntiles = (a.iend - a.ibegin + T - 1) / T;
    if (*a.ipart_count == 0u) return;  // no long run in this minibatch (the usual case)
    const uint32_t gen = a.ipart_gen;
    for (uint32_t tile = blockIdx.x * GPB + grp; tile < ntiles; tile += gridDim.x * GPB) {
        const size_t slot = 2 * (size_t)tile + 1;
        const uint32_t fl = a.ipart_meta[2 * slot + 1];
        if ((fl >> SLK_IPART_GEN_SHIFT) != gen) continue;  // no long run leaves this tile
        const uint32_t key = a.ipart_meta[2 * slot];
        const float *pp = a.ipart + slot * (size_t)a.IPS;
        slk_vec<VEC> gv = rows_on ? slk_vload<VEC>(pp + d0) : slk_vzero<VEC>();
        float gb = pp[a.IPS - 1];
        bool any = (fl & SLK_IPART_ANY) != 0;
        uint32_t t2 = tile + 1;
        bool more = true;
        // A round looks at the next TPR = R * G tiles: lane l reads the metas of tiles t2 + l + j * G ({key, flags}: one 8-B load;
        // the tiles that continue this run are a prefix), and the partials the round can consume are fetched WITH the metas --
        // their addresses do not depend on them -- so a run that fills thousands of tiles costs its owner one round trip per TPR
        // tiles (measured before: one per tile, then three per G tiles; profiles/r03_*).  What the run does not reach is
        // dropped.  The sums are added in tile order, as the persistent kernel's walk adds them.
        constexpr int R = G <= 16 ? 2 : 1;
        constexpr int TPR = R * G;
        constexpr bool SPEC = G <= 16;  // wider groups: partials fetched after the metas, SLK_STITCH_BATCH at a time
        while (more) {
            // every load of the round is issued before anything waits: the partials first, then the metas (the compiler keeps
            // program order, and a compare right behind a meta load would put a full round trip in front of the partial loads)
            slk_vec<VEC> sc[SPEC ? TPR : 1];
            float sb[SPEC ? TPR : 1];
            if (SPEC) {
#pragma unroll
                for (int e = 0; e < (SPEC ? TPR : 1); ++e) {
                    sc[e] = slk_vzero<VEC>();
                    sb[e] = 0.0f;
                    if (t2 + (uint32_t)e < ntiles) {
                        const float *q = a.ipart + 2 * (size_t)(t2 + (uint32_t)e) * (size_t)a.IPS;
                        if (rows_on) sc[e] = slk_vload<VEC>(q + d0);
                        sb[e] = q[a.IPS - 1];
                    }
                }
            }
            uint2 kf[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const uint32_t tl = t2 + (uint32_t)(lane + j * G);
                kf[j] = make_uint2(0u, 0u);
                if (tl < ntiles) kf[j] = *reinterpret_cast<const uint2 *>(a.ipart_meta + 4 * (size_t)tl);
            }
            uint32_t f[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                f[j] = kf[j].y;
                if (t2 + (uint32_t)(lane + j * G) >= ntiles || (f[j] >> SLK_IPART_GEN_SHIFT) != gen || (f[j] & SLK_IPART_STARTS) ||
                    kf[j].x != key)
                    f[j] = 0u;
                else
                    f[j] |= 1u;  // bit 0: part of this run
            }
            // cnt = tiles of this round that belong to the run: up to and including the first one that ends it
            // (the lanes' flags as bit masks of the group, OR-reduced in log2(G) exchanges: a loop of G dependent
            // lane-to-lane reads per round was what the stitch of a run over thousands of tiles spent its time in)
            int cnt = 0;
            bool ended = false;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (!ended && cnt == j * G) {
                    const unsigned long long valid = slk_group_or<G>((f[j] & 1u) ? 1ull << lane : 0ull);
                    const unsigned long long ends = slk_group_or<G>((f[j] & SLK_IPART_ENDS) ? 1ull << lane : 0ull);
                    const unsigned long long full = G == 64 ? ~0ull : (1ull << G) - 1ull;
                    int c = (valid & full) == full ? G : __builtin_ctzll(~valid);      // the leading lanes that continue the run
                    const unsigned long long e_in = ends & (c == 64 ? ~0ull : (1ull << c) - 1ull);
                    if (e_in) {
                        c = __builtin_ctzll(e_in) + 1;                                 // ... up to and including the one that ends it
                        ended = true;
                    }
                    const unsigned long long anym = slk_group_or<G>((f[j] & SLK_IPART_ANY) ? 1ull << lane : 0ull);
                    any = any || (anym & (c == 64 ? ~0ull : (1ull << c) - 1ull)) != 0ull;
                    cnt += c;
                }
            }
            if (SPEC) {
#pragma unroll
                for (int e = 0; e < (SPEC ? TPR : 1); ++e) {
                    if (e < cnt) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) gv.v[i] += sc[e].v[i];
                        gb += sb[e];
                    }
                }
            }
            for (int l0 = 0; !SPEC && l0 < cnt; l0 += SLK_STITCH_BATCH) {
                slk_vec<VEC> cc[SLK_STITCH_BATCH];
                float cb[SLK_STITCH_BATCH];
#pragma unroll
                for (int e = 0; e < SLK_STITCH_BATCH; ++e) {
                    cc[e] = slk_vzero<VEC>();
                    cb[e] = 0.0f;
                    if (l0 + e < cnt) {
                        const float *q = a.ipart + 2 * (size_t)(t2 + (uint32_t)(l0 + e)) * (size_t)a.IPS;
                        if (rows_on) cc[e] = slk_vload<VEC>(q + d0);
                        cb[e] = q[a.IPS - 1];
                    }
                }
#pragma unroll
                for (int e = 0; e < SLK_STITCH_BATCH; ++e) {
                    if (l0 + e < cnt) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) gv.v[i] += cc[e].v[i];
                        gb += cb[e];
                    }
                }
            }
            more = !ended && cnt == TPR;
            t2 += (uint32_t)TPR;
        }
        slk_vec<VEC> p = slk_vzero<VEC>(), s = slk_vzero<VEC>();
        slk_item_apply<VEC, G, UPD, PART>(a, key & a.imask, D, d0, lane, rows_on, nt_rows, any, gv, gb, false, p, s, 0.0f, 0.0f);
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

// Workgroups of 256 threads of kernel `fn` a CU holds resident (by its registers and static LDS), cached per ctx; 1 << 20 when
// the runtime cannot tell.  A grid-stride pass launched with more workgroups per CU than that runs the surplus as a second,
// partial round (round 5: the SparseAdam user pass -- 72 VGPRs, 7 resident workgroups -- launched 8 per CU: 0.58 ms; capped:
// 0.44, profiles/r05_w_*, r05_x_*).
template <class Fn>
static inline int slk_occupancy_of(slk_ctx *ctx, Fn fn) {
    const void *key = reinterpret_cast<const void *>(fn);
    if (!key) return 1 << 20;
    for (const auto &e : ctx->occ_cache)
        if (e.first == key) return e.second;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1 << 20;
    }
    ctx->occ_cache.push_back({key, per_cu});
    return per_cu;
}

// grid of a grid-stride row pass: at most min("user_grid_mult", the kernel's own occupancy) workgroups per CU
template <class Fn>
static inline unsigned slk_grid_for_fn(slk_ctx *ctx, Fn fn, size_t work_items, unsigned per_block) {
    const int occ = slk_occupancy_of(ctx, fn);
    return slk_grid_for(ctx, work_items, per_block, ctx->opt_user_grid_mult < occ ? ctx->opt_user_grid_mult : occ);
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

// flags[mb] |= 1 iff some run of minibatch mb's window [mb * per_mb, min((mb + 1) * per_mb, n)) of the sorted occurrence
// list wholly covers one of the item pass's tiles (T positions, aligned to the window): first key == last key of the tile,
// for an item that is updated at all.  Ids only: computed once per chunk, read back by the host.
static __global__ __launch_bounds__(256) void k_item_long_flags(const uint32_t *ikey, uint32_t n, uint32_t per_mb, uint32_t T,
                                                         uint32_t imask, uint32_t pad_item, uint32_t pad_item2, int *flags) {
    const uint32_t tiles_per_mb = (per_mb + T - 1) / T;
    const uint32_t n_mb = (n + per_mb - 1) / per_mb;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_mb * tiles_per_mb; i += gridDim.x * 256) {
        const uint32_t mb = i / tiles_per_mb, q = i - mb * tiles_per_mb;
        const uint32_t w0 = mb * per_mb, w1 = (n - w0 < per_mb) ? n : w0 + per_mb;
        const uint32_t t0 = w0 + q * T;
        if (t0 >= w1) continue;
        if (w1 - t0 < T) continue;  // only full tiles make a run long
        const uint32_t t1 = t0 + T;
        const uint32_t k0 = ikey[t0], k1 = ikey[t1 - 1];
        const uint32_t item = k0 & imask;
        if (k0 == k1 && item != pad_item && item != pad_item2) flags[mb] = 1;
    }
}

// msorted[e] = 1 iff occurrence e of the item-sorted list shares its key with a neighbour inside its minibatch's window (the
// item occurs more than once in the minibatch), and for those mflag[payload] = 1 (mflag zeroed by the caller: on a catalogue far
// larger than a minibatch nearly every occurrence is the only one of its item and nothing is scattered for it).  Ids only: once
// per chunk, with the sorts.
static __global__ __launch_bounds__(256) void k_item_multi_flags(const uint32_t *ikey, const uint32_t *ipay, uint32_t n, uint32_t per_mb,
                                                          uint8_t *msorted, uint8_t *mflag) {
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const uint32_t w0 = e / per_mb * per_mb, w1 = (n - w0 < per_mb) ? n : w0 + per_mb;
        const uint32_t key = ikey[e];
        const bool multi = (e > w0 && ikey[e - 1] == key) || (e + 1 < w1 && ikey[e + 1] == key);
        msorted[e] = multi ? 1 : 0;
        if (multi) mflag[ipay[e]] = 1;
    }
}

// ---------------------------------------------------------------------------------------
// adaptive hinge: the per-column selection of _get_multiple_negative_predictions' view(n, B) layout
// (factorization/implicit.py:266-275) over the scores of every (interaction, pair).  A template only so that the one
// definition serves slk_bilinear.hip and slk_shard.hip (no relocatable device code in this build).
// ---------------------------------------------------------------------------------------
// one thread per column c of the [n, B] candidate matrix; k0 = chunk-local index of the
// minibatch's first interaction.  every entry of gk that belongs to this minibatch is written here (no memset before the launch).
//
// qk / live (optional): the two occurrences of column c that carry a gradient -- the positive of
// interaction k0 + c and the selected negative -- as item-pass payloads r = position * NP + pair
// (qk: chunk-local interaction -> user-sorted position), or ~0u twice when the hinge is inactive.
template <int UNUSED>
__global__ __launch_bounds__(256) void k_adaptive_select(const float *sk, float *gk, uint32_t k0, uint32_t bm,
                                                         int nn, float inv_b, double *loss_partial,
                                                         const uint32_t *qk, uint32_t *live) {
    __shared__ double red[256];
    const int NP = nn + 1;
    double lsum = 0.0;
    for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < bm; c += gridDim.x * 256) {
        const float sp = sk[(size_t)(k0 + c) * NP];
        float best = 0.0f;
        size_t best_at = 0;
        for (int r = 0; r < nn; ++r) {
            const uint32_t f = (uint32_t)r * bm + c;  // flat index into the n*B draws
            const size_t at = (size_t)(k0 + f / (uint32_t)nn) * NP + 1 + (f % (uint32_t)nn);
            const float sc = sk[at];
            if (r == 0 || sc > best) {  // torch.max(dim=0): first maximum wins ties
                best = sc;
                best_at = at;
            }
        }
        const float x = best - sp + 1.0f;
        lsum += (double)(x > 0.0f ? x : 0.0f);
        const float g = x >= 0.0f ? inv_b : 0.0f;
        gk[(size_t)(k0 + c) * NP] = -g;
        // the column's n draws are written by this thread alone, and the columns partition the n*B draws: every entry of
        // gk gets its value here (no memset before the launch)
        for (int r = 0; r < nn; ++r) {
            const uint32_t f = (uint32_t)r * bm + c;
            const size_t at = (size_t)(k0 + f / (uint32_t)nn) * NP + 1 + (f % (uint32_t)nn);
            gk[at] = at == best_at ? g : 0.0f;
        }
        if (live) {
            const uint32_t kb = (uint32_t)(best_at / (size_t)NP), sb = (uint32_t)(best_at - (size_t)kb * NP);
            live[2 * (size_t)c] = g != 0.0f ? qk[k0 + c] * (uint32_t)NP : 0xffffffffu;
            live[2 * (size_t)c + 1] = g != 0.0f ? qk[kb] * (uint32_t)NP + sb : 0xffffffffu;
        }
    }
    const double tot = slk_block_sum_256(lsum, red);
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = tot;
}


// the item pass and the stitch kernel that goes behind it (slk_launch_item_pass)
struct slk_item_fns {
    slk_pass_fn pass, stitch, pass_short;  // pass_short: k_item_pass<..., LONG = false>
    slk_pass_fn pass_lat;                  // ... with every head's row loaded early (launches of few tiles)
};

template <int VEC, int G, int MODE, int PART = SLK_PART_BOTH>
static slk_item_fns slk_item_pass_fn(int upd) {
    if (upd == SLK_UPD_ADAGRAD)
        return {k_item_pass<VEC, G, SLK_UPD_ADAGRAD, MODE, PART>, k_item_stitch<VEC, G, SLK_UPD_ADAGRAD, PART>,
                k_item_pass<VEC, G, SLK_UPD_ADAGRAD, MODE, PART, false>,
                k_item_pass<VEC, G, SLK_UPD_ADAGRAD, MODE, PART, false, 4>};
    if (upd == SLK_UPD_SPARSE_ADAM)
        return {k_item_pass<VEC, G, SLK_UPD_SPARSE_ADAM, MODE, PART>, k_item_stitch<VEC, G, SLK_UPD_SPARSE_ADAM, PART>,
                k_item_pass<VEC, G, SLK_UPD_SPARSE_ADAM, MODE, PART, false>,
                k_item_pass<VEC, G, SLK_UPD_SPARSE_ADAM, MODE, PART, false, 4>};
    if (upd == SLK_UPD_SGD)
        return {k_item_pass<VEC, G, SLK_UPD_SGD, MODE, PART>, k_item_stitch<VEC, G, SLK_UPD_SGD, PART>,
                k_item_pass<VEC, G, SLK_UPD_SGD, MODE, PART, false>,
                k_item_pass<VEC, G, SLK_UPD_SGD, MODE, PART, false, 4>};
    return {k_item_pass<VEC, G, SLK_UPD_GRAD_ONLY, MODE, PART>, k_item_stitch<VEC, G, SLK_UPD_GRAD_ONLY, PART>,
            k_item_pass<VEC, G, SLK_UPD_GRAD_ONLY, MODE, PART, false>,
                k_item_pass<VEC, G, SLK_UPD_GRAD_ONLY, MODE, PART, false, 4>};
}

// Launches the item pass over a.ibegin .. a.iend and the stitch kernel behind it (partials in ctx scratch).
// may_have_long = false: the caller KNOWS (k_item_long_flags, read back once per chunk) that no run of this window wholly
// covers a tile -- the plain pass alone.
// lat_div: the every-head-early form (pass_lat) is taken up to opt_item_lat_max_tiles / lat_div tiles (PoolNet's records are twice
// as long and its form stops paying earlier: measured, profiles/r03_y_*)
static inline int slk_launch_item_pass(slk_ctx *ctx, const slk_item_fns &fns, slk_pass_args &a, int g, hipStream_t s,
                                       const char *what, bool may_have_long = true, int lat_div = 1) {
    const unsigned gpb = 256u / (unsigned)g, T = 4u * gpb;
    const size_t n = (size_t)(a.iend - a.ibegin);
    if (n == 0) return SLK_OK;
    const size_t ntiles = (n + T - 1) / T;
    if (!may_have_long) {
        a.ipart = nullptr;
        a.ipart_meta = nullptr;
        a.ipart_count = nullptr;
        const bool lat = fns.pass_lat && (int64_t)ntiles * lat_div <= ctx->opt_item_lat_max_tiles;
        hipLaunchKernelGGL(lat ? fns.pass_lat : fns.pass_short, dim3(slk_grid_for(ctx, n, T, ctx->opt_item_grid_mult)), dim3(256), 0, s, a);
        SLK_LAUNCH_CHECK(ctx, what);
        return SLK_OK;
    }
    const int ips = (a.D + 3) / 4 * 4 + 4;
    int rc;
    if ((rc = slk_ensure(ctx, ctx->ipart, 2 * ntiles * (size_t)ips * 4))) return rc;
    // meta: [2 * ntiles][2] words, then the counter of long runs; flags carry the launch's stamp, so nothing is cleared
    // between launches except the counter (a fresh or regrown buffer is zeroed once: stamp 0 is never used)
    const size_t meta_bytes = 2 * ntiles * 8 + 64;
    if (meta_bytes > ctx->ipart_meta.cap) {
        if ((rc = slk_ensure(ctx, ctx->ipart_meta, meta_bytes))) return rc;
        SLK_HIP(ctx, hipMemsetAsync(ctx->ipart_meta.p, 0, ctx->ipart_meta.cap, s));
    }
    if (ctx->ipart_gen >= 0x0ffffffeu) {  // the 28-bit stamp wraps: forget every old partial
        SLK_HIP(ctx, hipMemsetAsync(ctx->ipart_meta.p, 0, ctx->ipart_meta.cap, s));
        ctx->ipart_gen = 0u;
    }
    ++ctx->ipart_gen;
    a.ipart = (float *)ctx->ipart.p;
    a.ipart_meta = (uint32_t *)ctx->ipart_meta.p;
    a.ipart_count = a.ipart_meta + 4 * ntiles;
    a.ipart_gen = ctx->ipart_gen;
    a.IPS = ips;
    SLK_HIP(ctx, hipMemsetAsync(a.ipart_count, 0, 4, s));
    hipLaunchKernelGGL(fns.pass, dim3(slk_grid_for(ctx, n, T, ctx->opt_item_grid_mult)), dim3(256), 0, s, a);
    SLK_LAUNCH_CHECK(ctx, what);
    ++ctx->stat_item_long;
    hipLaunchKernelGGL(fns.stitch, dim3(slk_grid_for(ctx, ntiles, gpb)), dim3(256), 0, s, a);
    SLK_LAUNCH_CHECK(ctx, "k_item_stitch");
    return SLK_OK;
}

// Row-update mode of the fused passes for an optimizer kind.
static inline int slk_upd_for(int opt_kind) {
    if (opt_kind == SLK_OPT_ADAGRAD) return SLK_UPD_ADAGRAD;
    if (opt_kind == SLK_OPT_SPARSE_ADAM) return SLK_UPD_SPARSE_ADAM;
    if (opt_kind == SLK_OPT_SGD) return SLK_UPD_SGD;
    return SLK_UPD_GRAD_ONLY;
}

// Per-step optimizer coefficients, formed in double and rounded to fp32 as torch does.
static inline void slk_set_opt_coeffs(slk_pass_args &a, const slk_optim *optim) {
    const double step = (double)(optim->step + 1);
    a.c_eps = (float)optim->eps;
    if (optim->kind == SLK_OPT_SGD) {
        a.c_lr = (float)optim->lr;
    } else if (optim->kind == SLK_OPT_ADAGRAD) {
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
