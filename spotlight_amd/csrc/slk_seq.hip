// slk_seq.hip -- PoolNet / ImplicitSequenceModel training and prediction for gfx950.
//
// Replaces the minibatch body of ImplicitSequenceModel.fit() with representation='pooling'
// (spotlight/sequence/implicit.py:225-255): PoolNet.user_representation (running, padding-aware
// average of the item embeddings seen so far, sequence/representations.py:76-114),
// PoolNet.forward for the positive targets (the sequence itself) and the sampled negatives
// (:116-144), the masked loss (losses.py, mask = sequence != 0), autograd's backward (the
// cumsum's reverse scan + embedding backward with duplicate rows summed, padding_idx rows
// skipped) and the optimizer step.
//
//   SEQUENCE PASS  one workgroup per sequence; the L item rows are staged once in LDS; row
//        groups (G = dim/4 lanes) each own a chunk of consecutive timesteps and the running
//        sums are stitched with a block-level scan (forward: prefix sum + non-zero count ->
//        representation -> scores -> loss -> dL/dscore; backward: suffix sum of
//        dL/d(prefix sum)).  Per timestep it writes one record
//             [ representation (D) | the own item's contribution g_pos * representation + history gradient (D) ]
//        (+ dL/dscore of the 1+n pairs in a side array) -- everything the item rows' owners need.  Round 5: the second half
//        used to be the history gradient alone and the own item's occurrence read BOTH halves (768 B of records read per
//        timestep; now each occurrence reads one half: 512 B) -- same products, same additions, same bits.  The backward
//        scan forms it from what the forward scan left in the record (its own stores: L2 hits), so the forward scan -- at its
//        register limit -- is untouched.
//   ITEM PASS (slk_kernels.h, SEQ mode)  occurrences (timestep, pair) sorted by item; one
//        owner group per unique item sums  the ready-made contribution (the sequence's own item)
//        or g * representation (a sampled item)  and applies the optimizer once.
#include <math.h>

#include "slk_kernels.h"

enum { SQ_MCOUNT = 12, SQ_REP, SQ_BIK0, SQ_BIK1, SQ_BIP0, SQ_BIP1, SQ_GSN = 23, SQ_MCOUNT_B = 30 };  // ctx->extra slots (24, 25: slk_eval.hip; 0..10 belong to slk_shard.hip, 16..23 and 38 to slk_bilinear.hip)

struct slk_seq_args {
    const float *E;         // item_embeddings
    const float *bias;      // item_biases
    int D, L, NP;           // NP = 1 + candidates per timestep
    const int64_t *seqs;    // chunk base, [n_seq][L]
    const uint32_t *neg32;  // this minibatch's draws (sampling order of sequence/implicit.py:266-286)
    uint32_t s_begin, s_end;  // this minibatch's sequences (chunk-local)
    float *rec;             // records of this minibatch: index (s - s_begin) * L + t
    int RS;
    float *gsn;             // dL/dscore per (record, pair): [record * NP + pair]
    const uint32_t *mcount;  // mask.sum() of this minibatch
    double *loss_partial;
    int loss_kind;
    int C;                  // timesteps per row-group chunk
    slk_bloom_dev ib;       // item_embedding_layer = BloomEmbedding (n_hash == 0: plain table)
};

// item vector of `id` in the sequence passes: BLOOM is a compile-time property there, so that the
// plain path keeps straight-line loads the compiler can batch (a runtime n_hash test cost the
// register-resident pass its memory-level parallelism: 0.20 -> 0.40 ms)
template <int VEC, bool BLOOM>
__device__ __forceinline__ slk_vec<VEC> slk_seq_vec(const slk_seq_args &a, uint32_t id, int D, int d0, bool on) {
    if (BLOOM) return slk_emb_vec<VEC>(a.E, a.ib, id, D, d0, on);
    return on ? slk_vload<VEC>(a.E + (size_t)id * D + d0) : slk_vzero<VEC>();
}

// non-zero sequence entries per minibatch (mask.sum(), losses.py:45-48)
__global__ __launch_bounds__(256) void k_seq_count(const int64_t *seqs, uint32_t n_seq, uint32_t L, uint32_t bsz,
                                                   uint32_t *mcount) {
    __shared__ unsigned red[256];
    const uint32_t mb = blockIdx.y;
    const uint32_t s0 = mb * bsz, s1 = (n_seq - s0 < bsz) ? n_seq : s0 + bsz;
    const size_t lo = (size_t)s0 * L, hi = (size_t)s1 * L;
    unsigned c = 0;
    for (size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (size_t)gridDim.x * 256)
        c += seqs[i] != 0;
    red[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0]) atomicAdd(&mcount[mb], red[0]);
}

template <int VEC, int G, bool ADAPT, bool BLOOM>
__global__ __launch_bounds__(256) void k_seq_pass(slk_seq_args a) {
    HIP_DYNAMIC_SHARED(float, lds)
    __shared__ double red[256];
    constexpr int NG = 256 / G;
    constexpr int DL = G * VEC;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D, L = a.L;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    float *sE = lds;                   // [L][DL]  item rows, later dL/d(prefix sum)
    float *sT = lds + (size_t)L * DL;  // [NG][DL] per-chunk sums
    float *sC = sT + NG * DL;          // [NG][DL] per-chunk non-zero counts
    const float M = (float)*a.mcount;
    const uint32_t Bs = a.s_end - a.s_begin;
    const int nn = a.NP - 1;
    const int t0 = grp * a.C < L ? grp * a.C : L;
    const int t1 = t0 + a.C < L ? t0 + a.C : L;
    double loss_acc = 0.0;

    for (uint32_t s = a.s_begin + blockIdx.x; s < a.s_end; s += gridDim.x) {
        const int64_t *seq = a.seqs + (size_t)s * L;
        const uint32_t bl = s - a.s_begin;
        float *recs = a.rec + (size_t)bl * L * a.RS;
        __syncthreads();  // LDS of the previous sequence no longer in use
        // ---- (A) stage the sequence's item rows
        for (int t = grp; t < L; t += NG) {
            const slk_vec<VEC> e = slk_seq_vec<VEC, BLOOM>(a, (uint32_t)seq[t], D, d0, on);
            slk_vstore<VEC>(sE + t * DL + d0, e);
        }
        __syncthreads();
        // ---- (B1) per-chunk sum and non-zero count
        {
            slk_vec<VEC> sum = slk_vzero<VEC>(), cnt = slk_vzero<VEC>();
            for (int t = t0; t < t1; ++t) {
                const slk_vec<VEC> e = slk_vload<VEC>(sE + t * DL + d0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    sum.v[i] += e.v[i];
                    cnt.v[i] += (e.v[i] != 0.0f) ? 1.0f : 0.0f;
                }
            }
            slk_vstore<VEC>(sT + grp * DL + d0, sum);
            slk_vstore<VEC>(sC + grp * DL + d0, cnt);
        }
        __syncthreads();
        // ---- (B2) exclusive prefix -> representation -> scores -> loss -> dL/d(prefix sum)
        {
            slk_vec<VEC> S = slk_vzero<VEC>(), Cn = slk_vzero<VEC>();
            for (int gq = 0; gq < grp; ++gq) {
                const slk_vec<VEC> x = slk_vload<VEC>(sT + gq * DL + d0), y = slk_vload<VEC>(sC + gq * DL + d0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    S.v[i] += x.v[i];
                    Cn.v[i] += y.v[i];
                }
            }
            for (int tb = t0; tb < t1; tb += 4) {
                // independent loads of up to 4 timesteps first (memory-level parallelism)
                uint32_t it[4], nid[4];
                slk_vec<VEC> nrow[4];
                float pb[4], nb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = tb + k;
                    it[k] = nid[k] = 0;
                    pb[k] = nb[k] = 0.0f;
                    nrow[k] = slk_vzero<VEC>();
                    if (t < t1) {
                        it[k] = (uint32_t)seq[t];
                        pb[k] = a.bias[it[k]];
                        if (!ADAPT) {
                            nid[k] = a.neg32[(size_t)bl * L + t];
                            nrow[k] = slk_seq_vec<VEC, BLOOM>(a, nid[k], D, d0, on);
                            nb[k] = a.bias[nid[k]];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = tb + k;
                    if (t >= t1) break;
                    const slk_vec<VEC> e = slk_vload<VEC>(sE + t * DL + d0);
                    slk_vec<VEC> rep, c1;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        c1.v[i] = Cn.v[i] + 1.0f;
                        rep.v[i] = S.v[i] / c1.v[i];
                    }
                    const float sp = pb[k] + slk_group_sum<G>(slk_vdot<VEC>(rep, e));
                    float sn;
                    int chosen = 0;
                    slk_vec<VEC> nr = nrow[k];
                    if (!ADAPT) {
                        sn = nb[k] + slk_group_sum<G>(slk_vdot<VEC>(rep, nr));
                    } else {
                        // losses.py:164-166: the highest-scoring of the n candidates drawn for this
                        // (sequence, timestep); row (r*B + b) of the (n*B, L) draw; first maximum wins
                        sn = 0.0f;
                        for (int rb = 0; rb < nn; rb += 4) {
                            slk_vec<VEC> cr[4];
                            float cb[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                cr[j] = slk_vzero<VEC>();
                                cb[j] = 0.0f;
                                if (rb + j < nn) {
                                    const uint32_t cid = a.neg32[((size_t)(rb + j) * Bs + bl) * L + t];
                                    cr[j] = slk_seq_vec<VEC, BLOOM>(a, cid, D, d0, on);
                                    cb[j] = a.bias[cid];
                                }
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (rb + j < nn) {
                                    const float sc = cb[j] + slk_group_sum<G>(slk_vdot<VEC>(rep, cr[j]));
                                    if (rb + j == 0 || sc > sn) {
                                        sn = sc;
                                        chosen = rb + j;
                                        nr = cr[j];
                                    }
                                }
                            }
                        }
                    }
                    float l, gp, gn;
                    slk_pair_loss(a.loss_kind, sp, sn, 1.0f, l, gp, gn);
                    const float mask = it[k] != 0u ? 1.0f : 0.0f;
                    const float w = mask / M;  // d(sum(loss*mask)/sum(mask)) / d loss
                    gp = gp * w;
                    gn = gn * w;
                    float *rec = recs + (size_t)t * a.RS;
                    if (on) slk_vstore<VEC>(rec + d0, rep);
                    if (lane == 0) {
                        loss_acc += (double)(l * mask);
                        // 32-bit index off the uniform base: no 64-bit address pair held in VGPRs
                        const uint32_t gi = (bl * (uint32_t)L + (uint32_t)t) * (uint32_t)a.NP;
                        a.gsn[gi] = gp;
                        if (!ADAPT) {
                            a.gsn[gi + 1] = gn;
                        } else {
                            for (int j = 0; j < nn; ++j) a.gsn[gi + 1 + j] = (j == chosen) ? gn : 0.0f;
                        }
                    }
                    // advance the running sums, then overwrite the staged row by dL/d(prefix sum)
                    slk_vec<VEC> gs;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        gs.v[i] = (gp * e.v[i] + gn * nr.v[i]) / c1.v[i];
                        S.v[i] += e.v[i];
                        Cn.v[i] += (e.v[i] != 0.0f) ? 1.0f : 0.0f;
                    }
                    slk_vstore<VEC>(sE + t * DL + d0, gs);
                }
            }
        }
        __syncthreads();  // every group has consumed the chunk sums
        // ---- (C) cumsum backward: row j receives the sum of dL/d(prefix sum) over t > j
        {
            slk_vec<VEC> sum = slk_vzero<VEC>();
            for (int t = t0; t < t1; ++t) {
                const slk_vec<VEC> x = slk_vload<VEC>(sE + t * DL + d0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) sum.v[i] += x.v[i];
            }
            slk_vstore<VEC>(sT + grp * DL + d0, sum);
        }
        __syncthreads();
        {
            slk_vec<VEC> suf = slk_vzero<VEC>();
            for (int gq = NG - 1; gq > grp; --gq) {
                const slk_vec<VEC> x = slk_vload<VEC>(sT + gq * DL + d0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) suf.v[i] += x.v[i];
            }
            for (int t = t1 - 1; t >= t0; --t) {
                if (on) {  // the own item's ready-made contribution (see k_seq_pass_reg)
                    float *rec = recs + (size_t)t * a.RS;
                    slk_vec<VEC> c0 = slk_vload<VEC>(rec + d0);
                    const float gpt = a.gsn[(bl * (uint32_t)L + (uint32_t)t) * (uint32_t)a.NP];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) c0.v[i] = gpt * c0.v[i] + suf.v[i];
                    slk_vstore<VEC>(rec + D + d0, c0);
                }
                const slk_vec<VEC> x = slk_vload<VEC>(sE + t * DL + d0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) suf.v[i] += x.v[i];
            }
        }
    }
    const double tot = slk_block_sum_256(loss_acc, red);
    if (threadIdx.x == 0) a.loss_partial[blockIdx.x] = tot / (double)M;
}

// SEQUENCE PASS, register-resident variant: used when a row group's chunk of timesteps fits CMAX
// rows of registers (L <= 256 and ceil(L / NG) <= 16: e.g. L = 200 at dim <= 64).  Same chunks,
// same scans and the same summation order as k_seq_pass -- results are bit-identical -- but a
// group keeps its chunk's item rows in VGPRs instead of staging the whole sequence in LDS:
//   * LDS per workgroup drops from L*D*4 + 8 KB (59 KB at L=200, D=64: 2 workgroups per CU) to
//     the 8 KB of chunk sums, so occupancy is set by registers (4 workgroups per CU);
//   * the ids of the chunk are fetched by one coalesced load per group (lane k holds timestep
//     t0 + k, broadcast by __shfl) and all of the chunk's row loads are issued back to back
//     instead of one dependent (id -> row) pair per loop iteration.
template <int VEC, int G, bool ADAPT, bool BLOOM, int CMAX>
__global__ __launch_bounds__(256) SLK_WAVES_PER_EU(2) void k_seq_pass_reg(slk_seq_args a) {
    constexpr int NG = 256 / G;
    constexpr int DL = G * VEC;
    __shared__ double red[256];
    __shared__ __attribute__((aligned(16))) float sT[NG * DL];  // per-chunk sums
    __shared__ __attribute__((aligned(16))) float sC[NG * DL];  // per-chunk non-zero counts
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int D = a.D, L = a.L;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const float M = (float)*a.mcount;
    const uint32_t Bs = a.s_end - a.s_begin;
    const int nn = a.NP - 1;
    const int t0 = grp * a.C < L ? grp * a.C : L;
    const int t1 = t0 + a.C < L ? t0 + a.C : L;
    const int cnt = t1 - t0;  // <= CMAX <= G
    double loss_acc = 0.0;

    for (uint32_t s = a.s_begin + blockIdx.x; s < a.s_end; s += gridDim.x) {
        const int64_t *seq = a.seqs + (size_t)s * L;
        const uint32_t bl = s - a.s_begin;
        float *recs = a.rec + (size_t)bl * L * a.RS;
        // ---- (A) ids of the chunk: lane k holds timestep t0 + k; rows straight into registers
        uint32_t my_it = 0u, my_neg = 0u;
        if (lane < cnt) {
            my_it = (uint32_t)seq[t0 + lane];
            if (!ADAPT) my_neg = a.neg32[(size_t)bl * L + t0 + lane];
        }
        slk_vec<VEC> e[CMAX];
#pragma unroll
        for (int k = 0; k < CMAX; ++k) {
            const uint32_t id = __shfl(my_it, k, G);
            e[k] = slk_seq_vec<VEC, BLOOM>(a, id, D, d0, on && k < cnt);
        }
        // ---- (B1) per-chunk sum and non-zero count (rows beyond the chunk are zero: exact no-ops)
        {
            slk_vec<VEC> sum = slk_vzero<VEC>(), cn = slk_vzero<VEC>();
#pragma unroll
            for (int k = 0; k < CMAX; ++k) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    sum.v[i] += e[k].v[i];
                    cn.v[i] += (e[k].v[i] != 0.0f) ? 1.0f : 0.0f;
                }
            }
            __syncthreads();  // chunk sums of the previous sequence no longer in use
            slk_vstore<VEC>(sT + grp * DL + d0, sum);
            slk_vstore<VEC>(sC + grp * DL + d0, cn);
        }
        __syncthreads();
        // ---- (B2) exclusive prefix -> representation -> scores -> loss -> dL/d(prefix sum)
        {
            slk_vec<VEC> S = slk_vzero<VEC>(), Cn = slk_vzero<VEC>();
            for (int gq = 0; gq < grp; ++gq) {
                const slk_vec<VEC> x = slk_vload<VEC>(sT + gq * DL + d0), y = slk_vload<VEC>(sC + gq * DL + d0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    S.v[i] += x.v[i];
                    Cn.v[i] += y.v[i];
                }
            }
#pragma unroll
            for (int kb = 0; kb < CMAX; kb += 4) {
                if (kb >= cnt) continue;
                // independent loads of up to 4 timesteps first (memory-level parallelism)
                uint32_t it[4], nid[4];
                slk_vec<VEC> nrow[4];
                float pb[4], nb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    it[k] = __shfl(my_it, (kb + k) & (G - 1), G);
                    nid[k] = __shfl(my_neg, (kb + k) & (G - 1), G);
                    pb[k] = nb[k] = 0.0f;
                    nrow[k] = slk_vzero<VEC>();
                    if (kb + k < cnt) {
                        pb[k] = a.bias[it[k]];
                        if (!ADAPT) {
                            nrow[k] = slk_seq_vec<VEC, BLOOM>(a, nid[k], D, d0, on);
                            nb[k] = a.bias[nid[k]];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (kb + k < CMAX && kb + k < cnt) {
                        const int t = t0 + kb + k;
                        const slk_vec<VEC> ev = e[(kb + k) < CMAX ? (kb + k) : 0];
                        slk_vec<VEC> rep, c1;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            c1.v[i] = Cn.v[i] + 1.0f;
                            rep.v[i] = S.v[i] / c1.v[i];
                        }
                        const float sp = pb[k] + slk_group_sum<G>(slk_vdot<VEC>(rep, ev));
                        float sn;
                        int chosen = 0;
                        slk_vec<VEC> nr = nrow[k];
                        if (!ADAPT) {
                            sn = nb[k] + slk_group_sum<G>(slk_vdot<VEC>(rep, nr));
                        } else {
                            // losses.py:164-166: the highest-scoring of the n candidates drawn for this
                            // (sequence, timestep); row (r*B + b) of the (n*B, L) draw; first maximum wins
                            sn = 0.0f;
                            for (int rb = 0; rb < nn; rb += 4) {
                                slk_vec<VEC> cr[4];
                                float cb[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    cr[j] = slk_vzero<VEC>();
                                    cb[j] = 0.0f;
                                    if (rb + j < nn) {
                                        const uint32_t cid = a.neg32[((size_t)(rb + j) * Bs + bl) * L + t];
                                        cr[j] = slk_seq_vec<VEC, BLOOM>(a, cid, D, d0, on);
                                        cb[j] = a.bias[cid];
                                    }
                                }
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    if (rb + j < nn) {
                                        const float sc = cb[j] + slk_group_sum<G>(slk_vdot<VEC>(rep, cr[j]));
                                        if (rb + j == 0 || sc > sn) {
                                            sn = sc;
                                            chosen = rb + j;
                                            nr = cr[j];
                                        }
                                    }
                                }
                            }
                        }
                        float l, gp, gn;
                        slk_pair_loss(a.loss_kind, sp, sn, 1.0f, l, gp, gn);
                        const float mask = it[k] != 0u ? 1.0f : 0.0f;
                        const float w = mask / M;  // d(sum(loss*mask)/sum(mask)) / d loss
                        gp = gp * w;
                        gn = gn * w;
                        float *rec = recs + (size_t)t * a.RS;
                        if (on) slk_vstore<VEC>(rec + d0, rep);
                        if (lane == 0) {
                            loss_acc += (double)(l * mask);
                            // 32-bit index off the uniform base: no 64-bit address pair held in VGPRs
                            const uint32_t gi = (bl * (uint32_t)L + (uint32_t)t) * (uint32_t)a.NP;
                            a.gsn[gi] = gp;
                            if (!ADAPT) {
                                a.gsn[gi + 1] = gn;
                            } else {
                                for (int j = 0; j < nn; ++j) a.gsn[gi + 1 + j] = (j == chosen) ? gn : 0.0f;
                            }
                        }
                        // advance the running sums; the row's registers now hold dL/d(prefix sum)
                        slk_vec<VEC> gs;
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            gs.v[i] = (gp * ev.v[i] + gn * nr.v[i]) / c1.v[i];
                            S.v[i] += ev.v[i];
                            Cn.v[i] += (ev.v[i] != 0.0f) ? 1.0f : 0.0f;
                        }
                        e[(kb + k) < CMAX ? (kb + k) : 0] = gs;
                    }
                }
            }
        }
        __syncthreads();  // every group has consumed the chunk sums
        // ---- (C) cumsum backward: row j receives the sum of dL/d(prefix sum) over t > j
        {
            slk_vec<VEC> sum = slk_vzero<VEC>();
#pragma unroll
            for (int k = 0; k < CMAX; ++k) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) sum.v[i] += e[k].v[i];
            }
            slk_vstore<VEC>(sT + grp * DL + d0, sum);
        }
        __syncthreads();
        {
            slk_vec<VEC> suf = slk_vzero<VEC>();
            for (int gq = NG - 1; gq > grp; --gq) {
                const slk_vec<VEC> x = slk_vload<VEC>(sT + gq * DL + d0);
#pragma unroll
                for (int i = 0; i < VEC; ++i) suf.v[i] += x.v[i];
            }
            // The own item's contribution, ready-made for its owner: g_pos * representation (what this lane and lane 0 of its
            // row group stored in (B2): the group's own stores, read back behind the barriers above -- L2 hits) + the history
            // gradient.  Four timesteps' read-backs are in flight at a time (one dependent round trip per timestep cost the
            // pass 20 %, profiles/r05_l_*).
            constexpr int RB = CMAX < 4 ? CMAX : 4;  // (CMAX is a power of two)
#pragma unroll
            for (int kb = CMAX - RB; kb >= 0; kb -= RB) {
                if (kb >= cnt) continue;
                slk_vec<VEC> c0[RB];
                float gpt[RB];
#pragma unroll
                for (int j = RB - 1; j >= 0; --j) {
                    c0[j] = slk_vzero<VEC>();
                    gpt[j] = 0.0f;
                    if (kb + j < cnt && on) {
                        c0[j] = slk_vload<VEC>(recs + (size_t)(t0 + kb + j) * a.RS + d0);
                        gpt[j] = a.gsn[(bl * (uint32_t)L + (uint32_t)(t0 + kb + j)) * (uint32_t)a.NP];
                    }
                }
#pragma unroll
                for (int j = RB - 1; j >= 0; --j) {
                    const int k = kb + j;
                    if (k < cnt) {
                        if (on) {
#pragma unroll
                            for (int i = 0; i < VEC; ++i) c0[j].v[i] = gpt[j] * c0[j].v[i] + suf.v[i];
                            slk_vstore<VEC>(recs + (size_t)(t0 + k) * a.RS + D + d0, c0[j]);
                        }
#pragma unroll
                        for (int i = 0; i < VEC; ++i) suf.v[i] += e[k < CMAX ? k : 0].v[i];
                    }
                }
            }
        }
    }
    const double tot = slk_block_sum_256(loss_acc, red);
    if (threadIdx.x == 0) a.loss_partial[blockIdx.x] = tot / (double)M;
}

// item of occurrence r = pos * NP + s (pos = chunk-local timestep; s = 0: the sequence's own item,
// s > 0: candidate s - 1 in the draw layout of sequence/implicit.py:266-286) and its minibatch
__device__ __forceinline__ uint32_t slk_seq_occ_item(const int64_t *seqs, const uint32_t *neg32, uint32_t r,
                                                     uint32_t n_seq, uint32_t L, uint32_t NP, uint32_t bsz,
                                                     uint32_t *mb_out) {
    const uint32_t nn = NP - 1;
    const uint32_t pos = r / NP, s = r - pos * NP;
    const uint32_t sc = pos / L, t = pos - sc * L;
    const uint32_t mb = sc / bsz;
    *mb_out = mb;
    if (s == 0) return (uint32_t)seqs[pos];
    const uint32_t b0 = mb * bsz, bl = sc - b0;
    const uint32_t Bs = (n_seq - b0 < bsz) ? n_seq - b0 : bsz;
    return neg32[(size_t)b0 * nn * L + ((size_t)(s - 1) * Bs + bl) * L + t];
}

// key = (minibatch, item), value = r
__global__ __launch_bounds__(256) void k_seq_item_keys(const int64_t *seqs, const uint32_t *neg32, uint32_t nocc,
                                                       uint32_t n_seq, uint32_t L, uint32_t NP, uint32_t bsz,
                                                       unsigned ibits, uint32_t *key, uint32_t *val) {
    for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < nocc; r += gridDim.x * 256) {
        uint32_t mb;
        const uint32_t item = slk_seq_occ_item(seqs, neg32, r, n_seq, L, NP, bsz, &mb);
        key[r] = (mb << ibits) | item;
        val[r] = r;
    }
}

// BloomEmbedding item layer: occurrence r feeds the n_hash hashed rows of its item:
// key[r*H + h] = (minibatch, row_h(item)), value = r (same record as the plain occurrence)
__global__ __launch_bounds__(256) void k_seq_item_bloom_keys(const int64_t *seqs, const uint32_t *neg32,
                                                             uint32_t nocc, uint32_t n_seq, uint32_t L, uint32_t NP,
                                                             uint32_t bsz, unsigned cbits, slk_bloom_dev ib,
                                                             uint32_t *key, uint32_t *val) {
    const uint32_t H = (uint32_t)ib.n_hash;
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < nocc * H; e += gridDim.x * 256) {
        const uint32_t r = e / H, h = e - r * H;
        uint32_t mb;
        const uint32_t item = slk_seq_occ_item(seqs, neg32, r, n_seq, L, NP, bsz, &mb);
        key[e] = (mb << cbits) | slk_bloom_row(ib, item, (int)h);
        val[e] = r;
    }
}

// PoolNet.user_representation's final state for ONE sequence (sequence/implicit.py:331-335)
template <int VEC, int G>
__global__ void k_seq_final_repr(const float *E, slk_bloom_dev ib, int D, const int64_t *seq, int L, float *rep) {
    const int lane = threadIdx.x;
    const int d0 = lane * VEC;
    if (d0 >= D) return;
    slk_vec<VEC> S = slk_vzero<VEC>(), Cn = slk_vzero<VEC>();
    for (int t = 0; t < L; ++t) {
        const slk_vec<VEC> e = slk_emb_vec<VEC>(E, ib, (uint32_t)seq[t], D, d0, true);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            S.v[i] += e.v[i];
            Cn.v[i] += (e.v[i] != 0.0f) ? 1.0f : 0.0f;
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) S.v[i] = S.v[i] / (Cn.v[i] + 1.0f);
    slk_vstore<VEC>(rep + d0, S);
}

// PoolNet.forward of one representation against many items (sequence/representations.py:136-144)
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_seq_predict(const float *rep, const float *V, const float *bi,
                                                     slk_bloom_dev ib, int D, const int64_t *items, int64_t n,
                                                     float *out) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    const slk_vec<VEC> r = on ? slk_vload<VEC>(rep + d0) : slk_vzero<VEC>();
    for (int64_t k = (int64_t)blockIdx.x * GPB + grp; k < n; k += (int64_t)gridDim.x * GPB) {
        const int64_t i = items ? items[k] : k;
        const slk_vec<VEC> b = slk_emb_vec<VEC>(V, ib, (uint32_t)i, D, d0, on);
        const float s = bi[i] + slk_chain_dot<VEC, G>(r, b);  // the score's definition: slk_kernels.h, slk_eval.hip
        if (lane == 0) out[k] = s;
    }
}

typedef void (*seq_pass_fn)(slk_seq_args);

static int poolnet_train_impl(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim, int64_t padding_idx,
                              const int64_t *d_sequences, int64_t n_seq, int64_t seq_len, int64_t batch_size,
                              int32_t loss, int32_t n_neg, const int64_t *d_neg_in, int64_t *d_neg_out,
                              float *d_mb_loss, void *stream, bool reserve_only);

SLK_EXPORT int slk_poolnet_train(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim, int64_t padding_idx,
                                 const int64_t *d_sequences, int64_t n_seq, int64_t seq_len, int64_t batch_size,
                                 int32_t loss, int32_t n_neg, const int64_t *d_neg_in, int64_t *d_neg_out,
                                 float *d_mb_loss, void *stream) {
    return poolnet_train_impl(ctx, tables, optim, padding_idx, d_sequences, n_seq, seq_len, batch_size, loss, n_neg,
                              d_neg_in, d_neg_out, d_mb_loss, stream, false);
}

SLK_EXPORT int slk_poolnet_reserve(slk_ctx *ctx, const slk_tables *tables, const slk_optim *optim, int64_t n_seq,
                                   int64_t seq_len, int64_t batch_size, int32_t loss, int32_t n_neg, void *stream) {
    if (!optim) return slk_fail(ctx, SLK_EINVAL, "optim is NULL");
    slk_optim o = *optim;
    return poolnet_train_impl(ctx, tables, &o, 0, nullptr, n_seq, seq_len, batch_size, loss, n_neg, nullptr, nullptr,
                              nullptr, stream, true);
}

static int poolnet_train_impl(slk_ctx *ctx, const slk_tables *tables, slk_optim *optim, int64_t padding_idx,
                              const int64_t *d_sequences, int64_t n_seq, int64_t seq_len, int64_t batch_size,
                              int32_t loss, int32_t n_neg, const int64_t *d_neg_in, int64_t *d_neg_out,
                              float *d_mb_loss, void *stream, bool reserve_only) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    const unsigned TM = 10u;  // tables 1 (item_embeddings) and 3 (item_biases)
    if ((rc = slk_check_tables(ctx, tables, TM, &vec, &g))) return rc;
    if ((rc = slk_check_optim(ctx, optim, TM))) return rc;
    if (n_seq < 0 || batch_size < 1 || seq_len < 1)
        return slk_fail(ctx, SLK_EINVAL, "slk_poolnet_train: n_seq %lld batch_size %lld seq_len %lld", (long long)n_seq,
                        (long long)batch_size, (long long)seq_len);
    if (loss < SLK_LOSS_POINTWISE || loss > SLK_LOSS_ADAPTIVE_HINGE)
        return slk_fail(ctx, SLK_EINVAL, "unknown loss kind %d", loss);
    const bool adaptive = loss == SLK_LOSS_ADAPTIVE_HINGE;
    const int nn = adaptive ? n_neg : 1;
    if (nn < 1 || nn > 1024) return slk_fail(ctx, SLK_EINVAL, "num_negative_samples %d outside [1, 1024]", nn);
    const int NP = nn + 1;
    if (padding_idx < -1 || padding_idx >= tables->num_items)
        return slk_fail(ctx, SLK_EINVAL, "padding_idx %lld outside [-1, num_items)", (long long)padding_idx);
    if (n_seq == 0) return SLK_OK;
    if (!reserve_only && (!d_sequences || !d_mb_loss)) return slk_fail(ctx, SLK_EINVAL, "slk_poolnet_train: NULL pointer");
    const int D = tables->dim;
    const int64_t L = seq_len;
    const int NG = 256 / g, DL = g * vec;
    const size_t lds_bytes = ((size_t)L * DL + 2 * (size_t)NG * DL) * 4;
    if (lds_bytes > 160 * 1024 - 4096)
        return slk_fail(ctx, SLK_EINVAL, "sequence length %lld x dim %d does not fit the 160 KB LDS of a CU",
                        (long long)L, D);
    const int64_t bsz = batch_size < n_seq ? batch_size : n_seq;
    if (bsz * L * NP * (tables->item_bloom ? tables->item_bloom->n_hash : 1) >= ((int64_t)1 << 31))
        return slk_fail(ctx, SLK_EINVAL, "batch_size * seq_len * lookups per timestep must be < 2^31");
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;

    const unsigned ibits = slk_bits_for((uint64_t)tables->num_items - 1);
    // item_embedding_layer = BloomEmbedding (sequence/representations.py:62-68): hashed-row owner pass
    slk_bloom_dev ibd;
    slk_bloom_to_dev(tables->item_bloom, &ibd);
    const int Hi = ibd.n_hash;
    const unsigned icbits = Hi ? slk_bits_for((uint64_t)ibd.rows - 1) : 0;
    const unsigned kbits = icbits > ibits ? icbits : ibits;
    // minibatches per chunk: keys fit 32 bits, occurrences < 2^31, ~8M timesteps of scratch
    int64_t mb_per_chunk = (int64_t)1 << (32 - kbits);
    const int64_t cap_ts = ctx->opt_chunk_interactions;  // timesteps of scratch per chunk (~8M)
    if (mb_per_chunk > 32768) mb_per_chunk = 32768;  // gridDim.y of the mask-count launch
    if (mb_per_chunk * bsz * L > cap_ts) mb_per_chunk = cap_ts / (bsz * L);
    const int64_t occ_mult = (int64_t)NP * (Hi ? Hi : 1);
    while (mb_per_chunk > 1 && mb_per_chunk * bsz * L * occ_mult >= ((int64_t)1 << 31)) mb_per_chunk >>= 1;
    if (mb_per_chunk < 1) mb_per_chunk = 1;
    const int64_t chunk_seqs = mb_per_chunk * bsz;
    const size_t ns_max = (size_t)(chunk_seqs < n_seq ? chunk_seqs : n_seq);
    const size_t nts_max = ns_max * L;

    // The value-independent prep of a chunk (negatives, mask counts, the sort of the occurrences by item, the long-run flags)
    // is prepared on the ctx's second stream while the previous chunk's passes run, as in slk_bilinear_train (option
    // "overlap_prep"; timesteps per minibatch >= "overlap_min_batch"): its buffers exist per set (ctx->pb[0|1]).
    const int64_t n_chunks = (n_seq + chunk_seqs - 1) / chunk_seqs;
    const int nsets = (ctx->opt_overlap_prep && n_chunks > 1 && bsz * L >= ctx->opt_overlap_min_batch) ? 2 : 1;
    for (int st = 0; st < nsets; ++st) {
        slk_prep_bufs &pb = ctx->pb[st];
        if ((rc = slk_ensure(ctx, pb.neg32, nts_max * nn * 4))) return rc;
        for (int b = 0; b < 2; ++b) {
            if ((rc = slk_ensure(ctx, pb.ikey[b], nts_max * NP * 4))) return rc;
            if ((rc = slk_ensure(ctx, pb.ipay[b], nts_max * NP * 4))) return rc;
            if (Hi && (rc = slk_ensure(ctx, pb.bik[b], nts_max * NP * Hi * 4))) return rc;
            if (Hi && (rc = slk_ensure(ctx, pb.bip[b], nts_max * NP * Hi * 4))) return rc;
        }
        if ((rc = slk_ensure(ctx, pb.lflags, (size_t)mb_per_chunk * 4))) return rc;
        if ((rc = slk_ensure(ctx, ctx->extra[SQ_MCOUNT + (st ? SQ_MCOUNT_B - SQ_MCOUNT : 0)], (size_t)mb_per_chunk * 4))) return rc;
    }
    const int RS = 2 * ((D + 3) / 4 * 4);  // record = [representation | history gradient], 16-B granular halves
    if ((rc = slk_ensure(ctx, ctx->snap, (size_t)bsz * L * RS * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[SQ_GSN], (size_t)bsz * L * NP * 4))) return rc;  // dL/dscore per (timestep, pair)
    const unsigned max_grid = (unsigned)ctx->num_cus * 8;
    if ((rc = slk_ensure(ctx, ctx->losspart, (size_t)max_grid * 8))) return rc;
    const bool dense = optim->kind == SLK_OPT_ADAM_DENSE || optim->kind == SLK_OPT_ADAGRAD_DENSE;
    if (dense) {
        const size_t elems[4] = {0, (size_t)(Hi ? ibd.rows : tables->num_items) * D, 0, (size_t)tables->num_items};
        if ((rc = slk_ensure_dgrad(ctx, elems, TM, s))) return rc;
    }
    const int upd = slk_upd_for(optim->kind);
    seq_pass_fn spass = nullptr;
    slk_item_fns ipass = {nullptr, nullptr}, ipass_rows = ipass, ipass_bias = ipass;
    // register-resident sequence pass when a group's chunk of timesteps fits 16 rows of VGPRs
    const bool reg_pass = L <= 256 && (L + NG - 1) / NG <= 16 && ctx->opt_seq_variant != 0;
#define SLK_PICK(V_, G_)                                                                 \
    do {                                                                                 \
        constexpr int CM_ = (G_) < 16 ? (G_) : 16;                                       \
        if (reg_pass && Hi)                                                              \
            spass = adaptive ? k_seq_pass_reg<V_, G_, true, true, CM_> : k_seq_pass_reg<V_, G_, false, true, CM_>; \
        else if (reg_pass)                                                               \
            spass = adaptive ? k_seq_pass_reg<V_, G_, true, false, CM_> : k_seq_pass_reg<V_, G_, false, false, CM_>; \
        else if (Hi)                                                                     \
            spass = adaptive ? k_seq_pass<V_, G_, true, true> : k_seq_pass<V_, G_, false, true>; \
        else                                                                             \
            spass = adaptive ? k_seq_pass<V_, G_, true, false> : k_seq_pass<V_, G_, false, false>; \
        ipass = slk_item_pass_fn<V_, G_, SLK_ITEM_SEQ>(upd);                             \
        if (Hi) {                                                                        \
            ipass_rows = slk_item_pass_fn<V_, G_, SLK_ITEM_SEQ, SLK_PART_ROWS>(upd);     \
            ipass_bias = slk_item_pass_fn<V_, G_, SLK_ITEM_SEQ, SLK_PART_BIAS>(upd);     \
        }                                                                                \
    } while (0)
    SLK_FOR_LAYOUT(vec, g, SLK_PICK);
#undef SLK_PICK
    if (!reg_pass && lds_bytes > 48 * 1024)
        SLK_HIP(ctx, hipFuncSetAttribute((const void *)spass, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes));
    const unsigned gpb = 256u / (unsigned)g;

    if (reserve_only) {
        if (nsets == 2 && (rc = slk_prep_stream_init(ctx))) return rc;
        for (int st = 0; st < nsets; ++st) {  // pinned read-back buffers + events of the long-run flags (slk_bilinear.hip)
            slk_prep_bufs &pb = ctx->pb[st];
            if ((rc = slk_ensure_lflags_host(ctx, pb, (size_t)mb_per_chunk))) return rc;
            pb.h_lflags_n = 0;
            if (!pb.ev_lflags) SLK_HIP(ctx, hipEventCreateWithFlags(&pb.ev_lflags, hipEventDisableTiming));
        }
        // sampler and sort scratch of the largest chunk, so that the training call allocates nothing
        if ((rc = slk_sample_reserve(ctx, tables->num_items, (int64_t)nts_max * nn))) return rc;
        return slk_sort_reserve(ctx, nts_max * (size_t)occ_mult);
    }
    int64_t mb_global = 0;
    // ---- prep of the chunk starting at sequence c0 into buffer set `set`, on stream s (ids only: value-independent)
    auto do_prep = [&](int64_t c0, int set, hipStream_t s) -> int {
        int rc;
        slk_prep_bufs &fb = ctx->pb[set];
        const uint32_t ns = (uint32_t)((n_seq - c0 < chunk_seqs) ? (n_seq - c0) : chunk_seqs);
        const uint32_t nts = ns * (uint32_t)L;
        const uint32_t nocc = nts * (uint32_t)NP;
        const int64_t *cs = d_sequences + c0 * L;
        uint32_t *neg32 = (uint32_t *)fb.neg32.p;
        const uint32_t n_mb = (uint32_t)((ns + bsz - 1) / bsz);
        // ---- negatives: one randint per minibatch == one contiguous draw over the chunk
        if (d_neg_in) {
            slk_prof_begin(ctx, SLK_K_SAMPLE, s);
            if ((rc = slk_launch_i64_to_u32(ctx, d_neg_in + c0 * L * nn, neg32, (size_t)nts * nn, s))) return rc;
            if (d_neg_out)
                SLK_HIP(ctx, hipMemcpyAsync(d_neg_out + c0 * L * nn, d_neg_in + c0 * L * nn, (size_t)nts * nn * 8,
                                            hipMemcpyDeviceToDevice, s));
            slk_prof_end(ctx, s);
        } else {
            if ((rc = slk_sample_u32(ctx, tables->num_items, (int64_t)nts * nn, neg32,
                                     d_neg_out ? d_neg_out + c0 * L * nn : nullptr, s)))
                return rc;
        }
        // ---- prep: mask counts; occurrences sorted by (minibatch, item)
        slk_prof_begin(ctx, SLK_K_PREP, s);
        uint32_t *mcount = (uint32_t *)ctx->extra[set ? SQ_MCOUNT_B : SQ_MCOUNT].p;
        SLK_HIP(ctx, hipMemsetAsync(mcount, 0, (size_t)n_mb * 4, s));
        {
            unsigned gx = (unsigned)(((size_t)bsz * L + 8191) / 8192);
            if (gx > 256) gx = 256;
            hipLaunchKernelGGL(k_seq_count, dim3(gx, n_mb), dim3(256), 0, s, cs, ns, (uint32_t)L, (uint32_t)bsz,
                               mcount);
            SLK_LAUNCH_CHECK(ctx, "k_seq_count");
        }
        const unsigned mbbits = slk_bits_for((uint64_t)(n_mb - 1));
        hipLaunchKernelGGL(k_seq_item_keys, dim3(slk_grid_for(ctx, nocc, 256)), dim3(256), 0, s, cs,
                           (const uint32_t *)neg32, nocc, ns, (uint32_t)L, (uint32_t)NP, (uint32_t)bsz, ibits,
                           (uint32_t *)fb.ikey[0].p, (uint32_t *)fb.ipay[0].p);
        SLK_LAUNCH_CHECK(ctx, "k_seq_item_keys");
        if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)fb.ikey[0].p, (uint32_t *)fb.ikey[1].p,
                                         (const uint32_t *)fb.ipay[0].p, (uint32_t *)fb.ipay[1].p, nocc,
                                         ibits + mbbits, s, true)))
            return rc;
        // which minibatches hold a LONG run of the plain occurrence list (slk_kernels.h, k_item_long_flags): fetched once
        // per chunk; the usual minibatch (none) gets the plain item pass with no stitch kernel behind it
        if ((rc = slk_ensure(ctx, fb.lflags, (size_t)n_mb * 4))) return rc;
        SLK_HIP(ctx, hipMemsetAsync(fb.lflags.p, 0, (size_t)n_mb * 4, s));
        hipLaunchKernelGGL(k_item_long_flags, dim3(slk_grid_for(ctx, nocc / (4 * gpb) + n_mb, 256)), dim3(256), 0, s,
                           (const uint32_t *)fb.ikey[1].p, nocc, (uint32_t)bsz * (uint32_t)L * (uint32_t)NP, 4u * gpb,
                           (uint32_t)((1ull << ibits) - 1), padding_idx < 0 ? 0xffffffffu : (uint32_t)padding_idx, 0xffffffffu,
                           (int *)fb.lflags.p);
        SLK_LAUNCH_CHECK(ctx, "k_item_long_flags");
        if ((rc = slk_ensure_lflags_host(ctx, fb, n_mb))) return rc;
        SLK_HIP(ctx, hipMemcpyAsync(fb.h_lflags, fb.lflags.p, (size_t)n_mb * 4, hipMemcpyDeviceToHost, s));
        if (!fb.ev_lflags) SLK_HIP(ctx, hipEventCreateWithFlags(&fb.ev_lflags, hipEventDisableTiming));
        SLK_HIP(ctx, hipEventRecord(fb.ev_lflags, s));
        if (Hi) {
            hipLaunchKernelGGL(k_seq_item_bloom_keys, dim3(slk_grid_for(ctx, (size_t)nocc * Hi, 256)), dim3(256), 0, s, cs,
                               (const uint32_t *)neg32, nocc, ns, (uint32_t)L, (uint32_t)NP, (uint32_t)bsz, icbits, ibd,
                               (uint32_t *)fb.bik[0].p, (uint32_t *)fb.bip[0].p);
            SLK_LAUNCH_CHECK(ctx, "k_seq_item_bloom_keys");
            if ((rc = slk_sort_pairs_u32_u32(ctx, (const uint32_t *)fb.bik[0].p,
                                             (uint32_t *)fb.bik[1].p,
                                             (const uint32_t *)fb.bip[0].p,
                                             (uint32_t *)fb.bip[1].p, (size_t)nocc * Hi, icbits + mbbits, s, true)))
                return rc;
        }
        slk_prof_end(ctx, s);
        return SLK_OK;
    };

    // ---- the minibatches of one prepared chunk, in order, on the caller's stream
    auto do_passes = [&](int64_t c0, int set) -> int {
        int rc;
        slk_prep_bufs &fb = ctx->pb[set];
        const uint32_t ns = (uint32_t)((n_seq - c0 < chunk_seqs) ? (n_seq - c0) : chunk_seqs);
        const int64_t *cs = d_sequences + c0 * L;
        uint32_t *neg32 = (uint32_t *)fb.neg32.p;
        uint32_t *mcount = (uint32_t *)ctx->extra[set ? SQ_MCOUNT_B : SQ_MCOUNT].p;
        bool lflags_ready = false;
        for (uint32_t b0 = 0, mb = 0; b0 < ns; b0 += (uint32_t)bsz, ++mb, ++mb_global) {
            const uint32_t b1 = (ns - b0 < (uint32_t)bsz) ? ns : b0 + (uint32_t)bsz;
            slk_seq_args q;
            memset(&q, 0, sizeof(q));
            q.E = tables->d_param[1];
            q.bias = tables->d_param[3];
            q.D = D;
            q.L = (int)L;
            q.NP = NP;
            q.seqs = cs;
            q.neg32 = neg32 + (size_t)b0 * nn * L;
            q.s_begin = b0;
            q.s_end = b1;
            q.rec = (float *)ctx->snap.p;
            q.RS = RS;
            q.gsn = (float *)ctx->extra[SQ_GSN].p;
            q.mcount = mcount + mb;
            q.loss_partial = (double *)ctx->losspart.p;
            q.loss_kind = loss;
            q.C = (int)((L + NG - 1) / NG);
            q.ib = ibd;
            unsigned sgrid = b1 - b0;
            if (sgrid > max_grid) sgrid = max_grid;
            slk_prof_begin(ctx, SLK_K_SEQ_PASS, s);
            hipLaunchKernelGGL(spass, dim3(sgrid), dim3(256), reg_pass ? 0 : lds_bytes, s, q);
            SLK_LAUNCH_CHECK(ctx, "k_seq_pass");
            slk_prof_end(ctx, s);

            slk_pass_args a;
            memset(&a, 0, sizeof(a));
            for (int t = 0; t < 4; ++t) {
                a.P[t] = tables->d_param[t];
                a.S1[t] = dense ? (float *)ctx->dgrad[t].p : optim->d_state1[t];
                a.S2[t] = optim->d_state2[t];
            }
            a.D = D;
            a.NP = NP;
            a.begin = b0 * (uint32_t)L;
            a.end = b1 * (uint32_t)L;
            a.snap = (float *)ctx->snap.p;
            a.RS = RS;
            a.gsn = (float *)ctx->extra[SQ_GSN].p;
            a.ibegin = a.begin * (uint32_t)NP;
            a.iend = a.end * (uint32_t)NP;
            a.ikey = (const uint32_t *)fb.ikey[1].p;
            a.imask = (uint32_t)((1ull << ibits) - 1);
            a.ipay = (const uint32_t *)fb.ipay[1].p;
            a.pad_item = padding_idx < 0 ? 0xffffffffu : (uint32_t)padding_idx;
            a.pad_item2 = 0xffffffffu;
            a.loss_partial = (double *)ctx->losspart.p;
            a.n_loss_partial = (int)sgrid;
            a.mb_loss_out = d_mb_loss + mb_global;
            a.loss_kind = loss;
            a.inv_b = 1.0f;  // the sequence pass already divided its partials by mask.sum()
            slk_set_opt_coeffs(a, optim);
            a.nt = ctx->opt_nt;
            if (!lflags_ready) {  // the chunk's first sequence pass is queued: the GPU is busy while the host waits
                SLK_HIP(ctx, hipEventSynchronize(fb.ev_lflags));
                lflags_ready = true;
            }
            const bool may_long = !ctx->opt_item_long_gate || fb.h_lflags[mb] != 0;
            slk_prof_begin(ctx, SLK_K_ITEM_PASS, s);
            if (!Hi) {
                if ((rc = slk_launch_item_pass(ctx, ipass, a, g, s, "k_item_pass<SEQ>", may_long, 4))) return rc;
            } else {
                // item biases are indexed by the item id: plain occurrence list, bias only ...
                if ((rc = slk_launch_item_pass(ctx, ipass_bias, a, g, s, "k_item_pass<SEQ,BIAS>", may_long, 4))) return rc;
                // ... while every occurrence feeds the n_hash hashed rows of the compressed table
                slk_pass_args r = a;
                r.mb_loss_out = nullptr;
                r.ikey = (const uint32_t *)fb.bik[1].p;
                r.ipay = (const uint32_t *)fb.bip[1].p;
                r.ibegin = a.ibegin * (uint32_t)Hi;
                r.iend = a.iend * (uint32_t)Hi;
                r.imask = (uint32_t)((1ull << icbits) - 1);
                r.pad_item = tables->item_bloom->skip_row < 0 ? 0xffffffffu : (uint32_t)tables->item_bloom->skip_row;
                if ((rc = slk_launch_item_pass(ctx, ipass_rows, r, g, s, "k_item_pass<SEQ,ROWS>"))) return rc;
            }
            slk_prof_end(ctx, s);
            if (dense && (rc = slk_dense_sweeps(ctx, tables->d_param, optim, TM, s))) return rc;
            optim->step += 1;
        }
        return SLK_OK;
    };

    auto do_chunk = [&](int64_t c0, int set) -> int { return do_passes(c0, set); };

    if (nsets == 1) {
        for (int64_t c0 = 0; c0 < n_seq; c0 += chunk_seqs) {
            if ((rc = do_prep(c0, 0, s))) return rc;
            if ((rc = do_chunk(c0, 0))) return rc;
        }
        return SLK_OK;
    }
    // pipeline: prep(c + 1) on ctx->prep_stream beside passes(c) on the caller's stream (slk_bilinear.hip)
    if ((rc = slk_prep_stream_init(ctx))) return rc;
    hipStream_t ps = ctx->prep_stream;
    SLK_HIP(ctx, hipEventRecord(ctx->ev_start, s));  // inputs produced on the caller's stream
    SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_start, 0));
    if ((rc = do_prep(0, 0, ps))) return rc;
    SLK_HIP(ctx, hipEventRecord(ctx->ev_prep[0], ps));
    int set = 0;
    for (int64_t c0 = 0; c0 < n_seq; c0 += chunk_seqs, set ^= 1) {
        SLK_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_prep[set], 0));
        if (c0 + chunk_seqs < n_seq) {
            if (c0 > 0) SLK_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_done[set ^ 1], 0));  // that set's last reader: the chunk before
            if ((rc = do_prep(c0 + chunk_seqs, set ^ 1, ps))) return rc;
            SLK_HIP(ctx, hipEventRecord(ctx->ev_prep[set ^ 1], ps));
            ++ctx->stat_overlapped;
        }
        if ((rc = do_chunk(c0, set))) return rc;
        SLK_HIP(ctx, hipEventRecord(ctx->ev_done[set], s));
    }
    ctx->last_stream = s;  // every prep is ordered before the tail of the caller's stream
    return SLK_OK;
}

SLK_EXPORT int slk_poolnet_predict(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_sequence,
                                   int64_t seq_len, const int64_t *d_items, int64_t n, float *d_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, tables, 10u, &vec, &g))) return rc;
    slk_bloom_dev ibd;
    slk_bloom_to_dev(tables->item_bloom, &ibd);
    if (n < 0 || seq_len < 1 || !d_sequence || (n > 0 && !d_out))
        return slk_fail(ctx, SLK_EINVAL, "slk_poolnet_predict: bad arguments");
    if (n == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    if ((rc = slk_ensure(ctx, ctx->extra[SQ_REP], 256 * 4 * 4))) return rc;
    float *rep = (float *)ctx->extra[SQ_REP].p;
    slk_prof_begin(ctx, SLK_K_SCORE, s);
#define SLK_SEQ_PREDICT(V_, G_)                                                                                  \
    do {                                                                                                         \
        hipLaunchKernelGGL((k_seq_final_repr<V_, G_>), dim3(1), dim3(64), 0, s, (const float *)tables->d_param[1], \
                           ibd, (int)tables->dim, d_sequence, (int)seq_len, rep);                                \
        hipLaunchKernelGGL((k_seq_predict<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n, 256 / G_)), dim3(256), 0, s, \
                           (const float *)rep, (const float *)tables->d_param[1],                                \
                           (const float *)tables->d_param[3], ibd, (int)tables->dim, d_items, n, d_out);         \
    } while (0)
    if (!d_items) {
        // every item: the representation, then the item table through the matrix cores (slk_eval.hip)
        if (n != tables->num_items)
            return slk_fail(ctx, SLK_EINVAL, "slk_poolnet_predict: d_items == NULL scores all %lld items, n is %lld",
                            (long long)tables->num_items, (long long)n);
#define SLK_SEQ_REP(V_, G_)                                                                                      \
        hipLaunchKernelGGL((k_seq_final_repr<V_, G_>), dim3(1), dim3(64), 0, s, (const float *)tables->d_param[1], \
                           ibd, (int)tables->dim, d_sequence, (int)seq_len, rep)
        SLK_FOR_LAYOUT(vec, g, SLK_SEQ_REP);
#undef SLK_SEQ_REP
        SLK_LAUNCH_CHECK(ctx, "k_seq_final_repr");
        rc = slk_eval_predict_all(ctx, tables, rep, nullptr, nullptr, d_out, s);
        slk_prof_end(ctx, s);
        return rc;
    }
    SLK_FOR_LAYOUT(vec, g, SLK_SEQ_PREDICT);
#undef SLK_SEQ_PREDICT
    SLK_LAUNCH_CHECK(ctx, "k_seq_predict");
    slk_prof_end(ctx, s);
    return SLK_OK;
}
