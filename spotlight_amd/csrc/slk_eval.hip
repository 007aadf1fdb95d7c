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

// ---------------------------------------------------------------------------------------------------------------------------
// Scores as a GEMM on the matrix cores.  scores[r][i] = <rep_r, V_i> is a (rows x D) x (D x items) product in fp32: the one
// place on this path where MFMA is the right unit.  v_mfma_f32_32x32x2_f32 is EXACT fp32 -- bit for bit the k-ordered chain
// acc = fmaf(a_k, b_k, acc) (cdna_hip_programming.md 3) -- so "the score" of this package is DEFINED as that chain over
// d = 0 .. D-1 (slk_chain_dot in slk_kernels.h is the same chain on the vector unit, for explicit (row, item) pairs), and
// predict(), the batched score rows and the fused ranking below agree bit for bit whichever unit formed them.
//
// k_score_gemm: a workgroup (4 waves) keeps a tile of RT = 32 * MT representations in LDS and sweeps a chunk of the item
// table in blocks of 128 rows (each wave 32 of them: one 32x32 accumulator tile per 32 representations), staging every block
// through registers into LDS while the previous one is multiplied.  MODE WRITE stores the scores (slk_*_scores, predict of
// one row against every item); MODE COUNT never stores them: every score is compared with its row's target score as it
// leaves the accumulators (rank = #greater + (#equal + 1) / 2, scipy's 'average'), per-lane packed counters, one atomic per
// row and workgroup at the end -- the [rows x items] score matrix of the round-3 path (written, then read back per target)
// does not exist.  Item traffic: items * (4 D + 4) bytes per row tile, i.e. per 64 rows.
// ---------------------------------------------------------------------------------------------------------------------------
#define SLK_GEMM_IB 128  // item rows per block of the sweep (4 waves x 32)
#define SLK_GEMM_KC 64   // depth of one staged chunk
#define SLK_GEMM_KS 65   // LDS row stride in floats (odd: the 32 lanes of an operand read hit 32 banks)

struct slk_gemm_args {
    const float *rep, *rbias;     // [groups][D], [groups] (nullptr: no row bias, PoolNet)
    const int64_t *rowmap;        // row r belongs to group rowmap[r] (nullptr: r)
    const int64_t *gmap;          // group g's representation is row gmap[g] of rep / rbias (nullptr: g) -- a plain user table
                                  // is read in place through the user ids
    const float *V, *bi;
    slk_bloom_dev ib;
    int D;
    int64_t R, I, items_per_wg;
    float *out;                   // WRITE: [R][I]
    const float *st;              // COUNT: the row's target score
    unsigned *gt, *eq;            // COUNT: per-row counts (atomicAdd)
};

#if defined(__HIPCC__)
typedef float slk_f32x16 __attribute__((ext_vector_type(16)));
#else
struct slk_f32x16 {
    float v[16];
    float &operator[](int i) { return v[i]; }
};
#endif

// element (item, d) of the item-side operand: one row of a plain table or the sum of a bloom id's hashed rows
__device__ __forceinline__ float slk_gemm_item_elem(const float *V, const slk_bloom_dev &b, uint32_t id, int D, int d) {
    if (b.n_hash == 0) return V[(size_t)id * D + d];
    float v = V[(size_t)slk_bloom_row(b, id, 0) * D + d];
    for (int h = 1; h < b.n_hash; ++h) v += V[(size_t)slk_bloom_row(b, id, h) * D + d];
    return v;
}

template <int MT, bool COUNT, bool VEC4, bool AREG>
__global__ __launch_bounds__(256) SLK_WAVES_PER_EU(2) void k_score_gemm(slk_gemm_args a) {  // (two workgroups per CU by LDS footprint, eval_gemm: 2 waves per SIMD)
    constexpr int RT = 32 * MT, IB = SLK_GEMM_IB, KC = SLK_GEMM_KC, KS = SLK_GEMM_KS;
    constexpr int BREG = VEC4 ? IB * KC / 256 : 1;  // staged floats per thread (plain tables of dim % 4 == 0)
    HIP_DYNAMIC_SHARED(float, lds)
    float *sA = lds;                      // [RT][KS]
    float *sB = sA + RT * KS;             // [IB][KS]
    float *s_rb = sB + IB * KS;           // [RT]
    float *s_st = s_rb + RT;              // [RT]
    unsigned *s_gt = reinterpret_cast<unsigned *>(s_st + RT);  // [RT]
    unsigned *s_eq = s_gt + RT;           // [RT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D;
    const int64_t r0 = (int64_t)blockIdx.y * RT;
    const int64_t i_begin = (int64_t)blockIdx.x * a.items_per_wg;
    const int64_t i_end = (a.I - i_begin < a.items_per_wg) ? a.I : i_begin + a.items_per_wg;
    if (i_begin >= i_end) return;
    const int nkc = (D + KC - 1) / KC;
    for (int t = tid; t < RT; t += 256) {
        const bool live = r0 + t < a.R;
        int64_t g = live ? (a.rowmap ? a.rowmap[r0 + t] : r0 + t) : 0;
        if (live && a.gmap) g = a.gmap[g];
        s_rb[t] = (live && a.rbias) ? a.rbias[g] : 0.0f;
        if (COUNT) {
            s_st[t] = live ? a.st[r0 + t] : 0.0f;
            s_gt[t] = 0u;
            s_eq[t] = 0u;
        }
    }
    // the representations' chunk [kc, kc + KC) -> sA (zero beyond D and beyond the last row)
    auto stage_a = [&](int kc) {
        for (int e = tid; e < RT * KC; e += 256) {
            const int row = e / KC, k = e - row * KC;
            float v = 0.0f;
            if (r0 + row < a.R && kc + k < D) {
                int64_t g = a.rowmap ? a.rowmap[r0 + row] : r0 + row;
                if (a.gmap) g = a.gmap[g];
                v = a.rep[(size_t)g * D + kc + k];
            }
            sA[row * KS + k] = v;
        }
    };
    // block [i0, i0 + IB) x chunk [kc, kc + KC) of the item operand -> registers (the loads are in flight while the previous
    // block is multiplied), then registers -> sB
    float breg[BREG];
    auto fetch_b = [&](int64_t i0, int kc) {
        if (VEC4) {
#pragma unroll
            for (int u = 0; u < BREG / 4; ++u) {
                const int q = tid + 256 * u, item = q / (KC / 4), k = (q - item * (KC / 4)) * 4;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (i0 + item < i_end && kc + k < D) v = *reinterpret_cast<const float4 *>(a.V + (size_t)(i0 + item) * D + kc + k);
                breg[4 * u] = v.x;
                breg[4 * u + 1] = v.y;
                breg[4 * u + 2] = v.z;
                breg[4 * u + 3] = v.w;
            }
        }
        // (any other table -- odd dims, BloomEmbedding rows -- is staged in store_b, element by element: not the fast route)
    };
    auto store_b = [&](int64_t i0, int kc) {
        if (VEC4) {
#pragma unroll
            for (int u = 0; u < BREG / 4; ++u) {
                const int q = tid + 256 * u, item = q / (KC / 4), k = (q - item * (KC / 4)) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) sB[item * KS + k + c] = breg[4 * u + c];
            }
        } else {
            for (int q = tid; q < IB * KC; q += 256) {
                const int item = q / KC, k = q - item * KC;
                sB[item * KS + k] =
                    (i0 + item < i_end && kc + k < D) ? slk_gemm_item_elem(a.V, a.ib, (uint32_t)(i0 + item), D, kc + k) : 0.0f;
            }
        }
    };
    // COUNT: (#equal << 16) | #greater among this lane's columns so far, per accumulator element's row.  (One wave vote per
    // comparison with the row's counter kept by one lane -- 16 * MT registers less -- was measured: 9.0 instead of 6.6 ms
    // for 4096 x 10^6 scores, the votes' scalar work sits in the epilogue of every block: profiles/r04_g_*.)
    unsigned cnt[MT * 16];
    if (COUNT) {
#pragma unroll
        for (int i = 0; i < MT * 16; ++i) cnt[i] = 0u;
    }
    if (nkc == 1) stage_a(0);
    fetch_b(i_begin, 0);
#if defined(__HIPCC__)
    float areg[AREG ? MT * (KC / 2) : 1];
    if (AREG) {
        __syncthreads();  // sA is complete (zero beyond D: the chain adds +0 for the padded depth)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int k = 0; k < KC; k += 2) areg[m * (KC / 2) + k / 2] = sA[(m * 32 + (lane & 31)) * KS + (lane >> 5) + k];
    }
#endif
    for (int64_t i0 = i_begin; i0 < i_end; i0 += IB) {
        slk_f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[m][v] = 0.0f;
        for (int c = 0; c < nkc; ++c) {
            const int kc = c * KC;
            __syncthreads();  // the previous chunk's operands have been read
            if (nkc > 1) stage_a(kc);
            store_b(i0, kc);
            __syncthreads();
            // the next chunk (or the next block's first) into the registers
            {
                const bool more_k = c + 1 < nkc;
                const int64_t ni = more_k ? i0 : i0 + IB;
                if (ni < i_end) fetch_b(ni, more_k ? kc + KC : 0);
            }
            const int kw = (D - kc < KC) ? ((D - kc + 1) & ~1) : KC;  // even: the pad column is zero
#if defined(__HIPCC__)
            const float *pa = sA + (lane & 31) * KS + (lane >> 5);
            const float *pb = sB + (wave * 32 + (lane & 31)) * KS + (lane >> 5);
            if (AREG) {
                // the representations' operand lives in registers for the whole sweep (dim <= 64: one chunk): one LDS read per
                // two matrix instructions instead of three
#pragma unroll
                for (int k = 0; k < KC; k += 2) {
                    const float b = pb[k];
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[m * (KC / 2) + k / 2], b, acc[m], 0, 0, 0);
                }
            } else {
#pragma unroll 2
                for (int k = 0; k < kw; k += 2) {
                    const float b = pb[k];
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[m * 32 * KS + k], b, acc[m], 0, 0, 0);
                }
            }
#else
            // the test harness's host build: the same k-ordered fmaf chain per accumulator element, operands from LDS
            for (int k = 0; k < kw; ++k)
                for (int m = 0; m < MT; ++m)
                    for (int v = 0; v < 16; ++v) {
                        const int row = m * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                        acc[m][v] = fmaf(sA[row * KS + k], sB[(wave * 32 + (lane & 31)) * KS + k], acc[m][v]);
                    }
#endif
        }
        // epilogue: accumulator element v of tile m = (row m*32 + (v&3) + 8*(v>>2) + 4*(lane>>5), item i0 + wave*32 + (lane&31))
        const int64_t item = i0 + wave * 32 + (lane & 31);
        const bool live = item < i_end;
        const float bias = live ? a.bi[item] : 0.0f;
        // (WRITE: this lane's output column; the row offsets below are wave-uniform multiples of the row stride)
        float *ocol = COUNT ? nullptr : a.out + (size_t)(r0 + 4 * (lane >> 5)) * a.I + item;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                // (a compiler barrier per four elements: the row constants are re-read from LDS where they are used instead
                // of all 32 * MT of them being hoisted into registers)
                if ((v & 3) == 0) asm volatile("" ::: "memory");
                const int row_a = m * 32 + (v & 3) + 8 * (v >> 2);  // lanes 0..31; lanes 32..63 hold row_a + 4
                const int row = row_a + 4 * (lane >> 5);
                // k_predict: (dot + bu) + bi;  k_seq_predict: bi + dot
                const float sc = a.rbias ? (acc[m][v] + s_rb[row]) + bias : bias + acc[m][v];
                if (COUNT) {
                    const float t = s_st[row];
                    cnt[m * 16 + v] += (live && sc > t ? 1u : 0u) + (live && sc == t ? 0x10000u : 0u);
                } else if (live && r0 + row < a.R) {
                    ocol[(size_t)row_a * a.I] = sc;
                }
            }
    }
    if (COUNT) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = m * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                const unsigned c = cnt[m * 16 + v];
                if (c & 0xffffu) atomicAdd(&s_gt[row], c & 0xffffu);
                if (c >> 16) atomicAdd(&s_eq[row], c >> 16);
            }
        __syncthreads();
        for (int t = tid; t < RT; t += 256)
            if (r0 + t < a.R) {
                if (s_gt[t]) atomicAdd(&a.gt[r0 + t], s_gt[t]);
                if (s_eq[t]) atomicAdd(&a.eq[r0 + t], s_eq[t]);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// A handful of rows against every item (predict(user): ONE row; implicit.py:277-311 with item_ids = None): a GEMV, bound by the
// one pass over the item table -- 32 rows of matrix-core tile around a single live row buy nothing.  k_score_rows streams the
// table in blocks of 256 item rows through registers into LDS (the next block's 64 KB are in flight while the current one is
// used; two workgroups per CU = 128 KB per CU in flight) and every thread walks the score chain of ONE item against the NR
// rows: acc[r] = fmaf(a_r[k], b[k], acc[r]), k ascending -- the chain of slk_chain_dot and of the MFMA, so the scores are the
// same bits whichever kernel forms them.  Plain tables with dim % 4 == 0 (anything else takes k_score_gemm's general loader).
// ---------------------------------------------------------------------------------------------------------------------------
#define SLK_ROWS_IB 256  // items per block: one per thread

template <int NR>
__global__ __launch_bounds__(256) void k_score_rows(slk_gemm_args a) {
    constexpr int IB = SLK_ROWS_IB, KC = SLK_GEMM_KC, KS = SLK_GEMM_KS;
    constexpr int BREG = IB * KC / 256;  // staged floats per thread
    HIP_DYNAMIC_SHARED(float, lds)
    float *sB = lds;              // [IB][KS]
    float *sA = sB + IB * KS;     // [NR][KC]
    float *s_rb = sA + NR * KC;   // [NR]
    const int tid = threadIdx.x;
    const int D = a.D;
    const int64_t i_begin = (int64_t)blockIdx.x * a.items_per_wg;
    const int64_t i_end = (a.I - i_begin < a.items_per_wg) ? a.I : i_begin + a.items_per_wg;
    if (i_begin >= i_end) return;
    const int nkc = (D + KC - 1) / KC;
    auto group_of = [&](int r) -> int64_t {
        int64_t g = a.rowmap ? a.rowmap[r] : r;
        return a.gmap ? a.gmap[g] : g;
    };
    if (tid < NR) s_rb[tid] = (tid < a.R && a.rbias) ? a.rbias[group_of(tid)] : 0.0f;
    auto stage_a = [&](int kc) {
        for (int e = tid; e < NR * KC; e += 256) {
            const int row = e / KC, k = e - row * KC;
            sA[e] = (row < a.R && kc + k < D) ? a.rep[(size_t)group_of(row) * D + kc + k] : 0.0f;
        }
    };
    float breg[BREG];
    auto fetch_b = [&](int64_t i0, int kc) {
#pragma unroll
        for (int u = 0; u < BREG / 4; ++u) {
            const int q = tid + 256 * u, item = q / (KC / 4), k = (q - item * (KC / 4)) * 4;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (i0 + item < i_end && kc + k < D) v = *reinterpret_cast<const float4 *>(a.V + (size_t)(i0 + item) * D + kc + k);
            breg[4 * u] = v.x;
            breg[4 * u + 1] = v.y;
            breg[4 * u + 2] = v.z;
            breg[4 * u + 3] = v.w;
        }
    };
    auto store_b = [&]() {
#pragma unroll
        for (int u = 0; u < BREG / 4; ++u) {
            const int q = tid + 256 * u, item = q / (KC / 4), k = (q - item * (KC / 4)) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) sB[item * KS + k + c] = breg[4 * u + c];
        }
    };
    if (nkc == 1) stage_a(0);
    fetch_b(i_begin, 0);
    for (int64_t i0 = i_begin; i0 < i_end; i0 += IB) {
        float acc[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = 0.0f;
        for (int c = 0; c < nkc; ++c) {
            const int kc = c * KC;
            __syncthreads();  // the previous chunk's operands have been read
            if (nkc > 1) stage_a(kc);
            store_b();
            __syncthreads();
            {
                const bool more_k = c + 1 < nkc;
                const int64_t ni = more_k ? i0 : i0 + IB;
                if (ni < i_end) fetch_b(ni, more_k ? kc + KC : 0);
            }
            const int kw = (D - kc < KC) ? D - kc : KC;
            const float *pb = sB + tid * KS;
#pragma unroll 8
            for (int k = 0; k < kw; ++k) {
                const float b = pb[k];
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[r] = fmaf(sA[r * KC + k], b, acc[r]);
            }
        }
        const int64_t item = i0 + tid;
        if (item < i_end) {
            const float bias = a.bi[item];
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (r < a.R) a.out[(size_t)r * a.I + item] = a.rbias ? (acc[r] + s_rb[r]) + bias : bias + acc[r];
        }
    }
}

// st[r] = score(representation of row r, its target item) on the vector unit (the same chain); a target that is on its own
// group's exclusion list scores -FLT_MAX, as the reference's `predictions[excluded] = FLOAT_MAX` makes it
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_rank_target_scores(const float *rep, const float *rbias, const int64_t *rowmap,
                                                            const int64_t *gmap, const float *V, const float *bi, slk_bloom_dev ib, int D,
                                                            const int64_t *tgt, const int64_t *exc_off, const int64_t *exc_items,
                                                            int64_t n_rows, float *st) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    for (int64_t r = (int64_t)blockIdx.x * GPB + grp; r < n_rows; r += (int64_t)gridDim.x * GPB) {
        const int64_t g = rowmap ? rowmap[r] : r, item = tgt[r];
        const int64_t src = gmap ? gmap[g] : g;
        const slk_vec<VEC> x = on ? slk_vload<VEC>(rep + (size_t)src * D + d0) : slk_vzero<VEC>();
        const slk_vec<VEC> y = slk_emb_vec<VEC>(V, ib, (uint32_t)item, D, d0, on);
        const float dot = slk_chain_dot<VEC, G>(x, y);
        float sc = rbias ? (dot + rbias[src]) + bi[item] : bi[item] + dot;
        if (exc_off) {
            unsigned long long hit = 0;
            for (int64_t e = exc_off[g] + lane; e < exc_off[g + 1]; e += G) hit |= (exc_items[e] == item) ? 1ull : 0ull;
            if (slk_group_or<G>(hit)) sc = -FLT_MAX;
        }
        if (lane == 0) st[r] = sc;
    }
}

// The sweep counted every item with its real score.  For every item on the row's exclusion list: take its real score's
// contribution out and put -FLT_MAX's in (dgt / deq: signed corrections, one writer per row)
template <int VEC, int G>
__global__ __launch_bounds__(256) void k_rank_exclusions(const float *rep, const float *rbias, const int64_t *rowmap,
                                                         const int64_t *gmap, const float *V, const float *bi, slk_bloom_dev ib, int D,
                                                         const int64_t *exc_off, const int64_t *exc_items, const float *st,
                                                         int64_t n_rows, int *dgt, int *deq) {
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G, grp = threadIdx.x / G;
    const int d0 = lane * VEC;
    const bool on = d0 < D;
    for (int64_t r = (int64_t)blockIdx.x * GPB + grp; r < n_rows; r += (int64_t)gridDim.x * GPB) {
        const int64_t g = rowmap ? rowmap[r] : r;
        const int64_t src = gmap ? gmap[g] : g;
        const slk_vec<VEC> x = on ? slk_vload<VEC>(rep + (size_t)src * D + d0) : slk_vzero<VEC>();
        const float t = st[r], rb = rbias ? rbias[src] : 0.0f;
        int cg = 0, ce = 0;
        for (int64_t e = exc_off[g]; e < exc_off[g + 1]; ++e) {
            const int64_t item = exc_items[e];
            const slk_vec<VEC> y = slk_emb_vec<VEC>(V, ib, (uint32_t)item, D, d0, on);
            const float dot = slk_chain_dot<VEC, G>(x, y);
            const float sc = rbias ? (dot + rb) + bi[item] : bi[item] + dot;
            cg += (-FLT_MAX > t ? 1 : 0) - (sc > t ? 1 : 0);
            ce += (-FLT_MAX == t ? 1 : 0) - (sc == t ? 1 : 0);
        }
        if (lane == 0) {
            dgt[r] = cg;
            deq[r] = ce;
        }
    }
}

__global__ __launch_bounds__(256) void k_rank_final(const unsigned *gt, const unsigned *eq, const int *dgt, const int *deq,
                                                    int64_t n_rows, double *rank) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
        const double g = (double)gt[r] + (dgt ? (double)dgt[r] : 0.0), e = (double)eq[r] + (deq ? (double)deq[r] : 0.0);
        rank[r] = g + (e + 1.0) * 0.5;
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

// ---- host side ---------------------------------------------------------------------------------------------------------------
enum { EV_ST = 37, EV_CNT = 39 };  // ctx->extra slots of the fused ranking (24, 25: the representations)

typedef void (*gemm_fn)(slk_gemm_args);

// areg: the representations' operand in registers for the whole sweep (plain tables of dim <= 64)
template <bool COUNT>
static gemm_fn gemm_kernel(int mt, bool vec4, bool areg) {
    if (areg) return mt == 1 ? k_score_gemm<1, COUNT, true, true> : k_score_gemm<2, COUNT, true, true>;
    if (mt == 1) return vec4 ? k_score_gemm<1, COUNT, true, false> : k_score_gemm<1, COUNT, false, false>;
    return vec4 ? k_score_gemm<2, COUNT, true, false> : k_score_gemm<2, COUNT, false, false>;
}

// one sweep of the item table per tile of 32 * mt rows.  `count`: compare with a.st and add into a.gt / a.eq, else store a.out
static int eval_gemm(slk_ctx *ctx, const slk_tables *tables, slk_gemm_args a, bool count, hipStream_t s) {
    slk_bloom_to_dev(tables->item_bloom, &a.ib);
    a.V = (const float *)tables->d_param[1];
    a.bi = (const float *)tables->d_param[3];
    a.D = tables->dim;
    a.I = tables->num_items;
    if (a.R <= 0 || a.I <= 0) return SLK_OK;
    if (!count && a.R <= 8 && a.ib.n_hash == 0 && a.D % 4 == 0) {
        // a handful of rows: the streaming form (k_score_rows), two workgroups per CU each sweeping whole blocks of 256 items
        int64_t per = (a.I + 2 * (int64_t)ctx->num_cus - 1) / (2 * (int64_t)ctx->num_cus);
        per = (per + SLK_ROWS_IB - 1) / SLK_ROWS_IB * SLK_ROWS_IB;
        a.items_per_wg = per;
        const int nr = a.R == 1 ? 1 : (a.R == 2 ? 2 : (a.R <= 4 ? 4 : 8));
        const size_t lds = ((size_t)SLK_ROWS_IB * SLK_GEMM_KS + (size_t)nr * SLK_GEMM_KC + nr) * 4;
        gemm_fn fn = nr == 1 ? k_score_rows<1> : (nr == 2 ? k_score_rows<2> : (nr == 4 ? k_score_rows<4> : k_score_rows<8>));
        SLK_HIP(ctx, hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fn, dim3((unsigned)((a.I + per - 1) / per)), dim3(256), lds, s, a);
        SLK_LAUNCH_CHECK(ctx, "k_score_rows");
        return SLK_OK;
    }
    // rows per workgroup: 64 (the item rows are streamed once per row tile; a 128-row tile was measured out: its four
    // accumulator tiles + the staged block leave no room for three workgroups per CU, and the sweep is bound by how many
    // blocks the CU has in flight), 32 for a handful (predict: one row -- a 32-row tile keeps the matrix-core time below the
    // table's streaming time)
    const int mt = a.R > 32 ? 2 : 1;
    const bool vec4 = a.ib.n_hash == 0 && a.D % 4 == 0;
    const int64_t row_tiles = (a.R + 32 * mt - 1) / (32 * mt);
    // item chunks: enough workgroups for the chip (~4 per CU over all row tiles; 2 per CU for a single row tile, whose
    // workgroups are all resident at once: longer sweeps amortise a workgroup's first, unhidden block), whole blocks of
    // SLK_GEMM_IB items
    int64_t want = ((row_tiles == 1 ? 2 : 4) * (int64_t)ctx->num_cus + row_tiles - 1) / row_tiles;
    if (want < 1) want = 1;
    int64_t per = (a.I + want - 1) / want;
    per = (per + SLK_GEMM_IB - 1) / SLK_GEMM_IB * SLK_GEMM_IB;
    if (per > ((int64_t)1 << 22)) per = (int64_t)1 << 22;  // (the packed per-lane counters hold 2^16 - 1 columns: 2^23 items)
    a.items_per_wg = per;
    const int64_t chunks = (a.I + per - 1) / per;
    if (row_tiles > 65535) return slk_fail(ctx, SLK_EINVAL, "scoring: %lld rows per call, at most %d", (long long)a.R, 65535 * 32 * mt);
    size_t lds = ((size_t)(32 * mt + SLK_GEMM_IB) * SLK_GEMM_KS + 4 * 32 * mt) * 4;
    // two resident workgroups per CU, by LDS footprint: the registers would allow three, and three share the LDS bandwidth and
    // the L2 worse (4096 x 10^6: 8.3 ms against 7.06, profiles/r04_h_bench_eval_{2,3}wg.json)
    // (the pad is half of the CU's LDS as the DEVICE reports it -- 160 KB on gfx950 -- and is skipped where a workgroup cannot be
    // granted that much: never more than the device can give)
    if (lds > ctx->lds_per_block)
        return slk_fail(ctx, SLK_EINVAL, "scoring: the sweep needs %zu B of LDS per workgroup, the device grants %zu", lds, ctx->lds_per_block);
    const size_t half_lds = ctx->lds_per_cu / 2 - 256;
    if (lds < half_lds && half_lds <= ctx->lds_per_block) lds = half_lds;
    const bool areg = vec4 && a.D <= SLK_GEMM_KC;
    gemm_fn fn = count ? gemm_kernel<true>(mt, vec4, areg) : gemm_kernel<false>(mt, vec4, areg);
    if (lds > 48 * 1024) SLK_HIP(ctx, hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, dim3((unsigned)chunks, (unsigned)row_tiles), dim3(256), lds, s, a);
    SLK_LAUNCH_CHECK(ctx, "k_score_gemm");
    return SLK_OK;
}

// The rows' representations.  Plain user table: the table itself, addressed through the ids (*rowmap = d_users; no kernel, no
// copy).  BloomEmbedding user table: the hashed rows are summed into scratch first (*rowmap = NULL).
static int eval_user_rows(slk_ctx *ctx, const slk_tables *tables, int vec, int g, const int64_t *d_users, int64_t n_users,
                          const float **rep, const float **rbias, const int64_t **rowmap, hipStream_t s) {
    int rc;
    if (!tables->user_bloom || tables->user_bloom->n_hash == 0) {
        *rep = (const float *)tables->d_param[0];
        *rbias = (const float *)tables->d_param[2];
        *rowmap = d_users;
        return SLK_OK;
    }
    *rowmap = nullptr;
    const int D = tables->dim;
    if ((rc = slk_ensure(ctx, ctx->extra[EV_REP], (size_t)n_users * D * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[EV_RBIAS], (size_t)n_users * 4))) return rc;
    float *rep_w = (float *)ctx->extra[EV_REP].p, *rbias_w = (float *)ctx->extra[EV_RBIAS].p;
    *rep = rep_w;
    *rbias = rbias_w;
    slk_bloom_dev ubd;
    slk_bloom_to_dev(tables->user_bloom, &ubd);
#define SLK_ROWS(V_, G_)                                                                                          \
    hipLaunchKernelGGL((k_eval_user_rows<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n_users, 256 / G_)), dim3(256), 0, s, \
                       (const float *)tables->d_param[0], (const float *)tables->d_param[2], ubd, D, d_users,      \
                       n_users, rep_w, rbias_w)
    SLK_FOR_LAYOUT(vec, g, SLK_ROWS);
#undef SLK_ROWS
    SLK_LAUNCH_CHECK(ctx, "k_eval_user_rows");
    return SLK_OK;
}

static int eval_seq_rows(slk_ctx *ctx, const slk_tables *tables, int vec, int g, const int64_t *d_sequences, int64_t n_seq,
                         int64_t seq_len, float **rep, hipStream_t s) {
    int rc;
    const int D = tables->dim;
    if ((rc = slk_ensure(ctx, ctx->extra[EV_REP], (size_t)n_seq * D * 4))) return rc;
    *rep = (float *)ctx->extra[EV_REP].p;
    slk_bloom_dev ibd;
    slk_bloom_to_dev(tables->item_bloom, &ibd);
#define SLK_ROWS(V_, G_)                                                                                         \
    hipLaunchKernelGGL((k_eval_seq_rows<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n_seq, 256 / G_)), dim3(256), 0, s, \
                       (const float *)tables->d_param[1], ibd, D, d_sequences, n_seq, (int)seq_len, *rep)
    SLK_FOR_LAYOUT(vec, g, SLK_ROWS);
#undef SLK_ROWS
    SLK_LAUNCH_CHECK(ctx, "k_eval_seq_rows");
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
    slk_prof_begin(ctx, SLK_K_SCORE, s);
    slk_gemm_args a;
    memset(&a, 0, sizeof(a));
    const float *rep, *rbias;
    const int64_t *gmap;
    if ((rc = eval_user_rows(ctx, tables, vec, g, d_users, n_users, &rep, &rbias, &gmap, s))) return rc;
    a.rep = rep;
    a.rbias = rbias;
    a.gmap = gmap;
    a.R = n_users;
    a.out = d_out;
    rc = eval_gemm(ctx, tables, a, false, s);
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
    slk_prof_begin(ctx, SLK_K_SCORE, s);
    slk_gemm_args a;
    memset(&a, 0, sizeof(a));
    float *rep;
    if ((rc = eval_seq_rows(ctx, tables, vec, g, d_sequences, n_seq, seq_len, &rep, s))) return rc;
    a.rep = rep;
    a.R = n_seq;
    a.out = d_out;
    rc = eval_gemm(ctx, tables, a, false, s);
    slk_prof_end(ctx, s);
    return rc;
}

// Fused ranking: rank_out[r] = rankdata(-scores(group row_group[r]) with the group's exclusions pushed last)[row_target[r]]
// without a score matrix: the target scores (vector unit), ONE counting sweep of the item table per 64 rows (matrix cores),
// the exclusion lists' corrections (vector unit).  rep / rbias: the groups' representations.
static int rank_fused(slk_ctx *ctx, const slk_tables *tables, int vec, int g, const float *rep, const float *rbias,
                      const int64_t *gmap, const int64_t *d_row_group, const int64_t *d_row_target, int64_t n_rows, const int64_t *d_exc_off,
                      const int64_t *d_exc_items, double *d_rank_out, hipStream_t s) {
    int rc;
    if ((rc = slk_ensure(ctx, ctx->extra[EV_ST], (size_t)n_rows * 4))) return rc;
    if ((rc = slk_ensure(ctx, ctx->extra[EV_CNT], (size_t)n_rows * 16))) return rc;
    float *st = (float *)ctx->extra[EV_ST].p;
    unsigned *gt = (unsigned *)ctx->extra[EV_CNT].p, *eq = gt + n_rows;
    int *dgt = (int *)(eq + n_rows), *deq = dgt + n_rows;
    SLK_HIP(ctx, hipMemsetAsync(gt, 0, (size_t)n_rows * 8, s));
    slk_bloom_dev ibd;
    slk_bloom_to_dev(tables->item_bloom, &ibd);
    const float *V = (const float *)tables->d_param[1], *bi = (const float *)tables->d_param[3];
    const int D = tables->dim;
#define SLK_TGT(V_, G_)                                                                                             \
    hipLaunchKernelGGL((k_rank_target_scores<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n_rows, 256 / G_)), dim3(256), 0, s, rep, \
                       rbias, d_row_group, gmap, V, bi, ibd, D, d_row_target, d_exc_off, d_exc_items, n_rows, st)
    SLK_FOR_LAYOUT(vec, g, SLK_TGT);
#undef SLK_TGT
    SLK_LAUNCH_CHECK(ctx, "k_rank_target_scores");
    slk_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.rep = rep;
    a.rbias = rbias;
    a.rowmap = d_row_group;
    a.gmap = gmap;
    a.R = n_rows;
    a.st = st;
    a.gt = gt;
    a.eq = eq;
    if ((rc = eval_gemm(ctx, tables, a, true, s))) return rc;
    if (d_exc_off) {
#define SLK_EXC(V_, G_)                                                                                             \
    hipLaunchKernelGGL((k_rank_exclusions<V_, G_>), dim3(slk_grid_for(ctx, (size_t)n_rows, 256 / G_)), dim3(256), 0, s, rep,   \
                       rbias, d_row_group, gmap, V, bi, ibd, D, d_exc_off, d_exc_items, (const float *)st, n_rows, dgt, deq)
        SLK_FOR_LAYOUT(vec, g, SLK_EXC);
#undef SLK_EXC
        SLK_LAUNCH_CHECK(ctx, "k_rank_exclusions");
    }
    hipLaunchKernelGGL(k_rank_final, dim3(slk_grid_for(ctx, (size_t)n_rows, 256)), dim3(256), 0, s, (const unsigned *)gt,
                       (const unsigned *)eq, d_exc_off ? (const int *)dgt : (const int *)nullptr,
                       d_exc_off ? (const int *)deq : (const int *)nullptr, n_rows, d_rank_out);
    SLK_LAUNCH_CHECK(ctx, "k_rank_final");
    return SLK_OK;
}

static int check_rank_args(slk_ctx *ctx, const char *who, int64_t n_groups, const void *groups, const int64_t *d_row_group,
                           const int64_t *d_row_target, int64_t n_rows, const int64_t *d_exc_off, const int64_t *d_exc_items,
                           const double *d_rank_out) {
    if (n_groups < 0 || n_rows < 0 || (n_groups > 0 && !groups) || (n_rows > 0 && (!d_row_group || !d_row_target || !d_rank_out)) ||
        (d_exc_off && !d_exc_items && n_groups > 0 && false))
        return slk_fail(ctx, SLK_EINVAL, "%s: bad arguments", who);
    if (n_rows > 0 && n_groups == 0) return slk_fail(ctx, SLK_EINVAL, "%s: rows without groups", who);
    return SLK_OK;
}

SLK_EXPORT int slk_bilinear_rank(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_group_users, int64_t n_groups,
                                 const int64_t *d_row_group, const int64_t *d_row_target, int64_t n_rows,
                                 const int64_t *d_exc_off, const int64_t *d_exc_items, double *d_rank_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, tables, 15u, &vec, &g))) return rc;
    if ((rc = check_rank_args(ctx, "slk_bilinear_rank", n_groups, d_group_users, d_row_group, d_row_target, n_rows, d_exc_off,
                              d_exc_items, d_rank_out)))
        return rc;
    if (n_rows == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    slk_prof_begin(ctx, SLK_K_SCORE, s);
    const float *rep, *rbias;
    const int64_t *gmap;
    if ((rc = eval_user_rows(ctx, tables, vec, g, d_group_users, n_groups, &rep, &rbias, &gmap, s))) return rc;
    rc = rank_fused(ctx, tables, vec, g, rep, rbias, gmap, d_row_group, d_row_target, n_rows, d_exc_off, d_exc_items, d_rank_out, s);
    slk_prof_end(ctx, s);
    return rc;
}

SLK_EXPORT int slk_poolnet_rank(slk_ctx *ctx, const slk_tables *tables, const int64_t *d_group_sequences, int64_t n_groups,
                                int64_t seq_len, const int64_t *d_row_group, const int64_t *d_row_target, int64_t n_rows,
                                const int64_t *d_exc_off, const int64_t *d_exc_items, double *d_rank_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int vec, g, rc;
    if ((rc = slk_check_tables(ctx, tables, 10u, &vec, &g))) return rc;
    if (seq_len < 1) return slk_fail(ctx, SLK_EINVAL, "slk_poolnet_rank: seq_len %lld", (long long)seq_len);
    if ((rc = check_rank_args(ctx, "slk_poolnet_rank", n_groups, d_group_sequences, d_row_group, d_row_target, n_rows, d_exc_off,
                              d_exc_items, d_rank_out)))
        return rc;
    if (n_rows == 0) return SLK_OK;
    SLK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    slk_prof_begin(ctx, SLK_K_SCORE, s);
    float *rep;
    if ((rc = eval_seq_rows(ctx, tables, vec, g, d_group_sequences, n_groups, seq_len, &rep, s))) return rc;
    rc = rank_fused(ctx, tables, vec, g, rep, nullptr, nullptr, d_row_group, d_row_target, n_rows, d_exc_off, d_exc_items, d_rank_out, s);
    slk_prof_end(ctx, s);
    return rc;
}

// predict() of ONE representation against EVERY item (slk_bilinear_predict / slk_poolnet_predict with d_items == NULL):
// the same sweep with a 32-row tile holding one live row
int slk_eval_predict_all(slk_ctx *ctx, const slk_tables *tables, const float *rep, const float *rbias, const int64_t *gmap,
                         float *d_out, hipStream_t s) {
    slk_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.rep = rep;
    a.rbias = rbias;
    a.gmap = gmap;
    a.R = 1;
    a.out = d_out;
    return eval_gemm(ctx, tables, a, false, s);
}

int slk_eval_user_rep(slk_ctx *ctx, const slk_tables *tables, int vec, int g, const int64_t *d_user, const float **rep,
                      const float **rbias, const int64_t **gmap, hipStream_t s) {
    return eval_user_rows(ctx, tables, vec, g, d_user, 1, rep, rbias, gmap, s);
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
