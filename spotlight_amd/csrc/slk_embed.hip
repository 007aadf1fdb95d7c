// slk_embed.hip -- the embedding front-end for encoders whose body stays on stock PyTorch-ROCm / MIOpen
// (LSTMNet, CNNNet, MixtureLSTMNet: spotlight/sequence/representations.py:147-596).
//
// forward : out[k][:] = W[ids[k]][:]  (ScaledEmbedding / ZeroEmbedding, layers.py:23-56), or the sum of the
//           H hashed rows of a BloomEmbedding (layers.py:236-242), hashes computed in-kernel.
// backward: the gradient w.r.t. W of that gather, given dL/dout.  Occurrences are sorted by table row with a
//           stable radix sort and every distinct row receives the sum of its gradient rows through a chunked
//           multi-level segmented reduction (k_emb_reduce) -- no atomics, bit-reproducible, balanced under
//           skew.  Rows equal to padding_idx get no gradient, like nn.Embedding.
//           Output either dense [rows, D] (zero where untouched) or coalesced COO (distinct rows + sums)
//           for sparse=True layers feeding SparseAdam / sparse Adagrad.
#include "slk_common.h"
#include "slk_kernels.h"

enum { EM_SORT = 32, EM_KEY0 = 40, EM_KEY1, EM_PAY0, EM_PAY1, EM_HEADS, EM_PART0, EM_PART1 };  // ctx->extra slots

static int em_threads_per_row(int D, int VEC) {
    int need = (D + VEC - 1) / VEC, t = 1;
    while (t < need && t < 64) t <<= 1;
    return t;
}

static int em_grid(slk_ctx *ctx, uint64_t groups, int groups_per_block) {
    uint64_t b = (groups + groups_per_block - 1) / groups_per_block;
    uint64_t cap = (uint64_t)ctx->num_cus * 32;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

template <int VEC>
__global__ __launch_bounds__(256) void k_emb_gather(const float *W, slk_bloom_dev b, const int64_t *ids, size_t n, int D, int T,
                                                     float *out) {
    const int lane = threadIdx.x % T;
    const size_t grp = threadIdx.x / T, gpb = 256 / T;
    for (size_t k = blockIdx.x * gpb + grp; k < n; k += (size_t)gridDim.x * gpb) {
        const uint32_t id = (uint32_t)ids[k];
        for (int d0 = lane * VEC; d0 < D; d0 += T * VEC)
            slk_vstore<VEC>(out + k * D + d0, slk_emb_vec<VEC>(W, b, id, D, d0, true));
    }
}

// one (row key, occurrence) pair per looked-up table row; rows that receive no gradient get the key `rows`
__global__ __launch_bounds__(256) void k_emb_keys(slk_bloom_dev b, uint32_t rows, uint32_t skip_row, const int64_t *ids, size_t n,
                                                   uint32_t *key, uint32_t *pay) {
    const int H = b.n_hash ? b.n_hash : 1;
    const size_t total = n * (size_t)H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t k = i / H;
        const int h = (int)(i - k * H);
        const uint32_t id = (uint32_t)ids[k];
        uint32_t row = b.n_hash ? slk_bloom_row(b, id, h) : id;
        if (row == skip_row || row >= rows) row = rows;
        key[i] = row;
        pay[i] = (uint32_t)k;
    }
}

// Segmented sums over the row-sorted occurrences, balanced for skewed ids (one hot item can own a sixth of a
// Zipf batch): the sorted list is cut into chunks of EM_CHUNK consecutive occurrences, one lane group per chunk.
// A run of equal rows that lies strictly inside a chunk is summed and written to its destination; the chunk's
// first and last run may continue in the neighbouring chunks, so their partial sums go, with their row, into
// entries 2c and 2c + 1 of the next level's list -- which is again sorted by row and is reduced by the same
// kernel, EM_CHUNK / 2 times shorter per level, until one chunk remains.  Summation order is fixed by the
// positions alone: bit-reproducible, no atomics.
#define EM_CHUNK 16
// `key` is the destination row of `dst` ([rows][D]): the table row for the dense gradient, the segment index
// for the COO values; keys >= rows (the run of lookups without a gradient) are dropped.
template <int VEC>
__global__ __launch_bounds__(256) void k_emb_reduce(const uint32_t *key, const uint32_t *pay, uint32_t n, uint32_t nchunks,
                                                     uint32_t rows, int D, int T, const float *gin, float *dst,
                                                     uint32_t *key_next, float *part_next) {
    const int lane = threadIdx.x % T;
    const uint32_t grp = threadIdx.x / T, gpb = 256 / T;
    for (uint32_t c = blockIdx.x * gpb + grp; c < nchunks; c += gridDim.x * gpb) {
        const uint32_t start = c * EM_CHUNK;
        const uint32_t end = start + EM_CHUNK < n ? start + EM_CHUNK : n;
        for (int d0 = lane * VEC; d0 < D; d0 += T * VEC) {
            uint32_t row = key[start];
            bool first = true;
            slk_vec<VEC> acc = slk_vzero<VEC>();
            for (uint32_t q = start; q <= end; ++q) {
                const uint32_t r = q < end ? key[q] : 0xffffffffu;
                if (r != row) {  // the run of `row` ends at q
                    const bool last = q == end;
                    if (nchunks == 1 || (!first && !last)) {
                        if (row < rows) slk_vstore<VEC>(dst + (size_t)row * D + d0, acc);
                    } else {
                        const uint32_t slot = 2 * c + (first ? 0u : 1u);
                        slk_vstore<VEC>(part_next + (size_t)slot * D + d0, acc);
                        if (d0 == 0) key_next[slot] = row;
                        if (first && last) {  // one run fills the chunk: the second entry is an exact zero
                            slk_vstore<VEC>(part_next + (size_t)(slot + 1) * D + d0, slk_vzero<VEC>());
                            if (d0 == 0) key_next[slot + 1] = row;
                        }
                    }
                    first = false;
                    row = r;
                    acc = slk_vzero<VEC>();
                }
                if (q < end) {
                    const slk_vec<VEC> x = slk_vload<VEC>(gin + (size_t)(pay ? pay[q] : q) * D + d0);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc.v[i] += x.v[i];
                }
            }
        }
    }
}

// rows_out[s] = the row of segment s (ascending): the indices of the coalesced COO gradient
__global__ __launch_bounds__(256) void k_emb_segment_rows(const uint32_t *key, const uint32_t *heads, uint32_t n_rows,
                                                           int64_t *rows_out) {
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < n_rows; s += gridDim.x * 256) rows_out[s] = (int64_t)key[heads[s]];
}

static int em_check(slk_ctx *ctx, const char *who, int64_t rows, int32_t D, const slk_bloom *bloom, int64_t n) {
    if (rows < 1 || rows >= ((int64_t)1 << 31) || D < 1 || D > 4096)
        return slk_fail(ctx, SLK_EINVAL, "%s: rows = %lld, dim = %d out of range", who, (long long)rows, D);
    if (n < 0 || n * (bloom && bloom->n_hash ? bloom->n_hash : 1) >= ((int64_t)1 << 31))
        return slk_fail(ctx, SLK_EINVAL, "%s: %lld lookups out of range", who, (long long)n);
    if (bloom && bloom->n_hash) {
        if (bloom->n_hash < 1 || bloom->n_hash > 8 || bloom->rows != rows)
            return slk_fail(ctx, SLK_EINVAL, "%s: bloom descriptor does not match the table (rows %lld vs %lld, %d hashes)",
                            who, (long long)bloom->rows, (long long)rows, bloom->n_hash);
    }
    return SLK_OK;
}

SLK_EXPORT int slk_embedding_forward(slk_ctx *ctx, const float *d_weight, int64_t rows, int32_t dim, const slk_bloom *bloom,
                                     const int64_t *d_ids, int64_t n, float *d_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    int rc = em_check(ctx, "slk_embedding_forward", rows, dim, bloom, n);
    if (rc) return rc;
    if (n == 0) return SLK_OK;
    if (!d_weight || !d_ids || !d_out) return slk_fail(ctx, SLK_EINVAL, "slk_embedding_forward: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    slk_bloom_dev b;
    slk_bloom_to_dev(bloom && bloom->n_hash ? bloom : nullptr, &b);
    slk_prof_begin(ctx, SLK_K_SCORE, s);
    if (dim % 4 == 0) {
        const int T = em_threads_per_row(dim, 4);
        hipLaunchKernelGGL(k_emb_gather<4>, dim3(em_grid(ctx, (uint64_t)n, 256 / T)), dim3(256), 0, s, d_weight, b, d_ids, (size_t)n,
                           (int)dim, T, d_out);
    } else {
        const int T = em_threads_per_row(dim, 1);
        hipLaunchKernelGGL(k_emb_gather<1>, dim3(em_grid(ctx, (uint64_t)n, 256 / T)), dim3(256), 0, s, d_weight, b, d_ids, (size_t)n,
                           (int)dim, T, d_out);
    }
    SLK_LAUNCH_CHECK(ctx, "k_emb_gather");
    slk_prof_end(ctx, s);
    return SLK_OK;
}

SLK_EXPORT int slk_embedding_backward_plan(slk_ctx *ctx, int64_t rows, int32_t dim, const slk_bloom *bloom, int64_t padding_idx,
                                           const int64_t *d_ids, int64_t n, int64_t *num_rows_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    ctx->em_occ = -1;
    int rc = em_check(ctx, "slk_embedding_backward_plan", rows, dim, bloom, n);
    if (rc) return rc;
    if (n > 0 && !d_ids) return slk_fail(ctx, SLK_EINVAL, "slk_embedding_backward_plan: NULL ids");
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const int H = bloom && bloom->n_hash ? bloom->n_hash : 1;
    const size_t occ = (size_t)n * H;
    ctx->em_rows = rows;
    ctx->em_dim = dim;
    ctx->em_segments = -1;
    if (num_rows_out) *num_rows_out = 0;
    if (occ == 0) {
        ctx->em_occ = 0;
        ctx->em_segments = 0;
        return SLK_OK;
    }
    slk_buf *E = ctx->extra;
    for (int b = 0; b < 4; ++b)
        if ((rc = slk_ensure(ctx, E[EM_KEY0 + b], occ * 4))) return rc;
    uint32_t *k0 = (uint32_t *)E[EM_KEY0].p, *k1 = (uint32_t *)E[EM_KEY1].p;
    uint32_t *p0 = (uint32_t *)E[EM_PAY0].p, *p1 = (uint32_t *)E[EM_PAY1].p;
    slk_bloom_dev b;
    slk_bloom_to_dev(bloom && bloom->n_hash ? bloom : nullptr, &b);
    // a BloomEmbedding's inner table is an nn.Embedding with padding_idx = skip_row: that ROW is skipped
    const int64_t skip = (bloom && bloom->n_hash) ? bloom->skip_row : padding_idx;
    const uint32_t skip_row = (skip < 0 || skip >= rows) ? 0xffffffffu : (uint32_t)skip;
    slk_prof_begin(ctx, SLK_K_PREP, s);
    hipLaunchKernelGGL(k_emb_keys, dim3(em_grid(ctx, occ, 256)), dim3(256), 0, s, b, (uint32_t)rows, skip_row, d_ids, (size_t)n, k0, p0);
    SLK_LAUNCH_CHECK(ctx, "k_emb_keys");
    unsigned bits = 0;
    for (uint64_t r = (uint64_t)rows; r; r >>= 1) ++bits;  // keys go up to `rows` inclusive
    if ((rc = slk_sort_pairs_u32_u32_in(ctx, E[EM_SORT], k0, k1, p0, p1, occ, bits, s, true))) return rc;
    if (num_rows_out) {
        if ((rc = slk_ensure(ctx, E[EM_HEADS], (occ + 1) * 4))) return rc;
        uint32_t nseg = 0, last = 0;
        // segment index of every sorted lookup -> EM_PAY0 (the sort's input payload, free by now): the key the
        // reduction files the COO values under
        if ((rc = slk_compact_heads(ctx, k1, (uint32_t)occ, (uint32_t *)E[EM_HEADS].p, p0, &nseg, s))) return rc;
        SLK_HIP(ctx, hipMemcpyAsync(&last, k1 + occ - 1, 4, hipMemcpyDeviceToHost, s));
        SLK_HIP(ctx, hipStreamSynchronize(s));
        if (last >= (uint32_t)rows) --nseg;  // the trailing run of occurrences without a gradient
        ctx->em_segments = nseg;
        *num_rows_out = nseg;
    }
    slk_prof_end(ctx, s);
    ctx->em_occ = (int64_t)occ;
    return SLK_OK;
}

SLK_EXPORT int slk_embedding_backward_fill(slk_ctx *ctx, const float *d_grad_out, float *d_grad_dense, int64_t *d_rows_out,
                                           float *d_values_out, void *stream) {
    if (!ctx) return SLK_EINVAL;
    if (ctx->em_occ < 0) return slk_fail(ctx, SLK_EINVAL, "slk_embedding_backward_fill: no plan (call slk_embedding_backward_plan first)");
    const bool sparse = d_grad_dense == nullptr;
    if (sparse && ctx->em_segments < 0)
        return slk_fail(ctx, SLK_EINVAL, "slk_embedding_backward_fill: the plan did not count rows (pass num_rows_out) but no dense output was given");
    hipStream_t s = (hipStream_t)stream;
    ctx->last_stream = s;
    const int64_t occ = ctx->em_occ, rows = ctx->em_rows;
    const int D = ctx->em_dim;
    ctx->em_occ = -1;
    if (!sparse) SLK_HIP(ctx, hipMemsetAsync(d_grad_dense, 0, (size_t)rows * D * sizeof(float), s));
    if (occ == 0 || (sparse && ctx->em_segments == 0)) return SLK_OK;
    if (!d_grad_out || (sparse && (!d_rows_out || !d_values_out)))
        return slk_fail(ctx, SLK_EINVAL, "slk_embedding_backward_fill: NULL argument");
    slk_buf *E = ctx->extra;
    const uint32_t n_rows = sparse ? (uint32_t)ctx->em_segments : 0u;
    const int VEC = D % 4 == 0 ? 4 : 1;
    const int T = em_threads_per_row(D, VEC);
    int rc;
    const uint32_t chunks1 = ((uint32_t)occ + EM_CHUNK - 1) / EM_CHUNK;
    for (int b = 0; b < 2; ++b)
        if ((rc = slk_ensure(ctx, E[EM_PART0 + b], (size_t)2 * (b ? (chunks1 + EM_CHUNK - 1) / EM_CHUNK * 2 : chunks1) * D * sizeof(float) + 64)))
            return rc;
    slk_prof_begin(ctx, SLK_K_ITEM_PASS, s);
    if (sparse) {
        hipLaunchKernelGGL(k_emb_segment_rows, dim3(em_grid(ctx, n_rows, 256)), dim3(256), 0, s, (const uint32_t *)E[EM_KEY1].p,
                           (const uint32_t *)E[EM_HEADS].p, n_rows, d_rows_out);
        SLK_LAUNCH_CHECK(ctx, "k_emb_segment_rows");
    }
    // level 1 reads the sorted (row | segment, lookup) pairs and the caller's gradient rows; deeper levels read
    // the previous level's (key, partial sum) entries.  Key buffers: the sort's input arrays are free by now
    // (EM_PAY0 holds the segment ids only until level 1 has consumed them).
    const uint32_t *key = (const uint32_t *)E[sparse ? EM_PAY0 : EM_KEY1].p, *pay = (const uint32_t *)E[EM_PAY1].p;
    const float *gin = d_grad_out;
    float *dst = sparse ? d_values_out : d_grad_dense;
    const uint32_t bound = sparse ? n_rows : (uint32_t)rows;
    uint32_t n = (uint32_t)occ;
    for (int level = 0;; ++level) {
        const uint32_t nchunks = (n + EM_CHUNK - 1) / EM_CHUNK;
        uint32_t *key_next = (uint32_t *)E[(level & 1) ? EM_PAY0 : EM_KEY0].p;
        float *part_next = (float *)E[EM_PART0 + (level & 1)].p;
        const dim3 grid(em_grid(ctx, nchunks, 256 / T));
        if (VEC == 4)
            hipLaunchKernelGGL(k_emb_reduce<4>, grid, dim3(256), 0, s, key, pay, n, nchunks, bound, D, T, gin, dst, key_next, part_next);
        else
            hipLaunchKernelGGL(k_emb_reduce<1>, grid, dim3(256), 0, s, key, pay, n, nchunks, bound, D, T, gin, dst, key_next, part_next);
        if (nchunks == 1) break;
        key = key_next;
        pay = nullptr;
        gin = part_next;
        n = 2 * nchunks;
    }
    SLK_LAUNCH_CHECK(ctx, "k_emb_reduce");
    slk_prof_end(ctx, s);
    return SLK_OK;
}
