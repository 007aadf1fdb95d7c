// slk_sort.hip -- the one place the engine calls a ROCm library: rocPRIM's device radix sort
// (keys: (minibatch, row id) packed in 32 bits; only the bits in use are sorted).
#include <cstring>  // rocPRIM's texture iterator header needs host memset declared first

#include <rocprim/rocprim.hpp>

#include "slk_common.h"

template <class V, class K = uint32_t>
static int sort_impl(slk_ctx *ctx, const K *kin, K *kout, const V *vin, V *vout, size_t n,
                     unsigned end_bit, hipStream_t s, slk_buf *scratch = nullptr) {
    if (n == 0) return SLK_OK;
    if (end_bit > 8 * sizeof(K)) end_bit = 8 * sizeof(K);
    slk_buf &buf = scratch ? *scratch : ctx->sort_tmp;
    size_t tmp = 0;
    SLK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0u, end_bit, s));
    int rc = slk_ensure(ctx, buf, tmp);
    if (rc) return rc;
    SLK_HIP(ctx, rocprim::radix_sort_pairs(buf.p, tmp, kin, kout, vin, vout, n, 0u, end_bit, s));
    return SLK_OK;
}

// same, with the caller's temporary storage: for sorts that may run on another stream than the
// training passes' sorts (the epoch shuffle prepared ahead, slk_shuffle.hip)
int slk_sort_pairs_u32_u32_in(slk_ctx *ctx, slk_buf &scratch, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                              uint32_t *vout, size_t n, unsigned end_bit, hipStream_t s) {
    return sort_impl<uint32_t>(ctx, kin, kout, vin, vout, n, end_bit, s, &scratch);
}

// 64-bit keys (timestamps whose range needs more than 32 bits, slk_seqprep.hip)
int slk_sort_pairs_u64_u32_in(slk_ctx *ctx, slk_buf &scratch, const uint64_t *kin, uint64_t *kout, const uint32_t *vin,
                              uint32_t *vout, size_t n, unsigned end_bit, hipStream_t s) {
    return sort_impl<uint32_t, uint64_t>(ctx, kin, kout, vin, vout, n, end_bit, s, &scratch);
}

int slk_sort_pairs_u32_u32(slk_ctx *ctx, const uint32_t *kin, uint32_t *kout, const uint32_t *vin,
                           uint32_t *vout, size_t n, unsigned end_bit, hipStream_t s) {
    return sort_impl<uint32_t>(ctx, kin, kout, vin, vout, n, end_bit, s);
}

int slk_sort_pairs_u32_u64(slk_ctx *ctx, const uint32_t *kin, uint32_t *kout, const uint64_t *vin,
                           uint64_t *vout, size_t n, unsigned end_bit, hipStream_t s) {
    // (a 9-bit onesweep configuration -- 27 key bits in 3 iterations instead of 4 -- was measured and not kept:
    // profiles/README.md, round 1)
    return sort_impl<uint64_t>(ctx, kin, kout, vin, vout, n, end_bit, s);
}

// Sizes the temporary storage for pair sorts of up to n elements (both value widths).
int slk_sort_reserve(slk_ctx *ctx, size_t n) {
    if (n == 0) return SLK_OK;
    size_t t32 = 0, t64 = 0;
    SLK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, t32, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (const uint32_t *)nullptr, (uint32_t *)nullptr, n, 0u, 32u, (hipStream_t)0));
    SLK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, t64, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (const uint64_t *)nullptr, (uint64_t *)nullptr, n, 0u, 32u, (hipStream_t)0));
    return slk_ensure(ctx, ctx->sort_tmp, t32 > t64 ? t32 : t64);
}
