// tests/emu/rocprim/rocprim.hpp -- TEST HARNESS ONLY.  Host stand-in for the one rocPRIM
// entry point the engine uses (device radix_sort_pairs), for the fiber emulator build.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {
template <class Key, class Value, class Size>
hipError_t radix_sort_pairs(void *temporary_storage, size_t &storage_size, const Key *keys_input,
                            Key *keys_output, const Value *values_input, Value *values_output,
                            Size size, unsigned begin_bit = 0, unsigned end_bit = 8 * sizeof(Key),
                            hipStream_t = 0, bool = false) {
    if (temporary_storage == nullptr) {
        storage_size = 64;
        return hipSuccess;
    }
    const Key mask = (end_bit - begin_bit >= 8 * sizeof(Key))
                         ? ~Key(0)
                         : (Key)(((Key(1) << (end_bit - begin_bit)) - 1) << begin_bit);
    std::vector<size_t> idx((size_t)size);
    std::iota(idx.begin(), idx.end(), size_t(0));
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
        return (keys_input[a] & mask) < (keys_input[b] & mask);
    });
    for (size_t i = 0; i < (size_t)size; ++i) {
        keys_output[i] = keys_input[idx[i]];
        values_output[i] = values_input[idx[i]];
    }
    return hipSuccess;
}
}  // namespace rocprim
