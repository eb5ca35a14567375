// TEST INFRASTRUCTURE ONLY — host stand-in for the one hipcub call the product uses (see ../hip/hip_runtime.h).
#pragma once
#include <hip/hip_runtime.h>

#include <numeric>

namespace hipcub {
struct DeviceRadixSort {
    // stable LSD radix sort semantics: pairs ordered by key bits [begin_bit, end_bit), ties keep input order
    template <class K, class V>
    static hipError_t SortPairs(void* tmp, size_t& tmp_bytes, const K* keys_in, K* keys_out, const V* vals_in, V* vals_out, int n,
                                int begin_bit = 0, int end_bit = (int)sizeof(K) * 8, hipStream_t = nullptr) {
        if (!tmp) {
            tmp_bytes = 256;
            return hipSuccess;
        }
        const int nb = end_bit - begin_bit;
        const K mask = nb >= (int)sizeof(K) * 8 ? ~K(0) : (K)(((K(1) << nb) - 1) << begin_bit);
        std::vector<int> order(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (keys_in[a] & mask) < (keys_in[b] & mask); });
        for (int i = 0; i < n; ++i) {
            keys_out[i] = keys_in[order[i]];
            vals_out[i] = vals_in[order[i]];
        }
        return hipSuccess;
    }
};
}  // namespace hipcub
