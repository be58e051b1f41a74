// TEST INFRASTRUCTURE: the two CUB device-wide calls of the reference (rasterizer_impl.cu:200-260, 332, 358) as plain host code with
// CUB's two-phase temporary-storage protocol: an inclusive sum, and a STABLE sort of (key, value) pairs on key bits [begin, end).
#pragma once
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>
#include "../cuda_runtime.h"
namespace cub {
struct DeviceScan {
    template <class In, class Out>
    static cudaError_t InclusiveSum(void* tmp, size_t& bytes, In in, Out out, int n)
    {
        if (!tmp) { bytes = 16; return cudaSuccess; }
        typename std::remove_reference<decltype(out[0])>::type run = 0;
        for (int i = 0; i < n; i++) { run += in[i]; out[i] = run; }
        return cudaSuccess;
    }
};
struct DeviceReduce {
    template <class In, class Out, class Op, class T>
    static cudaError_t Reduce(void* tmp, size_t& bytes, In in, Out out, int n, Op op, T init)
    {
        if (!tmp) { bytes = 16; return cudaSuccess; }
        T acc = init;
        for (int i = 0; i < n; i++) acc = op(acc, in[i]);
        *out = acc;
        return cudaSuccess;
    }
};
struct DeviceRadixSort {
    template <class K, class V>
    static cudaError_t SortPairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, int n, int begin_bit = 0, int end_bit = (int)sizeof(K) * 8)
    {
        if (!tmp) { bytes = 16; return cudaSuccess; }
        const K mask = (end_bit - begin_bit >= (int)sizeof(K) * 8) ? ~K(0) : (((K(1) << (end_bit - begin_bit)) - 1) << begin_bit);
        std::vector<int> idx((size_t)n);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return (kin[a] & mask) < (kin[b] & mask); });
        for (int i = 0; i < n; i++) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
        return cudaSuccess;
    }
};
}
