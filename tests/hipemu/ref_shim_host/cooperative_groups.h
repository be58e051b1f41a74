// TEST INFRASTRUCTURE: the handful of cooperative-groups calls the reference makes (forward.cu:313, 434-438; backward.cu:158, 665-669)
#pragma once
#include "cuda_runtime.h"
namespace cooperative_groups {
struct grid_group {
    unsigned long long thread_rank() const
    {
        const dim3 g = gridDim, b = blockDim; const auto bi = blockIdx; const auto t = threadIdx;
        const unsigned long long block = ((unsigned long long)bi.z * g.y + bi.y) * g.x + bi.x;
        return block * ((unsigned long long)b.x * b.y * b.z) + ((unsigned long long)t.z * b.y + t.y) * b.x + t.x;
    }
};
inline grid_group this_grid() { return grid_group(); }
struct thread_block {
    unsigned thread_rank() const { const dim3 b = blockDim; const auto t = threadIdx; return (t.z * b.y + t.y) * b.x + t.x; }
    dim3 group_index() const { const auto bi = blockIdx; return dim3(bi.x, bi.y, bi.z); }
    dim3 thread_index() const { const auto t = threadIdx; return dim3(t.x, t.y, t.z); }
    void sync() const { __syncthreads(); }
};
inline thread_block this_thread_block() { return thread_block(); }
}
