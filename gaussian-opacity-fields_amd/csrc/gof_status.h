// gof_status.h -- device-side status conventions shared by the binning kernels and the host orchestration.
#pragma once
#include <stdint.h>
namespace gof {
// Instance count reported when the depth sort failed (the bounded look-back poll of a single-kernel radix pass expired, radix.hip):
// gather_rects writes it instead of the real counts, so that the count the host reads back is impossible (real counts are < 2^31)
// and the forward call returns GOF_E_DEVICE instead of rendering from a list that is not sorted.
constexpr uint32_t GOF_SORT_FAILED_COUNT = 0xFFFFFFF0u;

// The item count of a launch sized for a CAPACITY (sync-free forward, api.hip: gof_forward_fused): the scanned instance count on
// the device, clamped to the capacity -- and 0 when the depth sort failed: the kernels queued behind the scan (tile sort, tile
// ranges, and through the zeroed ranges the blend) must then not walk workspace nobody wrote (round-2 review).
__device__ __forceinline__ uint32_t device_item_count(uint32_t capacity, const uint32_t* __restrict__ n_dev)
{
    if (!n_dev) return capacity;
    const uint32_t c = *n_dev;
    return c >= GOF_SORT_FAILED_COUNT ? 0u : (c < capacity ? c : capacity);
}
}
