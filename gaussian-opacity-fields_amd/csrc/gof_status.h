// gof_status.h -- device-side status conventions shared by the binning kernels and the host orchestration.
#pragma once
#include <stdint.h>
namespace gof {
// Instance count reported when the depth sort failed (the bounded look-back poll of a single-kernel radix pass expired, radix.hip):
// gather_rects writes it instead of the real counts, so that the count the host reads back is impossible (real counts are < 2^31)
// and the forward call returns GOF_E_DEVICE instead of rendering from a list that is not sorted.
constexpr uint32_t GOF_SORT_FAILED_COUNT = 0xFFFFFFF0u;
}
