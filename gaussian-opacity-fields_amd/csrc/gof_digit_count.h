// gof_digit_count.h -- digit counting for the radix sort's global histograms (radix.hip: os_hist).
#pragma once
#include "gof_common.h"

namespace gof {

// Digit counting into an LDS histogram (radix.hip: os_hist) with few LDS atomics where the digits of a wave are few: the high bytes of depth keys take a handful of values (sign, exponent),
// those of tile ids one or two -- a same-address ds_add serialises its lanes (~3.5 cycles each).  Up to four groups of equal digits
// are peeled off with one ballot each and counted by one lane; what is left (digits spread over the bins: few conflicts) is counted
// lane by lane.  Called by all 64 lanes of a wave.
__device__ __forceinline__ void os_count(uint32_t* __restrict__ h, uint32_t d, bool valid)
{
    uint64_t rem = __ballot(valid);
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll 1
    for (int g = 0; g < 4 && rem; g++) {
        const int first = (int)__builtin_ctzll(rem);
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
        const uint64_t m = __ballot(valid && d == d0) & rem;
        const uint32_t c = (uint32_t)__popcll(m);
        if (c < 8u) break;                                  // (wave-uniform) not a crowded digit: the rest goes lane by lane
        if (lane == (uint32_t)first) atomicAdd(&h[d0], c);
        rem &= ~m;
    }
    if ((rem >> lane) & 1ull) atomicAdd(&h[d], 1u);
}

} // namespace gof
