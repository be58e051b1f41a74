// binning.hip -- tile binning: scan, key/value emission, stable radix sort, tile ranges.
//
//   scan_tiles        K2  replaces cub::DeviceScan::InclusiveSum          (reference rasterizer_impl.cu:332)
//   duplicate_keys    K4  replaces duplicateWithKeys                      (reference rasterizer_impl.cu:70-111)
//   sort_pairs        K5  replaces cub::DeviceRadixSort::SortPairs        (reference rasterizer_impl.cu:355-363)
//   tile_ranges       K6  replaces cudaMemset + identifyTileRanges        (reference rasterizer_impl.cu:365-373, 149-171)
//   point_keys        K12 replaces createWithKeys                         (reference rasterizer_impl.cu:113-144)
//
// Contract that matters for parity: keys are (tile << 32) | float_bits(depth); tiles of one
// Gaussian are emitted y-major / x-minor; the sort is STABLE over bits [0, 32 + msb(tiles)), so
// equal (tile, depth) entries keep ascending Gaussian index.
#include "gof_common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace gof {

// rasterizer_impl.cu:35-50
uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

size_t scan_temp_bytes(size_t n)
{
    size_t bytes = 0;
    uint32_t* p = nullptr;
    (void)rocprim::inclusive_scan(nullptr, bytes, p, p, n, rocprim::plus<uint32_t>());
    return bytes;
}

hipError_t scan_tiles(void* tmp, size_t tmp_bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t stream)
{
    return rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::plus<uint32_t>(), stream);
}

size_t sort_temp_bytes(size_t n)
{
    size_t bytes = 0;
    uint64_t* k = nullptr; uint32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, n, 0, 64);
    return bytes;
}

hipError_t sort_pairs(void* tmp, size_t tmp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                      const uint32_t* vals_in, uint32_t* vals_out, size_t n, int end_bit, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, stream);
}

__global__ void __launch_bounds__(256)
duplicate_keys(int P, const SplatRec* __restrict__ rec, const float* __restrict__ depths, const uint32_t* __restrict__ offsets,
               uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, const int32_t* __restrict__ radii, uint32_t gx, uint32_t gy)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const int r = radii[idx];
    if (r > 0) {
        uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
        const float px = rec[idx].f[REC_XY], py = rec[idx].f[REC_XY + 1];
        uint32_t minx, miny, maxx, maxy;
        get_rect(px, py, r, minx, miny, maxx, maxy, gx, gy);
        const uint64_t dbits = __float_as_uint(depths[idx]);
        for (uint32_t y = miny; y < maxy; y++)
            for (uint32_t x = minx; x < maxx; x++) {
                uint64_t key = y * gx + x;
                key <<= 32;
                key |= dbits;
                keys[off] = key;
                vals[off] = (uint32_t)idx;
                off++;
            }
    }
}

__global__ void __launch_bounds__(256)
point_keys(int PN, const float2* __restrict__ points2D, const float* __restrict__ depths, const uint32_t* __restrict__ offsets,
           const uint32_t* __restrict__ tiles_touched, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t gx, uint32_t gy)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= PN) return;
    if (tiles_touched[idx] > 0) {
        const uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
        const float2 p = points2D[idx];
        const int x = (int)min(gx - 1, (uint32_t)max(0, (int)(p.x / TILE_X)));
        const int y = (int)min(gy - 1, (uint32_t)max(0, (int)(p.y / TILE_Y)));
        uint64_t key = (uint64_t)(y * gx + x);
        key <<= 32;
        key |= (uint64_t)__float_as_uint(depths[idx]);
        keys[off] = key;
        vals[off] = (uint32_t)idx;
    }
}

__global__ void __launch_bounds__(256)
tile_ranges(uint32_t L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges)
{
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L) return;
    const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) ranges[currtile].x = 0;
    else {
        const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
        if (currtile != prevtile) {
            ranges[prevtile].y = idx;
            ranges[currtile].x = idx;
        }
    }
    if (idx == L - 1) ranges[currtile].y = L;
}

} // namespace gof
