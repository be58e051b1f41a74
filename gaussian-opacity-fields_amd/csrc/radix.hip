// radix.hip -- hand-written device-wide primitives for the binning stage (gfx950, wave64):
//   * exclusive / inclusive scan of u32 (3 kernels: block sums, scan of sums, apply)
//   * stable LSD radix sort of (u32 key, u32 value) pairs with 8-bit digits
//
// They replace cub::DeviceScan::InclusiveSum and cub::DeviceRadixSort::SortPairs of the reference
// (reference rasterizer_impl.cu:332, 355-363).  The reference sorts R (tile << 32 | depth) 64-bit keys
// over 32 + log2(tiles) bits (6 passes at 1600x1063); binning.hip instead sorts the P Gaussians by depth
// once and then only the tile bits of the R instances (2 passes), which yields the identical order.
//
// Sort design: a 256-thread block owns RS_BLOCK consecutive items, each of its 4 waves a contiguous quarter.
//   rs_hist    -- per-block digit histogram (wave-level multisplit: 8 ballots give every lane the mask of
//                 lanes holding the same digit; the lowest such lane adds the group size to an LDS counter)
//   (scan)     -- exclusive scan of the flat [digit][block] histogram = global offset of every (digit, block)
//   rs_scatter -- keys/values stay in registers; a lane's rank is (per-wave LDS cursor of its digit) + (number
//                 of same-digit lanes below it) -- a STABLE rank because waves own ascending quarters and visit
//                 them in index order; the block exchanges the items through LDS into digit-sorted order and
//                 writes each digit's run contiguously at its global offset (coalesced).
#include "gof_common.h"

namespace gof {

constexpr int SCAN_ITEMS = 16;                       // per thread
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;         // 4096 items per block
#ifndef GOF_RS_CHUNK
#define GOF_RS_CHUNK 1024
#endif
constexpr int RS_CHUNK = GOF_RS_CHUNK;               // items per wave (unit) in the radix sort
constexpr int RS_DIGITS = 256;

// ---------------------------------------------------------------------------------------------------
// scan
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(v, o);
        if ((int)(threadIdx.x & 63) >= o) v += n;
    }
    return v;
}
// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total, uint32_t* s_wave /*[4]*/)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t s = s_wave[w]; if ((uint32_t)w < wave) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(256)
scan_block_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += in[base + k];
    uint32_t total;
    block_exclusive_scan(s, &total, s_wave);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// single block: exclusive scan of sums[0..nb) in place; sums[nb] = grand total
__global__ void __launch_bounds__(256)
scan_sums(uint32_t* __restrict__ sums, uint32_t nb)
{
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (uint32_t c = 0; c < nb; c += 256) {
        const uint32_t i = c + threadIdx.x;
        const uint32_t v = i < nb ? sums[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, &total, s_wave);
        if (i < nb) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) sums[nb] = carry;
}
// out[i] = (inclusive ? in[0..i] : in[0..i)) summed; optional gather: in[idx[i]] instead of in[i]
template <bool INCLUSIVE, bool GATHER>
__global__ void __launch_bounds__(256)
scan_apply(const uint32_t* __restrict__ in, const uint32_t* __restrict__ idx, uint32_t n, const uint32_t* __restrict__ sums,
           uint32_t* __restrict__ out)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = 0;
        if (base + k < n) v[k] = GATHER ? in[idx[base + k]] : in[base + k];
        s += v[k];
    }
    uint32_t total;
    uint32_t run = sums[blockIdx.x] + block_exclusive_scan(s, &total, s_wave);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = INCLUSIVE ? run + v[k] : run;
        run += v[k];
    }
}
template <bool GATHER>
__global__ void __launch_bounds__(256)
scan_block_sums_gather(const uint32_t* __restrict__ in, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += GATHER ? in[idx[base + k]] : in[base + k];
    uint32_t total;
    block_exclusive_scan(s, &total, s_wave);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

size_t scan_tmp_words(size_t n) { return (n + SCAN_BLOCK - 1) / SCAN_BLOCK + 2; }

// out = scan(in[idx]) if idx != nullptr else scan(in).  tmp: scan_tmp_words(n) u32.  The grand total is left in
// tmp[nblocks] (device); total_dev_out (optional) receives its address.
hipError_t device_scan_u32(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool inclusive, uint32_t* tmp,
                           const uint32_t** total_dev_out, hipStream_t stream)
{
    const uint32_t nb = (uint32_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    if (total_dev_out) *total_dev_out = tmp + nb;
    if (n == 0) return hipMemsetAsync(tmp, 0, 2 * sizeof(uint32_t), stream);
    if (idx) hipLaunchKernelGGL(scan_block_sums_gather<true>, dim3(nb), dim3(256), 0, stream, in, idx, (uint32_t)n, tmp);
    else hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(256), 0, stream, in, (uint32_t)n, tmp);
    hipLaunchKernelGGL(scan_sums, dim3(1), dim3(256), 0, stream, tmp, nb);
    if (idx) {
        if (inclusive) hipLaunchKernelGGL((scan_apply<true, true>), dim3(nb), dim3(256), 0, stream, in, idx, (uint32_t)n, tmp, out);
        else hipLaunchKernelGGL((scan_apply<false, true>), dim3(nb), dim3(256), 0, stream, in, idx, (uint32_t)n, tmp, out);
    } else {
        if (inclusive) hipLaunchKernelGGL((scan_apply<true, false>), dim3(nb), dim3(256), 0, stream, in, idx, (uint32_t)n, tmp, out);
        else hipLaunchKernelGGL((scan_apply<false, false>), dim3(nb), dim3(256), 0, stream, in, idx, (uint32_t)n, tmp, out);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------------------------------
// mask of the lanes (among `valid`) whose 8-bit digit equals this lane's digit
__device__ __forceinline__ uint64_t match_digit(uint32_t d, uint64_t valid)
{
    uint64_t peers = valid;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    return peers;
}

// ---- block kernels: 256 threads, RS_BLOCK items; wave w owns the contiguous quarter [w*RS_BLOCK/4, ...) ----
constexpr int RS_STEPS = RS_CHUNK / 64;              // steps of 64 items per wave
constexpr int RS_BLOCK = 4 * RS_CHUNK;               // items per block

__global__ void __launch_bounds__(256)
rs_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ hist, uint32_t nblocks)
{
    __shared__ uint32_t s_cnt[4][RS_DIGITS];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; k++) s_cnt[wave][lane + 64 * k] = 0;
    const uint32_t begin = blockIdx.x * RS_BLOCK + wave * RS_CHUNK;
#pragma unroll 4
    for (int s = 0; s < RS_STEPS; s++) {
        const uint32_t i = begin + s * 64 + lane;
        const bool ok = i < n;
        const uint64_t valid = __ballot(ok);
        if (valid == 0ull) break;
        const uint32_t d = ok ? ((keys[i] >> shift) & 0xFFu) : 0u;
        const uint64_t peers = match_digit(d, valid);
        if (ok && (uint32_t)(__ffsll((long long)peers) - 1) == lane) s_cnt[wave][d] += (uint32_t)__popcll(peers);
    }
    __syncthreads();
    const uint32_t d = threadIdx.x;
    hist[(size_t)d * nblocks + blockIdx.x] = s_cnt[0][d] + s_cnt[1][d] + s_cnt[2][d] + s_cnt[3][d];
}

// Stable scatter of one block: keys/values are ranked in registers (wave-level multisplit with per-wave LDS
// cursors), exchanged through LDS into digit-sorted order and written out as contiguous runs per digit.
__global__ void __launch_bounds__(256)
rs_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
           uint32_t* __restrict__ vals_out, uint32_t n, int shift, const uint32_t* __restrict__ offs, uint32_t nblocks)
{
    __shared__ uint32_t s_cur[4][RS_DIGITS];     // per-wave digit counts, then running cursors
    __shared__ uint32_t s_bstart[RS_DIGITS];     // block-local start of every digit in sorted order
    __shared__ uint32_t s_gbase[RS_DIGITS];      // global start of this block's run of every digit
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_key[RS_BLOCK];
    __shared__ uint32_t s_val[RS_BLOCK];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < 4; k++) s_cur[wave][lane + 64 * k] = 0;
    const uint32_t blk_begin = blockIdx.x * RS_BLOCK;
    const uint32_t begin = blk_begin + wave * RS_CHUNK;

    uint32_t key[RS_STEPS], val[RS_STEPS];
    uint16_t rnk[RS_STEPS], cnt[RS_STEPS];       // rank among same-digit lanes of the step; group size (leader only)
#pragma unroll
    for (int s = 0; s < RS_STEPS; s++) {
        const uint32_t i = begin + s * 64 + lane;
        key[s] = 0xFFFFFFFFu; val[s] = 0;
        if (i < n) { key[s] = keys_in[i]; val[s] = vals_in[i]; }
    }
#pragma unroll
    for (int s = 0; s < RS_STEPS; s++) {
        const uint32_t i = begin + s * 64 + lane;
        const bool ok = i < n;
        const uint64_t valid = __ballot(ok);
        const uint32_t d = (key[s] >> shift) & 0xFFu;
        const uint64_t peers = match_digit(d, valid);
        const uint32_t r = (uint32_t)__popcll(peers & lt);
        const uint32_t c = (uint32_t)__popcll(peers);
        rnk[s] = (uint16_t)r;
        cnt[s] = (uint16_t)((ok && r == 0) ? c : 0);
        if (ok && r == 0) s_cur[wave][d] += c;
    }
    __syncthreads();
    {   // digit d = threadIdx.x: wave bases (exclusive over waves), block start (exclusive over digits), global base
        const uint32_t d = threadIdx.x;
        const uint32_t c0 = s_cur[0][d], c1 = s_cur[1][d], c2 = s_cur[2][d], c3 = s_cur[3][d];
        const uint32_t tot = c0 + c1 + c2 + c3;
        uint32_t blk_total;
        const uint32_t start = block_exclusive_scan(tot, &blk_total, s_scan);
        s_bstart[d] = start;
        s_gbase[d] = offs[(size_t)d * nblocks + blockIdx.x];
        s_cur[0][d] = start; s_cur[1][d] = start + c0; s_cur[2][d] = start + c0 + c1; s_cur[3][d] = start + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RS_STEPS; s++) {
        const uint32_t i = begin + s * 64 + lane;
        const bool ok = i < n;
        const uint32_t d = (key[s] >> shift) & 0xFFu;
        const uint32_t base = ok ? s_cur[wave][d] : 0u;              // all lanes of a digit group read the same cursor
        if (cnt[s]) s_cur[wave][d] = base + cnt[s];                 // the group's lowest lane advances it
        if (ok) { s_key[base + rnk[s]] = key[s]; s_val[base + rnk[s]] = val[s]; }
    }
    __syncthreads();
    const uint32_t blk_n = min((uint32_t)RS_BLOCK, n - blk_begin);
#pragma unroll
    for (int j = 0; j < RS_BLOCK / 256; j++) {
        const uint32_t p = j * 256 + threadIdx.x;
        if (p < blk_n) {
            const uint32_t k = s_key[p];
            const uint32_t d = (k >> shift) & 0xFFu;
            const uint32_t g = s_gbase[d] + (p - s_bstart[d]);
            keys_out[g] = k;
            vals_out[g] = s_val[p];
        }
    }
}

uint32_t rs_units(size_t n) { return (uint32_t)((n + RS_BLOCK - 1) / RS_BLOCK); }   // blocks
// u32 words of scratch: [256 * units] histogram + scan scratch
size_t rs_tmp_words(size_t n) { const size_t h = (size_t)RS_DIGITS * rs_units(n); return h + scan_tmp_words(h) + 64; }

// Stable sort of (key, value) pairs on key bits [0, end_bit), 8 bits per pass.  Buffers a* hold the input; the
// result ends up in (*keys_res, *vals_res), which alias either a* or b*.
hipError_t radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int end_bit,
                                uint32_t* tmp, uint32_t** keys_res, uint32_t** vals_res, hipStream_t stream)
{
    uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
    if (n > 0) {
        const uint32_t nunits = rs_units(n);
        const size_t hwords = (size_t)RS_DIGITS * nunits;
        uint32_t* hist = tmp;
        uint32_t* scan_tmp = tmp + hwords;
        const dim3 grid(nunits), block(256);
        for (int shift = 0; shift < end_bit; shift += 8) {
            hipLaunchKernelGGL(rs_hist, grid, block, 0, stream, ki, (uint32_t)n, shift, hist, nunits);
            hipError_t e = device_scan_u32(hist, nullptr, hist, hwords, false, scan_tmp, nullptr, stream);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(rs_scatter, grid, block, 0, stream, ki, vi, ko, vo, (uint32_t)n, shift, hist, nunits);
            uint32_t* t;
            t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
        }
    }
    *keys_res = ki;
    *vals_res = vi;
    return hipGetLastError();
}
int radix_passes(int end_bit) { return (end_bit + 7) / 8; }

} // namespace gof
