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
#include "gof_status.h"

namespace gof {

// items per thread of the scan kernels: 16 (4096 per workgroup), or -- inputs of up to 2 M words: the radix sorts' [digit][block]
// histograms of a 6M-Gaussian frame are 0.4 M (depth sort) and 1.4 M (tile sort) words -- 4 (1024 per workgroup: four times the
// workgroups, a fourth of the dependent loads per thread; the scans of one 6M-Gaussian frame took 0.33 ms of its 4.1 in tiles of 4096 --
// profiles/r06_large_p_6M_kernel_stats.md.  Measured, interleaved, tiles of 1024 / 2048 / 4096 words: S1M 2.386 / 2.397 / 2.400 ms per step,
// clustered 3.226 / 3.238 / 3.248, 6 M Gaussians 4.131 / 4.144 / 4.157 -- profiles/r06_ab_call5_hist_scan.txt)
#ifndef GOF_SCAN_SMALL_ITEMS
#define GOF_SCAN_SMALL_ITEMS 4
#endif
constexpr int SCAN_ITEMS_BIG = 16, SCAN_ITEMS_SMALL = GOF_SCAN_SMALL_ITEMS;
constexpr uint32_t SCAN_DIRECT_MAX = 2048;           // up to this many workgroups the scan runs as two launches
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

// hist_items_dev (nullable; the radix sort's [digit][block] histogram of a launch sized for a CAPACITY, rs_hist): the scan covers only
// the blocks that hold items -- RS_DIGITS x ceil(count / RS_BLOCK) words; the workgroups past that write a zero sum and leave
__device__ __forceinline__ uint32_t hist_words(uint32_t n, const uint32_t* __restrict__ hist_items_dev, uint32_t capacity);
template <int SCAN_ITEMS>
__global__ void __launch_bounds__(256)
scan_block_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ sums, const uint32_t* __restrict__ hist_items_dev, uint32_t capacity)
{
    constexpr uint32_t SCAN_BLOCK = 256u * SCAN_ITEMS;
    __shared__ uint32_t s_wave[4];
    n = hist_words(n, hist_items_dev, capacity);
    if (blockIdx.x * SCAN_BLOCK >= n) { if (threadIdx.x == 0) sums[blockIdx.x] = 0u; return; }
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
scan_sums(uint32_t* __restrict__ sums, uint32_t nb, uint32_t* __restrict__ total_host)
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
    if (threadIdx.x == 0) {
        sums[nb] = carry;
        if (total_host) __hip_atomic_store(total_host, carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// out[i] = (inclusive ? in[0..i] : in[0..i)) summed; optional gather: in[idx[i]] instead of in[i].
// DIRECT: `sums` holds the RAW block sums (scan_sums was not run): every workgroup adds up the sums of the workgroups before it
// itself (at most SCAN_DIRECT_MAX values -- cheaper than a third launch) and the last one leaves the grand total in sums[nb].
template <bool INCLUSIVE, bool GATHER, bool DIRECT, int SCAN_ITEMS>
__global__ void __launch_bounds__(256)
scan_apply(const uint32_t* __restrict__ in, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ sums,
           uint32_t* __restrict__ out, uint32_t* __restrict__ total_host, const uint32_t* __restrict__ hist_items_dev, uint32_t capacity)
{
    constexpr uint32_t SCAN_BLOCK = 256u * SCAN_ITEMS;
    __shared__ uint32_t s_wave[4];
    if (hist_items_dev) {                           // (nobody reads the grand total of a histogram scan)
        n = hist_words(n, hist_items_dev, capacity);
        if (blockIdx.x * SCAN_BLOCK >= n) return;
    }
    uint32_t before = 0;
    if (DIRECT) {
        __shared__ uint32_t s_pre[4];
        uint32_t pre = 0;
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256) pre += sums[i];
        for (int off = 32; off > 0; off >>= 1) pre += __shfl_down(pre, off, 64);
        if ((threadIdx.x & 63) == 0) s_pre[threadIdx.x >> 6] = pre;
        __syncthreads();
        before = (s_pre[0] + s_pre[1]) + (s_pre[2] + s_pre[3]);
    } else {
        before = sums[blockIdx.x];
    }
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
    uint32_t run = before + block_exclusive_scan(s, &total, s_wave);
    if (DIRECT && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        sums[gridDim.x] = before + total;
        // total_host (nullable): host-mapped pinned memory of the caller -- the grand total without a copy launch behind the scan
        if (total_host) __hip_atomic_store(total_host, before + total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = INCLUSIVE ? run + v[k] : run;
        run += v[k];
    }
}
template <bool GATHER, int SCAN_ITEMS>
__global__ void __launch_bounds__(256)
scan_block_sums_gather(const uint32_t* __restrict__ in, const uint32_t* __restrict__ idx, uint32_t n, uint32_t* __restrict__ sums)
{
    constexpr uint32_t SCAN_BLOCK = 256u * SCAN_ITEMS;
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) s += GATHER ? in[idx[base + k]] : in[base + k];
    uint32_t total;
    block_exclusive_scan(s, &total, s_wave);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

size_t scan_tmp_words(size_t n) { return (n + 256 * SCAN_ITEMS_SMALL - 1) / (256 * SCAN_ITEMS_SMALL) + 2; }      // (enough for either tile size)

// out = scan(in[idx]) if idx != nullptr else scan(in).  tmp: scan_tmp_words(n) u32.  The grand total is left in
// tmp[nblocks] (device); total_dev_out (optional) receives its address.
// total_host (nullable): a DEVICE-VISIBLE address of host memory (hipHostGetDevicePointer) that receives the grand total as well.
template <int ITEMS>
static hipError_t device_scan_t(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool inclusive, uint32_t* tmp,
                                const uint32_t** total_dev_out, hipStream_t stream, uint32_t* total_host,
                                const uint32_t* hist_items_dev, uint32_t capacity)
{
    constexpr size_t BLOCK = 256 * ITEMS;
    const uint32_t nb = (uint32_t)((n + BLOCK - 1) / BLOCK);
    if (total_dev_out) *total_dev_out = tmp + nb;
    if (n == 0) {
        if (total_host) *total_host = 0;       // (host-mapped: a plain host store; nothing is queued for an empty input)
        return hipMemsetAsync(tmp, 0, 2 * sizeof(uint32_t), stream);
    }
    if (idx) hipLaunchKernelGGL((scan_block_sums_gather<true, ITEMS>), dim3(nb), dim3(256), 0, stream, in, idx, (uint32_t)n, tmp);
    else hipLaunchKernelGGL((scan_block_sums<ITEMS>), dim3(nb), dim3(256), 0, stream, in, (uint32_t)n, tmp, hist_items_dev, capacity);
    const bool direct = nb <= SCAN_DIRECT_MAX;         // few workgroups: each adds up its predecessors' sums itself, two launches
    if (!direct) hipLaunchKernelGGL(scan_sums, dim3(1), dim3(256), 0, stream, tmp, nb, total_host);
#define GOF_SCAN_APPLY(INC, GA)                                                                                                        \
    do { if (direct) hipLaunchKernelGGL((scan_apply<INC, GA, true, ITEMS>), dim3(nb), dim3(256), 0, stream, in, idx, (uint32_t)n, tmp, out, total_host, hist_items_dev, capacity);   \
         else hipLaunchKernelGGL((scan_apply<INC, GA, false, ITEMS>), dim3(nb), dim3(256), 0, stream, in, idx, (uint32_t)n, tmp, out, (uint32_t*)nullptr, hist_items_dev, capacity); } while (0)
    if (idx) { if (inclusive) GOF_SCAN_APPLY(true, true); else GOF_SCAN_APPLY(false, true); }
    else { if (inclusive) GOF_SCAN_APPLY(true, false); else GOF_SCAN_APPLY(false, false); }
#undef GOF_SCAN_APPLY
    return hipGetLastError();
}
static hipError_t device_scan_impl(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool inclusive, uint32_t* tmp,
                                   const uint32_t** total_dev_out, hipStream_t stream, uint32_t* total_host,
                                   const uint32_t* hist_items_dev, uint32_t capacity)
{
    // small tiles while the two-launch form still holds (<= SCAN_DIRECT_MAX workgroups of 1024 items)
    if (n <= (size_t)SCAN_DIRECT_MAX * 256 * SCAN_ITEMS_SMALL)
        return device_scan_t<SCAN_ITEMS_SMALL>(in, idx, out, n, inclusive, tmp, total_dev_out, stream, total_host, hist_items_dev, capacity);
    return device_scan_t<SCAN_ITEMS_BIG>(in, idx, out, n, inclusive, tmp, total_dev_out, stream, total_host, hist_items_dev, capacity);
}
hipError_t device_scan_u32(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool inclusive, uint32_t* tmp,
                           const uint32_t** total_dev_out, hipStream_t stream)
{
    return device_scan_impl(in, idx, out, n, inclusive, tmp, total_dev_out, stream, nullptr, nullptr, 0);
}

// ---------------------------------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------------------------------
// mask of the lanes (among `valid`) whose 8-bit digit equals this lane's digit
// Per bit: s = the bit spread over a word (v_bfe_i32), its ballot, and peers &= ~(ballot ^ s) as ONE three-input boolean
// instruction per half (v_bitop3_b32, gfx950; truth table a & ~(b ^ c) = 0x90): 4-5 VALU per bit where select + xor + and took 8.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, uint64_t valid)
{
    uint32_t lo = (uint32_t)valid, hi = (uint32_t)(valid >> 32);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);      // 0 or 0xFFFFFFFF
        const uint64_t bal = __ballot(s != 0u);
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)bal, s, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(bal >> 32), s, 0x90);
    }
    return ((uint64_t)hi << 32) | lo;
}

// ---- block kernels: 256 threads, RS_BLOCK items; wave w owns the contiguous quarter [w*RS_BLOCK/4, ...) ----
constexpr int RS_STEPS = RS_CHUNK / 64;              // steps of 64 items per wave
constexpr int RS_BLOCK = 4 * RS_CHUNK;               // items per block
// blocks of a sort launch sized for `capacity` items that hold items when the count lives on the device (never 0: block 0 then
// writes an all-zero histogram row) -- the stride of the [digit][block] histogram and the bound of its scan
__device__ __forceinline__ uint32_t rs_active_blocks(uint32_t capacity, const uint32_t* __restrict__ n_dev)
{
    const uint32_t n = device_item_count(capacity, n_dev);
    return n ? (n + RS_BLOCK - 1) / RS_BLOCK : 1u;
}
__device__ __forceinline__ uint32_t hist_words(uint32_t n, const uint32_t* __restrict__ hist_items_dev, uint32_t capacity)
{
    if (!hist_items_dev) return n;
    const uint32_t w = (uint32_t)RS_DIGITS * rs_active_blocks(capacity, hist_items_dev);
    return w < n ? w : n;
}

__global__ void __launch_bounds__(256)
rs_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ hist, uint32_t nblocks,
        const uint32_t* __restrict__ n_dev)
{
    // device-side item count (sync-free forward): n is then the capacity.  Only the blocks that hold items take part, and they are the
    // histogram's stride: its scan and the scatter's blocks past the count cost nothing (a capacity learnt on the view with the most
    // instances serves every view)
    if (n_dev) nblocks = rs_active_blocks(n, n_dev);
    if (blockIdx.x >= nblocks) return;
    n = device_item_count(n, n_dev);
    // counting only (no ranks needed here): one LDS atomic per item, all 16 loads of a thread in flight together
    __shared__ uint32_t s_cnt[RS_DIGITS];
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t begin = blockIdx.x * RS_BLOCK + threadIdx.x;
    uint32_t k[RS_BLOCK / 256];
#pragma unroll
    for (int s = 0; s < RS_BLOCK / 256; s++) { const uint32_t i = begin + s * 256; k[s] = i < n ? keys[i] : 0u; }
#pragma unroll
    for (int s = 0; s < RS_BLOCK / 256; s++) { const uint32_t i = begin + s * 256; if (i < n) atomicAdd(&s_cnt[(k[s] >> shift) & 0xFFu], 1u); }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_cnt[threadIdx.x];
}

// Stable scatter of one block: keys/values are ranked in registers (wave-level multisplit with per-wave LDS
// cursors), exchanged through LDS into digit-sorted order and written out as contiguous runs per digit.
__global__ void __launch_bounds__(256)
rs_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
           uint32_t* __restrict__ vals_out, uint32_t n, int shift, const uint32_t* __restrict__ offs, uint32_t nblocks,
           const uint32_t* __restrict__ n_dev)
{
    if (n_dev) nblocks = rs_active_blocks(n, n_dev);      // (as rs_hist)
    if (blockIdx.x >= nblocks) return;
    n = device_item_count(n, n_dev);
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
    const uint32_t blk_n = blk_begin < n ? min((uint32_t)RS_BLOCK, n - blk_begin) : 0u;   // tiles past a device-side count are empty
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

// ---------------------------------------------------------------------------------------------------
// "onesweep" passes: ONE kernel per digit instead of histogram + 3 scan kernels + scatter.
//   os_hist      -- one read of the keys builds the global digit histograms of ALL passes (LDS atomics, then 256 global atomics
//                   per block and pass); every tile of a pass scans the pass's 256 counts into digit bases itself.
//   os_pass      -- a block takes a ticket (logical tile id = scheduling order, so every predecessor is resident), ranks its tile
//                   exactly like rs_scatter, publishes its per-digit counts as (AGGREGATE | count) descriptors, thread d walks the
//                   predecessors' descriptors of digit d backwards until it meets an inclusive PREFIX (decoupled look-back),
//                   publishes its own PREFIX, and scatters.  Per pass the keys/values are read once and written once.
// A block only ever waits for blocks with smaller tickets, which never wait for it: no deadlock.  The poll loop is bounded all
// the same (OS_SPIN_LIMIT): on expiry it raises a device flag and carries on with a wrong offset instead of hanging the GPU.
// ---------------------------------------------------------------------------------------------------
constexpr int OS_MAX_PASSES = 4;
constexpr uint32_t OS_FLAG_AGG = 1u << 30, OS_FLAG_PREFIX = 2u << 30, OS_VALUE = (1u << 30) - 1u;
constexpr uint32_t OS_SPIN_LIMIT = 1u << 18;
#ifndef GOF_OS_MAX_UNITS
#define GOF_OS_MAX_UNITS 512
#endif
constexpr uint32_t OS_MAX_UNITS = GOF_OS_MAX_UNITS;          // tiles of RS_BLOCK items: up to 2M pairs
#ifndef GOF_OS_LOOKBACK
#define GOF_OS_LOOKBACK 8
#endif
constexpr int OS_LOOKBACK = GOF_OS_LOOKBACK;                 // predecessor descriptors requested per look-back round trip (os_pass)
constexpr int OS_HDR = OS_MAX_PASSES * RS_DIGITS + 64;       // digit bases of every pass, then tickets[4], error flag

// one block per RS_BLOCK keys (grid-stride beyond 1024 blocks): all of a thread's loads of a round are in flight together (until round 5
// one key per loop trip: sixteen dependent round trips per thread).  One LDS atomic per key and pass: counting the crowded digits of a
// wave -- the high bytes of depth keys take a handful of values -- per group of equal digits instead (one ballot per group, one add per
// group) was built and measured no faster (depth sort 0.155 vs 0.149 ms at S1M, profiles/r05_ab_call7_binning.txt)
__global__ void __launch_bounds__(256)
os_hist(const uint32_t* __restrict__ keys, uint32_t n, int npass, uint32_t* __restrict__ gbase, const uint32_t* __restrict__ n_dev)
{
    n = device_item_count(n, n_dev);
    __shared__ uint32_t s_h[OS_MAX_PASSES][RS_DIGITS];
    for (int p = 0; p < npass; p++) s_h[p][threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t base = blockIdx.x * RS_BLOCK; base < n; base += gridDim.x * RS_BLOCK) {      // (block-uniform)
        uint32_t k[RS_BLOCK / 256];
#pragma unroll
        for (int s = 0; s < RS_BLOCK / 256; s++) { const uint32_t i = base + s * 256 + threadIdx.x; k[s] = i < n ? keys[i] : 0u; }
#pragma unroll
        for (int s = 0; s < RS_BLOCK / 256; s++) {
            const uint32_t i = base + s * 256 + threadIdx.x;
            if (base + s * 256 < n)                          // (block-uniform: whole rows past the end are skipped)
                if (i < n) for (int p = 0; p < npass; p++) atomicAdd(&s_h[p][(k[s] >> (8 * p)) & 0xFFu], 1u);
        }
    }
    __syncthreads();
    for (int p = 0; p < npass; p++) {
        const uint32_t c = s_h[p][threadIdx.x];
        if (c) atomicAdd(&gbase[p * RS_DIGITS + threadIdx.x], c);
    }
}

__global__ void __launch_bounds__(256)
os_pass(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
        uint32_t* __restrict__ vals_out, uint32_t n, int shift, const uint32_t* __restrict__ gbase, uint32_t* __restrict__ desc,
        uint32_t* __restrict__ ticket, uint32_t* __restrict__ err, const uint32_t* __restrict__ n_dev)
{
    n = device_item_count(n, n_dev);
    __shared__ uint32_t s_cur[4][RS_DIGITS];     // per-wave digit counts, then running cursors
    __shared__ uint32_t s_bstart[RS_DIGITS];     // block-local start of every digit in sorted order
    __shared__ uint32_t s_gbase[RS_DIGITS];      // global start of this block's run of every digit
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_key[RS_BLOCK];
    __shared__ uint32_t s_val[RS_BLOCK];
    __shared__ uint32_t s_tile;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
    for (int k = 0; k < 4; k++) s_cur[wave][lane + 64 * k] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t blk_begin = tile * RS_BLOCK;
    const uint32_t begin = blk_begin + wave * RS_CHUNK;
    const uint32_t hist_d = gbase[threadIdx.x];  // (requested early: needed behind the look-back)

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
    {   // digit d = threadIdx.x: counts of the 4 waves -> publish, look back, cursors
        const uint32_t d = threadIdx.x;
        const uint32_t c0 = s_cur[0][d], c1 = s_cur[1][d], c2 = s_cur[2][d], c3 = s_cur[3][d];
        const uint32_t tot = c0 + c1 + c2 + c3;
        uint32_t* my = desc + (size_t)tile * RS_DIGITS + d;
        uint32_t excl = 0;
        if (tile > 0) {
            __hip_atomic_store(my, OS_FLAG_AGG | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // Look-back, OS_LOOKBACK predecessors per round trip: the descriptors of tiles j, j - 1, ... are requested together and
            // consumed in order up to the first one that is not published yet (or the first PREFIX).  A descriptor is a device-scope
            // load (it is another XCD's store), and when the tiles of a pass start together every one of them finds AGGREGATEs on
            // its nearest predecessors.  Measured (profiles/r05_ab_call6_binning.txt): 1 / 8 descriptors per round trip = depth sort
            // 0.157 / 0.150 ms at S1M -- the passes are bound by their chain of dependent steps (ticket, loads, ranks, look-back,
            // exchange, stores), not by the look-back alone; and beyond 512 tiles the single-kernel passes still lose to histogram /
            // scan / scatter (tile sort of 8.8 M pairs 0.38 vs 0.23 ms), whatever the width.
            int j = (int)tile - 1;
            uint32_t spins = 0;
            bool done = false;
            while (j >= 0 && !done) {
                uint32_t v[OS_LOOKBACK];
#pragma unroll
                for (int k = 0; k < OS_LOOKBACK; k++)
                    v[k] = (j - k >= 0) ? __hip_atomic_load(desc + (size_t)(j - k) * RS_DIGITS + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                int used = 0;
#pragma unroll
                for (int k = 0; k < OS_LOOKBACK; k++) {
                    if (done || used != k || j - k < 0 || (v[k] >> 30) == 0u) continue;      // (consumed strictly in order)
                    excl += v[k] & OS_VALUE;
                    used = k + 1;
                    if (v[k] & OS_FLAG_PREFIX) done = true;
                }
                j -= used;
                if (used) { spins = 0; continue; }
                ++spins;
                if (spins > OS_SPIN_LIMIT || ((spins & 1023u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    atomicExch(err, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __hip_atomic_store(my, OS_FLAG_PREFIX | ((excl + tot) & OS_VALUE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t blk_total;
        const uint32_t start = block_exclusive_scan(tot, &blk_total, s_scan);
        s_bstart[d] = start;
        // gbase: the RAW global count of every digit of this pass (os_hist); its exclusive scan = where the digit's run starts is
        // formed here, by every tile for itself (256 values: one block scan) -- a single-workgroup launch between the histogram and
        // the first pass did it until round 5 (os_scan_hist: 12 us on the critical path for 1024 additions)
        uint32_t all;
        const uint32_t digit_base = block_exclusive_scan(hist_d, &all, s_scan);
        s_gbase[d] = digit_base + excl;
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
    const uint32_t blk_n = blk_begin < n ? min((uint32_t)RS_BLOCK, n - blk_begin) : 0u;   // tiles past a device-side count are empty
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

// u32 words of scratch: header (digit bases of 4 passes, tickets, error flag) + one descriptor per (pass, block, digit); never
// smaller than what the three-kernel fallback needs
size_t rs_tmp_words(size_t n)
{
    const size_t h = (size_t)RS_DIGITS * rs_units(n);
    const size_t onesweep = (size_t)OS_HDR + (size_t)OS_MAX_PASSES * h;
    const size_t classic = h + scan_tmp_words(h) + 64;
    return onesweep > classic ? onesweep : classic;
}

// Stable sort of (key, value) pairs on key bits [0, end_bit), 8 bits per pass.  Buffers a* hold the input; the
// result ends up in (*keys_res, *vals_res), which alias either a* or b*.  n_dev (nullable): the item count lives on the device
// (sync-free forward); n is then the CAPACITY the launches are sized for and every kernel clamps to min(n, *n_dev).
// zero_words_behind: that many u32 words right behind the sort's scratch (tmp + rs_tmp_words(n)) are cleared as well -- by the memset the
// single-kernel passes need anyway where it ends there, by one of its own otherwise (the caller's next kernel keeps its state there).
hipError_t radix_sort_pairs_u32_z(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int end_bit,
                                  uint32_t* tmp, uint32_t** keys_res, uint32_t** vals_res, hipStream_t stream, const uint32_t* n_dev,
                                  size_t zero_words_behind, bool first_hist_done, bool scratch_zeroed)
{
    uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
    const int npass = (end_bit + 7) / 8;
    if (n > 0 && npass > 0) {
        const uint32_t nunits = rs_units(n);
        const size_t hwords = (size_t)RS_DIGITS * nunits;
        const dim3 grid(nunits), block(256);
        // measured on MI355X: 1M pairs x 4 passes 0.142 -> 0.107 ms, but 8.8M pairs x 2 passes 0.178 -> 0.199 ms (round 3; round 5 with
        // 8 or 16 descriptors per look-back round trip: still 0.38 vs 0.23 ms, and 37 M query points 1.93 vs 1.75 ms): the single-kernel
        // passes are used where launch latency dominates
        if (nunits <= OS_MAX_UNITS && npass <= OS_MAX_PASSES) {
            uint32_t* gbase = tmp;
            uint32_t* tickets = tmp + OS_MAX_PASSES * RS_DIGITS;
            uint32_t* err = tickets + 8;
            uint32_t* desc = tmp + OS_HDR;
            if (!scratch_zeroed) {       // (else: the caller's kernel in front of the sort cleared radix_zero_words(n, end_bit) words at tmp)
            size_t zero_words = (size_t)OS_HDR + (size_t)npass * hwords;
            if (zero_words_behind && zero_words == rs_tmp_words(n)) { zero_words += zero_words_behind; zero_words_behind = 0; }
            hipError_t e = hipMemsetAsync(tmp, 0, zero_words * sizeof(uint32_t), stream);
            if (e != hipSuccess) return e;
            if (zero_words_behind) { e = hipMemsetAsync(tmp + rs_tmp_words(n), 0, zero_words_behind * sizeof(uint32_t), stream); if (e != hipSuccess) return e; }
            }
            hipLaunchKernelGGL(os_hist, dim3(nunits < 1024u ? nunits : 1024u), block, 0, stream, ki, (uint32_t)n, npass, gbase, n_dev);
            for (int p = 0; p < npass; p++) {
                hipLaunchKernelGGL(os_pass, grid, block, 0, stream, ki, vi, ko, vo, (uint32_t)n, 8 * p, gbase + p * RS_DIGITS,
                                   desc + (size_t)p * hwords, tickets + p, err, n_dev);
                uint32_t* t;
                t = ki; ki = ko; ko = t;
                t = vi; vi = vo; vo = t;
            }
            *keys_res = ki;
            *vals_res = vi;
            return hipGetLastError();
        }
        if (zero_words_behind) { hipError_t e = hipMemsetAsync(tmp + rs_tmp_words(n), 0, zero_words_behind * sizeof(uint32_t), stream); if (e != hipSuccess) return e; }
        uint32_t* hist = tmp;
        uint32_t* scan_tmp = tmp + hwords;
        for (int shift = 0; shift < end_bit; shift += 8) {
            // (first_hist_done: the kernel that produced the keys left the first pass's [digit][block] histogram at `tmp`: radix_classic_hist)
            if (!(first_hist_done && shift == 0)) hipLaunchKernelGGL(rs_hist, grid, block, 0, stream, ki, (uint32_t)n, shift, hist, nunits, n_dev);
            hipError_t e = device_scan_impl(hist, nullptr, hist, hwords, false, scan_tmp, nullptr, stream, nullptr, n_dev, (uint32_t)n);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(rs_scatter, grid, block, 0, stream, ki, vi, ko, vo, (uint32_t)n, shift, hist, nunits, n_dev);
            uint32_t* t;
            t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
        }
    }
    else if (zero_words_behind) { hipError_t e = hipMemsetAsync(tmp + rs_tmp_words(n), 0, zero_words_behind * sizeof(uint32_t), stream); if (e != hipSuccess) return e; }
    *keys_res = ki;
    *vals_res = vi;
    return hipGetLastError();
}
hipError_t radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int end_bit,
                                uint32_t* tmp, uint32_t** keys_res, uint32_t** vals_res, hipStream_t stream, const uint32_t* n_dev)
{
    return radix_sort_pairs_u32_z(keys_a, vals_a, keys_b, vals_b, n, end_bit, tmp, keys_res, vals_res, stream, n_dev, 0, false, false);
}
int radix_passes(int end_bit) { return (end_bit + 7) / 8; }
// Where a sort of n pairs on end_bit bits that runs as histogram / scan / scatter launches expects the [digit][block] histogram of its FIRST
// pass (blocks of rs_block_items() consecutive items; stride = the blocks that hold items) -- nullptr if that sort runs as single-kernel
// passes.  A kernel that produces the keys block by block can leave it there (first_hist_done of radix_sort_pairs_u32_z).
uint32_t* radix_classic_hist(uint32_t* tmp, size_t n, int end_bit)
{
    const int npass = (end_bit + 7) / 8;
    if (n == 0 || npass <= 0 || (rs_units(n) <= OS_MAX_UNITS && npass <= OS_MAX_PASSES)) return nullptr;
    return tmp;
}
uint32_t rs_block_items() { return (uint32_t)RS_BLOCK; }
// u32 words at the start of the scratch that a sort of n pairs on end_bit bits wants cleared before its first launch (its single-kernel
// passes' digit counts, tickets, error flag and descriptors); 0 if it runs as histogram / scan / scatter launches
size_t radix_zero_words(size_t n, int end_bit)
{
    const int npass = (end_bit + 7) / 8;
    if (n == 0 || npass <= 0 || rs_units(n) > OS_MAX_UNITS || npass > OS_MAX_PASSES) return 0;
    return (size_t)OS_HDR + (size_t)npass * (size_t)RS_DIGITS * rs_units(n);
}

// Device word that is non-zero after radix_sort_pairs_u32(.., n, end_bit, tmp, ..) if a single-kernel pass gave up waiting for a
// predecessor (bounded look-back poll); nullptr when that sort runs as histogram / scan / scatter launches, which cannot time out.
const uint32_t* radix_sort_error_flag(const uint32_t* tmp, size_t n, int end_bit)
{
    const int npass = (end_bit + 7) / 8;
    if (n == 0 || npass <= 0 || rs_units(n) > OS_MAX_UNITS || npass > OS_MAX_PASSES) return nullptr;
    return tmp + OS_MAX_PASSES * RS_DIGITS + 8;
}

} // namespace gof
