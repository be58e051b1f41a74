// binning.hip -- tile binning: which Gaussians cover which tile, front to back.
//
// Replaces, with the identical resulting order (reference rasterizer_impl.cu:332-373):
//   cub InclusiveSum + duplicateWithKeys + cub SortPairs on (tile << 32 | depth) over 32 + log2(T) bits
//   + cudaMemset + identifyTileRanges.
//
// The reference sorts R instances with 64-bit keys (6 radix passes of 12 B/instance at 1600x1063).  The
// same order -- per tile, ascending depth bits, ties in ascending Gaussian index -- is obtained with far
// less traffic by sorting where the information lives:
//   1. stable sort of the P Gaussians by the 32 depth bits (4 passes over P, not R); ties keep ascending id;
//   2. exclusive scan of tiles_touched in that order -> first instance of every sorted Gaussian;
//   3. emit instances in depth order: (tile id, Gaussian id), tiles y-major / x-minor as the reference;
//   4. stable sort of the R instances by tile id only (ceil(log2(T)/8) = 2 passes of 8 B/instance);
//   5. tile ranges from the boundaries of the sorted tile ids.
// Equal (tile, depth) entries end up in ascending Gaussian index exactly as with the reference's single
// stable sort, because both sorts are stable and step 3 preserves the order of step 1.
// Scan and sort are the hand-written kernels of radix.hip.
#include "gof_common.h"
#include "gof_status.h"

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

// Everything the instance emission needs of a Gaussian, gathered into DEPTH order with one random 8-byte read per Gaussian (the tile
// rectangle preprocess_fwd packed) + the exclusive scan of the instance counts, in ONE launch (replaces cub::DeviceScan::InclusiveSum
// of rasterizer_impl.cu:332 together with the gather in front of it; until round 5: a gather kernel, scan_block_sums, scan_apply = three
// launches, 31 us on the critical path of a 1M-Gaussian frame, of which the kernels' own work is a third).  A workgroup takes a ticket
// (= its tile of GS_BLOCK consecutive positions of the depth order: every tile with a smaller ticket is running or done, so waiting
// for them cannot deadlock), gathers the rectangles -- a culled Gaussian (sort key 0xFFFFFFFF: they sort last) has the empty
// rectangle and its random read is skipped; after a depth sort whose bounded look-back poll expired (`order` is then not a sorted
// permutation) every rectangle is empty and the LAST count is GOF_SORT_FAILED_COUNT, so that the call fails instead of rendering
// garbage and the launches already queued see 0 items --, scans its counts (wave by wave: a wave owns 1024 consecutive positions, 16 coalesced steps
// of 64), publishes its total as an AGGREGATE descriptor, and wave 0 looks back over 64 predecessors per round trip -- their
// descriptors in one load instruction, consumed in order up to the first PREFIX -- until it knows the sum in front of its tile;
// then the tile publishes its PREFIX and stores its offsets.  Descriptors are 64-bit (flag << 32 | value): the value is an instance
// count, and the failure sentinel of a timed-out depth sort (GOF_SORT_FAILED_COUNT, gof_status.h) must survive the scan unchanged.
// state (u32 words, zeroed in front of the launch): [0] ticket, [1] grand total (the device-side instance count), [2] raised by a tile
// whose look-back poll expired, [4..] descriptors.
#ifndef GOF_GS_ITEMS
#define GOF_GS_ITEMS 4
#endif
constexpr int GS_ITEMS = GOF_GS_ITEMS;                 // per lane
// waves per workgroup (round 6: as the single-kernel radix passes, radix.hip OS_WAVES -- the same tile spread over more waves, each
// scanning fewer steps of 64 positions in sequence)
#ifndef GOF_GS_WAVES
#define GOF_GS_WAVES 16      // (measured, profiles/r06_ab_call1_binning.txt: 16 waves x 4 positions per lane against 4 x 16 -- scan_tiles 0.0285 -> 0.0227 ms at S1M, 6M: unchanged)
#endif
constexpr int GS_WAVES = GOF_GS_WAVES;
constexpr uint32_t GS_THREADS = 64u * GS_WAVES;
constexpr uint32_t GS_BLOCK = GS_THREADS * GS_ITEMS;   // positions per workgroup
uint32_t gather_scan_threads() { return GS_THREADS; }
constexpr uint32_t GS_SPIN_LIMIT = 1u << 18;
uint32_t gather_scan_tiles(size_t n) { return (uint32_t)((n + GS_BLOCK - 1) / GS_BLOCK); }
size_t gather_scan_state_words(size_t n) { return 4 + 2 * (size_t)gather_scan_tiles(n) + 2; }
__device__ __forceinline__ unsigned long long* gather_scan_desc(uint32_t* state)
{
    return reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(state + 4) + 7u) & ~(uintptr_t)7u);
}
__global__ void __launch_bounds__(GS_THREADS)
gather_scan_rects(uint32_t n, const uint2* __restrict__ rect, const uint32_t* __restrict__ order, const uint32_t* __restrict__ keys_sorted,
                  uint32_t* __restrict__ minxy_sorted, uint32_t* __restrict__ wh_sorted, uint32_t* __restrict__ order_off,
                  const uint32_t* __restrict__ sort_error, uint2* __restrict__ ranges, uint32_t ntiles, uint32_t* __restrict__ state,
                  uint32_t* __restrict__ total_host)
{
    __shared__ uint32_t s_tile, s_base;
    __shared__ uint32_t s_wtot[GS_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // (the tile ranges must be zero before tile_ranges fills them in -- cudaMemset of rasterizer_impl.cu:365 --: cleared here, on the way)
    for (uint32_t t = blockIdx.x * GS_THREADS + tid; t < ntiles; t += gridDim.x * GS_THREADS) ranges[t] = make_uint2(0u, 0u);
    if (tid == 0) s_tile = atomicAdd(&state[0], 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const bool failed = sort_error && *sort_error != 0u;          // the depth sort timed out (see above)
    const uint32_t wbase = tile * GS_BLOCK + wave * (64u * GS_ITEMS);
    uint32_t ord[GS_ITEMS], cnt[GS_ITEMS];
    uint2 r[GS_ITEMS];
#pragma unroll
    for (int s = 0; s < GS_ITEMS; s++) {
        const uint32_t i = wbase + 64u * s + lane;
        // no rectangle to read (0xFFFFFFFF): past the end, a failed sort, or a culled Gaussian (sort key 0xFFFFFFFF).  Key
        // and id are requested together -- one round trip, not two: this kernel is a chain of dependent loads
        const bool ok = i < n && !failed;
        const uint32_t key = (ok && keys_sorted) ? keys_sorted[i] : 0u;
        const uint32_t id = ok ? order[i] : 0xFFFFFFFFu;
        ord[s] = key == 0xFFFFFFFFu ? 0xFFFFFFFFu : id;
    }
#pragma unroll
    for (int s = 0; s < GS_ITEMS; s++) r[s] = (ord[s] != 0xFFFFFFFFu) ? rect[ord[s]] : make_uint2(0u, 0u);
    uint32_t run = 0;                                  // (wave-uniform) sum of the wave's counts so far
#pragma unroll
    for (int s = 0; s < GS_ITEMS; s++) {
        const uint32_t i = wbase + 64u * s + lane;
        uint32_t c = (r[s].y & 0xFFFFu) * (r[s].y >> 16);
        if (failed) c = (i == n - 1u) ? GOF_SORT_FAILED_COUNT : 0u;
        if (i < n) { minxy_sorted[i] = r[s].x; wh_sorted[i] = r[s].y; } else c = 0u;
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)inc, o); if (lane >= (uint32_t)o) inc += up; }
        cnt[s] = run + inc - c;                        // exclusive within the wave
        run += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) s_wtot[wave] = run;
    __syncthreads();
    uint32_t total = 0, wave_excl = 0;
#pragma unroll
    for (int w = 0; w < GS_WAVES; w++) { const uint32_t tw = s_wtot[w]; if ((uint32_t)w < wave) wave_excl += tw; total += tw; }
    if (wave == 0) {
        unsigned long long* const desc = gather_scan_desc(state);
        constexpr unsigned long long AGG = 1ull << 32, PREFIX = 2ull << 32;
        if (tile > 0 && lane == 0) __hip_atomic_store(desc + tile, AGG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0, spins = 0;
        int j = (int)tile - 1;
        bool expired = false;
        while (j >= 0) {
            const int idx = j - (int)lane;
            // (a lane past tile 0 stands for "PREFIX 0": the walk always ends at a PREFIX)
            const unsigned long long v = idx >= 0 ? __hip_atomic_load(desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : PREFIX;
            const uint32_t flag = (uint32_t)(v >> 32);
            const unsigned long long ready = __ballot(flag != 0u);
            const int nready = (ready == ~0ull) ? 64 : (int)__builtin_ctzll(~ready);             // published descriptors, counted from the nearest predecessor
            const unsigned long long pref = __ballot(flag == 2u) & (nready == 64 ? ~0ull : ((1ull << nready) - 1ull));
            const int upto = pref ? (int)__builtin_ctzll(pref) + 1 : nready;                      // lanes [0, upto) are consumed
            uint32_t val = ((int)lane < upto) ? (uint32_t)v : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) val += (uint32_t)__shfl_xor((int)val, o);
            excl += val;
            if (pref) break;
            j -= upto;
            if (upto) { spins = 0; continue; }
            if (++spins > GS_SPIN_LIMIT) { expired = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        // a poll that expired (GPU heavily oversubscribed): this tile's offsets are wrong.  It says so in state[2] BEFORE it publishes;
        // the last tile, whose own look-back cannot end before every predecessor has published, then writes the failure sentinel as
        // the grand total: the call fails instead of rendering from wrong offsets (as after a timed-out radix pass, radix.hip)
        // (ordering: the flag is stored, a RELEASE fence, then the PREFIX -- relaxed, as every descriptor --; the last tile has seen every
        // PREFIX through relaxed loads, passes an ACQUIRE fence and reads the flag: fence-to-fence synchronisation, so a PREFIX that is
        // visible carries the flag stored in front of it.  Until round 6 the release sat on the flag's store itself, which orders what
        // comes BEFORE it, not the PREFIX behind it.)
        if (expired && lane == 0) {
            __hip_atomic_store(&state[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        if (lane == 0) {
            __hip_atomic_store(desc + tile, PREFIX | (unsigned long long)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = excl;
            if (tile == gridDim.x - 1u) {
                uint32_t grand = excl + total;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (__hip_atomic_load(&state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) grand = GOF_SORT_FAILED_COUNT;
                state[1] = grand;
                if (total_host) __hip_atomic_store(total_host, grand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    const uint32_t base = s_base + wave_excl;
#pragma unroll
    for (int s = 0; s < GS_ITEMS; s++) {
        const uint32_t i = wbase + 64u * s + lane;
        if (i < n) order_off[i] = base + cnt[s];
    }
}

// Instance emission in depth order (replaces duplicateWithKeys, rasterizer_impl.cu:70-111).
//
// The OUTPUT is what is divided among the waves, not the Gaussians: wave k writes the slots [k EMIT_SLOTS, (k + 1) EMIT_SLOTS) of the
// instance arrays, whichever Gaussians they belong to.  (Until round 4 a wave owned 64 consecutive depth-sorted Gaussians and wrote
// all their instances.  Depth order puts Gaussians of similar distance next to each other -- and with them those of similar SIZE on
// screen: on S1M-clustered the 2 % large splats at the back of the scene, 130-1400 tiles each, are 60 % of the instances and sat in
// 1500 of the 15 600 waves; the launch lasted as long as those: 0.21 ms for 24.8 M instances against 0.035 ms for S1M's 8.8 M.)
//   1. the Gaussian slot p0 = k EMIT_SLOTS belongs to: the LAST i with order_off[i] <= p0, found by a 64-ary search over the
//      exclusive offsets (every lane probes, one ballot per level: 4 dependent loads for P <= 16.7 M);
//   2. from there, 64 depth-sorted Gaussians at a time (lane = Gaussian: rectangle and first slot, coalesced), the wave writes the
//      part of their slots that lies in its range cooperatively: slot p belongs to the lane o with off[o] <= p < off[o] + cnt[o]
//      (6-step binary search over the lanes' offsets with ds_bpermute), entry k = p - off[o] of that Gaussian's rectangle is tile
//      (miny + k / w, minx + k % w), y-major / x-minor as the reference: 64 consecutive slots per store instruction.
// The number of a Gaussian's first instance, by Gaussian id (inst_first: the backward numbers its partial gradient records with it),
// is stored on the way by the thread whose global index is the Gaussian's position in the order.
#ifndef GOF_EMIT_SLOTS
#define GOF_EMIT_SLOTS 1024
#endif
constexpr uint32_t EMIT_SLOTS = GOF_EMIT_SLOTS;       // output slots per wave
// workgroups of an emit_instances launch that may write up to `slots` instances (api.hip sizes the grid with it)
// (and never fewer than one thread per Gaussian: the threads also store inst_first, one Gaussian each -- a view that sees few of many
// Gaussians would otherwise leave that loop to a handful of workgroups)
uint32_t emit_block_slots() { return 4u * EMIT_SLOTS; }      // slots a workgroup of emit_instances writes
uint32_t emit_instances_grid(uint32_t slots, int P)
{
    const size_t by_slots = ((size_t)slots + 4 * EMIT_SLOTS - 1) / (4 * EMIT_SLOTS), by_gaussians = ((size_t)P + 255) / 256;
    return (uint32_t)(by_slots > by_gaussians ? by_slots : by_gaussians);
}
__global__ void __launch_bounds__(256)
emit_instances(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ order_off, const uint32_t* __restrict__ minxy_sorted,
               const uint32_t* __restrict__ wh_sorted, uint32_t* __restrict__ tiles, uint32_t* __restrict__ gids, uint32_t gx, uint32_t capacity,
               uint32_t* __restrict__ inst_first, uint32_t* __restrict__ hist0, uint32_t hist_stride)
{
    // hist0 (nullable): the [digit][block] histogram of the tile sort's FIRST pass (radix.hip: rs_hist, digit = the tile id's low byte) where
    // that sort runs as histogram / scan / scatter launches over blocks of exactly this workgroup's 4 EMIT_SLOTS slots: the workgroup
    // counts the tile ids it writes (consecutive tiles of a rectangle row: the low bytes spread over the bins, few same-address adds) and
    // stores its row -- the sort starts without its first histogram launch and without that read of the keys
    __shared__ uint32_t s_cnt[256];
    if (hist0) { s_cnt[threadIdx.x] = 0u; __syncthreads(); }
    const uint32_t lane = threadIdx.x & 63u;
    if (inst_first)
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < (uint32_t)P; i += gridDim.x * 256u) {
            // (a depth sort whose look-back poll expired leaves slots of `order` unwritten: whatever lies there must not become an
            // address -- gather_scan_rects guards its reads with the sort's error flag, this store with the range itself)
            const uint32_t id = order[i];
            if (id < (uint32_t)P) inst_first[id] = order_off[i];
        }
    // the instance count: exclusive offset + count of the last Gaussian in the order (0 after a failed depth sort: gather_scan_rects)
    const uint32_t wh_last = wh_sorted[P - 1];
    const uint32_t total = order_off[P - 1] + (wh_last & 0xFFFFu) * (wh_last >> 16);
    const uint32_t limit = min(total, capacity);       // capacity < the instance count only in the sync-free forward (then redone)
    const uint64_t p0_wide = (uint64_t)(blockIdx.x * 4u + (threadIdx.x >> 6)) * EMIT_SLOTS;
    if (p0_wide < (uint64_t)limit) {                    // (wave-uniform; no early return: the histogram's barrier below is the workgroup's)
    const uint32_t p0 = (uint32_t)p0_wide;
    const uint32_t p1 = (uint32_t)min((uint64_t)limit, p0_wide + EMIT_SLOTS);
    // 1. the last position i of the order with order_off[i] <= p0 (offsets are non-decreasing, order_off[0] = 0 <= p0; a run of equal
    //    offsets = Gaussians without instances in front of the one that owns the slot)
    uint32_t lo = 0, span = (uint32_t)P;                // the answer lies in [lo, lo + span), order_off[lo] <= p0
    while (span > 1u) {
        const uint32_t step = (span + 63u) >> 6;
        const uint32_t probe = lo + lane * step;
        const bool ok = lane * step < span && order_off[probe] <= p0;
        const uint32_t n = (uint32_t)__popcll(__ballot(ok));        // >= 1: lane 0 probes lo itself
        const uint32_t adv = (n - 1u) * step;
        lo += adv;
        span = min(step, span - adv);
    }
    // 2. groups of 64 Gaussians from there on
    for (uint32_t g0 = lo; ; g0 += 64u) {
        const uint32_t i = g0 + lane;
        uint32_t idx = 0, off = 0, cnt = 0, minx = 0, miny = 0, w = 1;
        if (i < (uint32_t)P) {
            idx = order[i];
            off = order_off[i];
            const uint32_t mxy = minxy_sorted[i], wh = wh_sorted[i];
            if (wh) {
                minx = mxy & 0xFFFFu; miny = mxy >> 16;
                w = wh & 0xFFFFu;
                cnt = w * (wh >> 16);
            }
        }
        // the group's slots [first, end): offsets are an exclusive scan in this order, so they are contiguous
        const int valid_lanes = (int)min(64u, (uint32_t)P - g0);
        const uint32_t group_end = (uint32_t)__shfl((int)off, valid_lanes - 1) + (uint32_t)__shfl((int)cnt, valid_lanes - 1);
        // lanes past P (last group only) carry off = 0: give them the end of the range so the search never selects them
        if (i >= (uint32_t)P) off = group_end;
        const uint32_t first = max(p0, (uint32_t)__shfl((int)off, 0));
        const uint32_t end = min(p1, group_end);
        for (uint32_t p = first + lane; __ballot(p < end) != 0ull; p += 64u) {
            int l = 0, h = 63;
#pragma unroll
            for (int s = 0; s < 6; s++) {
                const int mid = (l + h + 1) >> 1;
                const uint32_t v = (uint32_t)__shfl((int)off, mid);
                if (v <= p) l = mid; else h = mid - 1;
            }
            const uint32_t o_off = (uint32_t)__shfl((int)off, l), o_w = (uint32_t)__shfl((int)w, l), o_minx = (uint32_t)__shfl((int)minx, l),
                           o_miny = (uint32_t)__shfl((int)miny, l), o_idx = (uint32_t)__shfl((int)idx, l);
            if (p < end) {
                const uint32_t k = p - o_off;
                const uint32_t y = k / o_w, x = k - y * o_w;
                const uint32_t t = (o_miny + y) * gx + (o_minx + x);
                tiles[p] = t;
                gids[p] = o_idx;
                if (hist0) atomicAdd(&s_cnt[t & 0xFFu], 1u);
            }
        }
        if (group_end >= p1 || g0 + 64u >= (uint32_t)P) break;      // (wave-uniform)
    }
    }
    if (hist0) {
        __syncthreads();
        // the stride of the histogram: what the SORT behind it will use -- hist_stride, the host's block count, where the sort is launched
        // with a host-known count (gof_forward_render / gof_integrate_view: rs_units(R), whatever this kernel's device-side total is: a
        // caller that passes another R than the frame's count must not make the scatter read a wrongly strided histogram), else (0) the
        // blocks that hold items by the device-side count (radix.hip: rs_active_blocks; never 0)
        const uint32_t nb = hist_stride ? hist_stride : (limit ? (limit + 4u * EMIT_SLOTS - 1u) / (4u * EMIT_SLOTS) : 1u);
        if (blockIdx.x < nb) hist0[(size_t)threadIdx.x * nb + blockIdx.x] = s_cnt[threadIdx.x];
    }
}

// Query points: one instance per point inside the image (replaces createWithKeys, rasterizer_impl.cu:113-144), compacted by the
// inclusive scan `offsets`.  Sort key: tile id in the high bits, pixel of the tile (wave-quadrant order of tile_thread) in the
// low 8.  The reference sorts its points by (tile, depth); no output depends on the order of a tile's points except channel 8's
// re-count of the tile's deepest point, which integrate_points finds explicitly -- so the 4 depth passes are dropped, and
// grouping by pixel lets neighbouring lanes of integrate_points walk the SAME contributor mask.
__global__ void __launch_bounds__(256)
point_keys(int PN, const float4* __restrict__ pos, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ tiles_touched,
           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t gx, uint32_t gy)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= PN) return;
    if (tiles_touched[idx] > 0) {
        const uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
        const float4 p = pos[idx];
        const int x = (int)min(gx - 1, (uint32_t)max(0, (int)(p.x / TILE_X)));
        const int y = (int)min(gy - 1, (uint32_t)max(0, (int)(p.y / TILE_Y)));
        const uint32_t lx = ((uint32_t)p.x - (uint32_t)x * TILE_X) & (TILE_X - 1), ly = ((uint32_t)p.y - (uint32_t)y * TILE_Y) & (TILE_Y - 1);
        keys[off] = ((uint32_t)(y * gx + x) << 8) | tile_thread(lx, ly);
        vals[off] = (uint32_t)idx;
    }
}

// per-point data in LIST order (tile-major, pixel by pixel within the tile): the point pass of integrate reads it once per staged batch,
// a gather by point id there would touch a different DRAM sector per point per batch.  Position and depth of a point are one
// 16-byte line (PointWs::pos): ONE random sector per point here, where two arrays cost two (this gather was half of bin_points).
// What is stored is the point's RAY ((x - W/2) / focal_x, (y - H/2) / focal_y in double, rounded to fp32: forward.cu:1108-1109), not
// its pixel position (BinWs::pt_xy keeps its name): integrate_points needs the ray in every staged batch of the tile -- two fp64
// divisions per point and batch, ~25 batches per tile at the config-5 shape -- and the pixel of the tile only as the low byte of the
// point's sorted key (point_keys), which lies in list order already.
__global__ void __launch_bounds__(256)
gather_sorted_points(uint32_t NI, const uint32_t* __restrict__ sorted_ids, const float4* __restrict__ pos,
                     float2* __restrict__ pt_ray, float* __restrict__ pt_depth, int W, int H, float focal_x, float focal_y)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NI) return;
    const uint32_t id = sorted_ids[i];
    const float4 p = pos[id];
    pt_ray[i] = make_float2((float)(((double)p.x - W / 2.) / (double)focal_x), (float)(((double)p.y - H / 2.) / (double)focal_y));
    pt_depth[i] = p.z;
}

// replaces cudaMemset + identifyTileRanges (rasterizer_impl.cu:365-373, 149-171); ranges must be zeroed before.
// sort_error (nullable): the tile sort's time-out flag when it ran as single-kernel passes (<= 2M instances): the list is then not
// sorted, so the ranges stay empty (the frame renders as background instead of from a mis-sorted list) and the failure is raised in
// the host-mapped status word, which the next library call reports (api.hip: take_async_status).
__global__ void __launch_bounds__(256)
tile_ranges(uint32_t L, const uint32_t* __restrict__ tiles, uint2* __restrict__ ranges, int shift, const uint32_t* __restrict__ n_dev,
            const uint32_t* __restrict__ sort_error, uint32_t* __restrict__ async_status)
{
    L = device_item_count(L, n_dev);
    if (sort_error && *sort_error) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && async_status) *async_status = 1u;
        return;
    }
    // grid-stride (api.hip caps the grid: a launch sized for a capacity far above the count -- learnt on another view -- costs no
    // empty workgroups)
    for (uint32_t idx = blockIdx.x * 256 + threadIdx.x; idx < L; idx += gridDim.x * 256u) {
        const uint32_t currtile = tiles[idx] >> shift;
        if (idx == 0) ranges[currtile].x = 0;
        else {
            const uint32_t prevtile = tiles[idx - 1] >> shift;
            if (currtile != prevtile) {
                ranges[prevtile].y = idx;
                ranges[currtile].x = idx;
            }
        }
        if (idx == L - 1) ranges[currtile].y = L;
    }
}

// Dispatch order of the tile kernels (pop_tile, gof_common.h).  ONE workgroup:
//   1. classifies every tile by cost into half-octave buckets (costs of a bucket within 3:2), heaviest bucket first;
//   2. ranks the tiles of a bucket STABLY by tile id (wave w owns a contiguous range of tile ids and walks it in order: rank =
//      tiles of the bucket in earlier waves + earlier steps + lower lanes, the latter from 7 ballots);
//   3. gives every XCD one EIGHTH of every bucket in that order (rotated by the number of tiles in heavier buckets): every XCD gets (to within one tile per bucket) the same number
//      of tiles of every cost class -- equal counts and equal work -- and its share of a bucket is a spatially contiguous run of
//      tiles, so the tiles an XCD renders still share their Gaussians' records in its private L2 (a uniform scene gets round 2's
//      contiguous band per XCD back; dealing single ranks round-robin balanced as well but doubled the L2 -> fabric reads);
//   4. writes XCD e's queue (its shares of the buckets, heaviest bucket first) to order[e * stride ...], its length to
//      queue[8 + e], and resets the heads queue[e].
// cost = tile-list length (forward: an upper bound of what the tile walks), what the forward measured (backward), or that times
// the tile's query points (point pass).  What it buys is measured in bench.py's "clustered" leg (profiles/r03_tile_schedule.md).
__global__ void __launch_bounds__(1024)
order_tiles(uint32_t ntiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ cost_in, uint32_t* __restrict__ order,
            uint32_t* __restrict__ queue, const uint2* __restrict__ times_ranges, uint32_t* __restrict__ clear_cursors, uint32_t* __restrict__ staged_out,
            const uint32_t* __restrict__ cursors_in, uint32_t* __restrict__ usage_host)
{
    if (clear_cursors && threadIdx.x <= POOL_SHARDS) clear_cursors[threadIdx.x] = 0u;      // the mask pool of the frame's forward blend starts empty
    constexpr int NB = 128;                       // bucket = 2 * floor(log2(c)) + next bit, descending (64 used; quarter-octave classes ordered
                                                  // more finely but cut an XCD's share into more, shorter runs: +30 % L2 -> fabric reads in the backward)
    constexpr int NW = 16;                        // waves of the workgroup
    __shared__ uint32_t s_wc[NW][NB];             // per (wave, bucket): count, then the running stable rank base
    __shared__ uint32_t s_n[NB];                  // tiles per bucket
    __shared__ uint32_t s_g[NB];                  // tiles in heavier buckets (rotates the eighths: small buckets go round the XCDs)
    __shared__ uint32_t s_off[NXCD][NB];          // queue position of XCD e's share of bucket b
    __shared__ uint32_t s_carry[NXCD];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t stride = tile_queue_stride(ntiles);
    for (uint32_t k = tid; k < NW * NB; k += 1024) (&s_wc[0][0])[k] = 0u;
    __shared__ uint32_t s_staged;
    if (tid == 0) s_staged = 0u;
    __syncthreads();
    // (on the way: the sum over the tiles of min(cost, list length) -> queue[BW_STAGED_WORD]: what a kernel that walks `cost` entries
    // of every tile stages -- accumulated where the counting pass reads every tile's cost once anyway)
    uint32_t staged_sum = 0;
    bool counting = true;
    auto bucket = [&](uint32_t t) -> uint32_t {
        uint32_t c = cost_in ? cost_in[t] : (ranges[t].y - ranges[t].x);      // (a measured cost never exceeds the list length)
        if (counting) staged_sum += c;
        if (times_ranges) {            // the point pass of the opacity-field query: #points of the tile x (entries its pixels walked + a fixed per-point share)
            const unsigned long long m = (unsigned long long)(times_ranges[t].y - times_ranges[t].x) * (unsigned long long)(c + 32u);
            c = m > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)m;
        }
        if (c < 2u) return NB - 1u - c;                                            // 0, 1 -> the last buckets
        const uint32_t m = 31u - (uint32_t)__builtin_clz(c);
        const uint32_t b = 2u * m + ((c >> (m - 1u)) & 1u);                        // 2 .. 63: half-octave classes (costs of a class within 3:2)
        return NB - 1u - b;
    };
    // wave w owns tile ids [w * chunk, (w + 1) * chunk), walked 64 at a time; the first 8 steps keep their bucket in registers
    const uint32_t chunk = ((ntiles + NW - 1) / NW + 63u) & ~63u;
    const uint32_t t0 = wave * chunk, t1 = min(ntiles, t0 + chunk);
    const uint32_t steps = t1 > t0 ? (t1 - t0 + 63u) / 64u : 0u;
    // Counting without LDS atomics: in a uniform scene all 64 lanes of a wave hold the same one or two classes, and a same-address
    // ds_add serialises its lanes (~3.5 cycles each; 16 waves x 7 steps of it were 10 of this kernel's 15 us).  The 7 ballots that
    // rank a tile among the same-class lanes of its step give the count as well: the group's lowest lane adds it, plainly (a wave
    // owns its row of s_wc, a step has one leader per class).  Rank and count of the first 8 steps stay in registers for the placement.
    constexpr int KEEP = 8;
    const uint64_t lt = (1ull << lane) - 1ull;
    auto match = [&](bool live, uint32_t bkt, uint32_t& rank, uint32_t& lead_cnt) {      // whole wave
        uint64_t peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 7; bit++) {
            const uint64_t bal = __ballot((bkt >> bit) & 1u);
            peers &= ((bkt >> bit) & 1u) ? bal : ~bal;
        }
        rank = (uint32_t)__popcll(peers & lt);
        lead_cnt = (live && (peers & lt) == 0ull) ? (uint32_t)__popcll(peers) : 0u;     // > 0 only in the group's lowest lane
    };
    uint8_t mine[KEEP], rk[KEEP], lc[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; k++) { const uint32_t i = t0 + 64u * k + lane; mine[k] = ((uint32_t)k < steps && i < t1) ? (uint8_t)bucket(i) : (uint8_t)0xFF; }
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
        rk[k] = 0; lc[k] = 0;
        if ((uint32_t)k < steps) {                                               // steps is wave-uniform
            const uint32_t i = t0 + 64u * k + lane;
            uint32_t r_, c_;
            match(i < t1, mine[k], r_, c_);
            rk[k] = (uint8_t)r_; lc[k] = (uint8_t)c_;                             // (both <= 64)
            if (c_) s_wc[wave][mine[k]] += c_;
        }
    }
    for (uint32_t k = KEEP; k < steps; k++) {
        const uint32_t i = t0 + 64u * k + lane;
        const uint32_t bkt = i < t1 ? bucket(i) : 0xFFu;
        uint32_t r_, c_;
        match(i < t1, bkt, r_, c_);
        if (c_) s_wc[wave][bkt] += c_;
    }
    counting = false;
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) staged_sum += (uint32_t)__shfl_xor((int)staged_sum, o);
        if (lane == 0 && staged_sum) atomicAdd(&s_staged, staged_sum);
    }
    __syncthreads();
    if (tid == 0 && queue) queue[BW_STAGED_WORD] = s_staged;
    if (tid == 0 && staged_out) *staged_out = s_staged;          // (the backward's order: next to the mask pool's cursors, so that one copy brings the frame's counters to the host)
    // the frame's counters (GOF_USAGE_WORDS: the mask pool's cursors, final since the forward blend has ended, + the staged sum)
    // stored straight into the caller's host-mapped pinned memory: no copy launch behind the forward (two blit kernels, 9 us)
    if (usage_host) {
        if (tid <= POOL_SHARDS) __hip_atomic_store(usage_host + tid, cursors_in[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tid == 0) __hip_atomic_store(usage_host + POOL_SHARDS + 1, s_staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    {   // per bucket: exclusive prefix over the waves (ascending tile id), total n_b; then G_b = tiles in heavier buckets
        uint32_t run = 0, inc = 0;
        if (tid < NB) {
#pragma unroll
            for (int w = 0; w < NW; w++) { const uint32_t c = s_wc[w][tid]; s_wc[w][tid] = run; run += c; }
            s_n[tid] = run;
            inc = run;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)inc, o); if (lane >= (unsigned)o) inc += up; }
            if (tid == 63u) s_carry[0] = inc;
        }
        __syncthreads();
        if (tid < NB) s_g[tid] = inc - run + (tid >= 64u ? s_carry[0] : 0u);
    }
    __syncthreads();
    {   // queue position of every (XCD x, bucket b) share: thread (x, b), exclusive scan over b inside each group of 128 threads.
        // XCD x takes the eighth e = (x - G_b) mod 8 of bucket b: a bucket of fewer than 8 tiles continues the round where the
        // heavier buckets stopped (the handful of heaviest tiles does not pile up on XCD 0), a large one is split into 8 runs
        const uint32_t x = tid >> 7, b = tid & 127u;
        const uint32_t n = s_n[b];
        const uint32_t e = (x - s_g[b]) & 7u;
        const uint32_t lo = (e * n + 7u) >> 3, hi = ((e + 1u) * n + 7u) >> 3;      // ceil(e n / 8) .. ceil((e + 1) n / 8): ranks of the e-th eighth
        const uint32_t share = hi - lo;
        uint32_t inc = share;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)inc, o); if (lane >= (unsigned)o) inc += up; }
        __syncthreads();                                                           // (s_carry[0] of the previous step has been read)
        if ((tid & 127u) == 63u) s_carry[x] = inc;                                 // total of the group's first wave
        __syncthreads();
        const uint32_t before = (b >= 64u) ? s_carry[x] : 0u;
        s_off[x][b] = before + inc - share;
        if (b == 127u && queue) { queue[x] = 0u; queue[NXCD + x] = before + inc; }  // head, length
    }
    __syncthreads();
    auto place = [&](uint32_t i, bool live, uint32_t bkt, uint32_t rank, uint32_t lead_cnt) {
        if (live) {
            const uint32_t base = s_wc[wave][bkt];                                  // tiles of this class in earlier waves and earlier steps
            const uint32_t r = base + rank;                                         // stable rank of tile i inside its class
            if (lead_cnt) s_wc[wave][bkt] = base + lead_cnt;                        // the group's lowest lane advances the wave's base
            const uint32_t n = s_n[bkt];
            // the eighth this rank falls into: floor(8 r / n) = e  <=>  ceil(e n / 8) <= r < ceil((e + 1) n / 8) -- seven compares
            // against the thresholds instead of an integer division
            uint32_t e = 0;
#pragma unroll
            for (uint32_t k = 1; k < 8; k++) e += (r >= ((k * n + 7u) >> 3)) ? 1u : 0u;
            const uint32_t lo = (e * n + 7u) >> 3;
            const uint32_t x = (e + s_g[bkt]) & 7u;                                 // the XCD that takes this eighth of this class
            order[x * stride + s_off[x][bkt] + (r - lo)] = i;
        }
    };
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
        if ((uint32_t)k < steps) {
            const uint32_t i = t0 + 64u * k + lane;
            place(i, i < t1, mine[k], rk[k], lc[k]);
        }
    }
    for (uint32_t k = KEEP; k < steps; k++) {
        const uint32_t i = t0 + 64u * k + lane;
        const uint32_t bkt = i < t1 ? bucket(i) : 0xFFu;
        uint32_t r_, c_;
        match(i < t1, bkt, r_, c_);
        place(i, i < t1, bkt, r_, c_);
    }
}

// debug: the reference's 64-bit sort key of every sorted instance (tile << 32 | depth bits)
__global__ void __launch_bounds__(256)
rebuild_keys(uint32_t R, const uint32_t* __restrict__ tiles, const uint32_t* __restrict__ gids, const float* __restrict__ depths,
             uint64_t* __restrict__ keys)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    keys[i] = ((uint64_t)tiles[i] << 32) | (uint64_t)__float_as_uint(depths[gids[i]]);
}

} // namespace gof
