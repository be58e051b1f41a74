// blend_backward.hip -- K8, the per-pixel backward of the blend (replaces backward renderCUDA,
// reference backward.cu:634-955).
//
// Same tile / pixel decomposition as blend_forward.  Each tile list is traversed back to front.
//
// MI355X design:
//  * The reference issues 17 global fp32 atomicAdd per contributing (pixel, splat) pair
//    (backward.cu:836, 905-912, 943-952).  Here all 64 lanes of a wave (an 8x8 pixel quadrant) look at the SAME splat
//    at the same time, and the partial gradients are summed IN REGISTERS over the whole wave (16 of the 17: the last one,
//    dL_dview2gaussian[9] = -0.5 w G dL_dalpha, is -0.5 w times the opacity gradient with w a constant of the Gaussian):
//      1. every lane stores its 16 values into a [value][lane] LDS panel of its wave, in two halves of 8 (contiguous ds_write2_b32);
//      2. lane (v, p) = (lane / 8, lane % 8) reads back lanes 8 p .. 8 p + 7 of value v (two ds_read_b128), adds them, three DPP steps
//         finish: lanes 0, 8, 16, ... hold the wave totals of the half's 8 values;
//      3. ONE plain LDS store per half (8 active lanes) puts the totals into the wave's own slab
//         s_slab[wave][16][BATCH]: a wave visits an entry at most once per batch, so nothing is read back and no LDS
//         atomic is needed (measured: ds_add_f32 costs ~3.5 cycles per active LANE; the former 68 lane-atomics per
//         entry and wave kept the LDS busy ~70 % of the time);
//      4. after the batch, thread j adds the slabs of the waves that visited entry j (in wave order) and STORES the 16 sums as
//         the partial gradient of that (tile, Gaussian) instance -- plain stores into a 64-byte record (one line) of a POOL (one record per
//         staged instance; a wave takes its batch's slots with one atomic on the pool's cursor), whose slot + 1 goes into a word
//         indexed by an instance number e in which the instances of one Gaussian, and of consecutive Gaussians, are consecutive.  No
//         global atomic at all: gather_tile_partials (below) then adds the records of every Gaussian in ascending e.  The reference's 17
//         atomicAdd per contributing pair (and round 1's 45 M memory-side float atomics per frame, 795 MB of "writes" for
//         67 MB of accumulators) are gone, and the gradients are bit-reproducible from run to run.
//  * Entries behind the last contributor of EVERY pixel of the tile are never staged: the
//    traversal starts at max-over-tile(last_contributor) (the reference stages the full list and
//    skips per pixel, backward.cu:763-765).
//  * WHICH (pixel, entry) pairs contribute is not re-derived: blend_forward leaves one bit per pixel and
//    list position (contributor masks, see cmask_base); a wave visits only the entries in which one of
//    its pixels has the bit set.  (The set is the one the reference's backward re-discovers with its position /
//    alpha-threshold tests, backward.cu:763-805.)
//  * Since no decision depends on the backward's alpha any more, only its VALUE does: the fp32 prelude (normal, AA, BB)
//    is kept bit-identical to the forward's -- min_value is ill-conditioned in it -- but the quotient BB/AA is formed as an
//    fp32 hi + lo pair and min_value from it (pair_exact_backward), exp by v_exp_f32, the transmittance recurrence with
//    the hardware reciprocal, and the gradient formulas are compiled with FMA contraction (#pragma clang fp contract(fast)
//    on that block only).  The reference itself rounds every per-pair term to fp32 before its atomicAdd and accumulates
//    in arbitrary order; measured against the oracle's double accumulation: <= 1.6e-6 of the gradient maximum.
//
// Gradient semantics reproduced exactly (they are the training signal): dL_dweight is detached
// (backward.cu:851-852) so only dL_dmax_t carries the distortion gradient; the alpha channel's
// incoming gradient (channel 7) is never read; a 0.99-clamped alpha still back-propagates;
// the depth gradient goes only to contributor == max_contributor-1 (backward.cu:880-882);
// dL_dmean2D.z accumulates |.| (backward.cu:908-909).
#include "gof_common.h"

namespace gof {

constexpr int NGRAD = 16;   // colour 3, mean2D 3, opacity 1, view2gaussian 0..8 -- reduced per (wave, entry); view2gaussian 9 follows from the opacity total
// tile-list entries staged per batch.  64 keeps the LDS footprint (records 6 KB, masks 2 KB, four per-wave slabs 16 KB) at
// 25 KB, so that occupancy is bounded by registers (5 waves/SIMD), not by LDS.
#ifndef GOF_BW_BATCH
#define GOF_BW_BATCH 64
#endif
constexpr int BATCH = GOF_BW_BATCH;
static_assert(BATCH == 32 || BATCH == 64 || BATCH == 128, "BATCH must be 32, 64 or 128");
// How the 16 per-pair values are summed over the wave: through the LDS (round 4), in two halves of 8 values: every lane stores 8 values
// into a [value][lane] panel of its wave (ds_write2_b32, contiguous per instruction), then lane (v, p) = (lane / 8, lane % 8) reads back
// the 8 lanes 8 p .. 8 p + 7 of value v with two ds_read_b128, adds them, and three DPP steps finish: 18 VALU instructions per half; the
// transposition is done by the LDS crossbar, which this kernel leaves idle otherwise.  Row stride 68 words (b128 lane groups on distinct
// 16-byte slots: MI355X_MICROARCH.md, LDS).  8.5 KB more LDS: 31.5 KB, 5 workgroups per CU.
// Measured (profiles/r04_ab_call2_*.txt, S1M / S1M-clustered) against rounds 2-3's transposed reduction IN REGISTERS (v_permlane32_swap /
// v_permlane16_swap, DPP quad levels with selects, row rotations: ~75 VALU instructions of which 12 lane-group swaps at ~12 cycles, ~300
// of the trip's 749 SIMD-cycles; removed in round 6): 1.162 / 1.543 ms -> 1.070 / 1.373 ms (-8 % / -11 %).  The ISA count promised more
// (746 -> 598 SIMD-cycles per trip): an LDS store moves its source registers through the SIMD's register read ports (~2 cycles per
// dword), so the 16 stored values cost about what the 12 swaps did.  Also measured, not kept: all 16 values through one 17 KB panel
// (1.091 / 1.396, 4 workgroups per CU); that software-pipelined over the visits -- panel of visit k read back during visit k + 1's
// arithmetic (1.082 / 1.393: the round trip was not what limited it); lanes without a contribution storing a zero register instead of
// every lane clearing 16 registers first (1.174 / 1.521: eight more store instructions cost more than 16 v_mov); the batch's records
// staged by quads of lanes over all four waves instead of by one thread per entry (round 5: 1.075 vs 1.058 / 1.399 vs 1.364 ms,
// profiles/r05_ab_call6_binning.txt -- with five workgroups per CU resident the idle waves at the staging barrier cost nothing another
// workgroup cannot use; removed in round 6).
constexpr int RED_STRIDE = 68;
#ifdef GOF_STATS
// developer-only instrumentation (never in the shipped build): [0] wave iterations of the entry loop, [1] (row, iteration) pairs with
// an entry to visit, [2] contributing (lane, entry) pairs, [3] word fetches, [4] staged entries, [5] pairs of CONSECUTIVE visited
// entries of a wave whose contributing lanes are disjoint (greedy, within one mask word: what merging two entries into one gradient
// block could save), [6] (row, entry) pairs with at least one contributing lane (a walk per 16-lane row would visit those),
// [7] visited entries in which at most 32 lanes contribute, [8] sum over the batches of the LARGEST visit count among the tile's four
// waves (what the batch's barriers make every wave wait for), [9] batches, [10] sum over the tiles of the largest per-tile visit total
// among the four waves (what the tile would take if its waves ran without the batch barriers)
__device__ unsigned long long g_bw_stats[12];
#define BSTAT_ADD(i, v) atomicAdd(&g_bw_stats[i], (unsigned long long)(v))
#else
#define BSTAT_ADD(i, v)
#endif

#ifdef GOF_TILE_CLOCK
__device__ unsigned long long g_bw_tile_clock[2][1 << 16];
#define TILE_CLOCK_START() const unsigned long long _t0 = wall_clock64()
#define TILE_CLOCK_END(ARR) do { if (threadIdx.x == 0 && tile < (1u << 16)) { ARR[0][tile] = _t0; ARR[1][tile] = wall_clock64(); } } while (0)
#else
#define TILE_CLOCK_START()
#define TILE_CLOCK_END(ARR)
#endif

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(moved);
}
// sum over each 16-lane row; afterwards EVERY lane of the row holds the row total
__device__ __forceinline__ float row_sum(float v)
{
    v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);    // row_half_mirror
    v = dpp_add<0x140>(v);    // row_mirror
    return v;
}

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// OR over the 16-lane row; every lane of the row gets the result
__device__ __forceinline__ uint32_t row_or(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);    // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);    // row_mirror
    return v;
}
// OR over the wave as a wave-uniform (scalar) value
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    v = row_or(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) | (uint32_t)__builtin_amdgcn_readlane((int)v, 16) |
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) | (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ float lane_xor(float v, uint32_t lane, uint32_t mask)   // value of lane (lane ^ mask), through the LDS crossbar (no memory access)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane ^ mask) << 2), __float_as_int(v)));
}

// one tile; called by all 256 threads of the workgroup (persistent loop in blend_backward)
__device__ __forceinline__ void
blend_backward_tile(const uint32_t tile, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
                    const float4* __restrict__ conic, const MaskPool masks, int W, int H, float focal_x, float focal_y,
                    const float* __restrict__ bg_color, const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
                    const float* __restrict__ dL_dpixels, const uint2* __restrict__ rect, const uint32_t* __restrict__ inst_off,
                    float4* __restrict__ part16, uint32_t* __restrict__ slot_of, uint32_t* __restrict__ rec_cursor,
                    uint32_t rec_cap, uint32_t gx)
{
    TILE_CLOCK_START();
    const uint32_t tx = tile % gx, ty = tile / gx;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint32_t lx, ly;
    tile_pixel(tid, lx, ly);
    const uint32_t px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const size_t HW = (size_t)W * H;
    const float pixfx = (float)px + 0.5f, pixfy = (float)py + 0.5f;
    const float rx = (float)(((double)pixfx - W / 2.) / (double)focal_x);
    const float ry = (float)(((double)pixfy - H / 2.) / (double)focal_y);
    const f2 RX = { rx, rx }, RY = { ry, ry }, RXY = { rx, ry }, RYX = { ry, rx };
    const f2 PXM = { (float)px, (float)py };       // pixf - 0.5 (backward.cu:770), exact

    const uint2 range = ranges[tile];
    const uint32_t* const mask_entry = masks.table + mask_slot0(range.x, tile);      // this tile's chunks of the mask pool, one per forward batch of 256 entries

    // LDS record of a staged entry, arranged as the operand pairs of the packed arithmetic (v = view2gaussian):
    //   q0 = {v0, v1 | v1, v3}, q1 = {v2, v4 | v2, v6}, q2 = {v4, v7 | v5, v8}   (prelude, as in blend_forward)
    //   q3 = {CC, w | r, g}, q4 = {b, - | mean2D.x, mean2D.y}, q5 = {conic.x, conic.z | conic.y, conic.y}
    __shared__ f4 s_rec[6][BATCH];
    __shared__ uint32_t s_inst[BATCH];                 // instance index of the staged (tile, Gaussian) pair
    // per wave: the wave totals of the entries it visited in this batch.  Rows padded by GOF_BW_SLAB_PAD words: the 8 lanes that store a
    // visit's totals write 8 ROWS at the same column j -- with a row of exactly 64 words they all hit bank j (an 8-way conflict on both
    // stores of every visit: half of this kernel's bank-conflict cycles).  Measured: 1.068 -> 1.064 ms at S1M, 1.370 -> 1.360 ms clustered.
    // (Also measured, profiles/r04_ab_call6_*.txt: 32 staged entries per batch -- 20 KB of LDS, 79 VGPRs, SIX waves per SIMD -- is
    // slower, 1.090 / 1.418 ms: the batch's barriers and staging cost more than the sixth wave hides.)
#ifndef GOF_BW_SLAB_PAD
#define GOF_BW_SLAB_PAD 1
#endif
    __shared__ float s_slab[4][NGRAD][BATCH + GOF_BW_SLAB_PAD];
    __shared__ uint32_t s_vis[4][BATCH / 32];           // per wave: which entries those are
    __shared__ uint32_t s_max_last;
    __shared__ __attribute__((aligned(16))) float s_red[4][NGRAD / 2][RED_STRIDE];

    const float T_final = inside ? final_Ts[pix_id] : 0;
    float T = T_final;
    const float final_D = inside ? final_Ts[pix_id + HW] : 0;
    const float final_A = 1 - T_final;
    const float dL_dreg = inside ? dL_dpixels[8 * HW + pix_id] : 0;
    const uint32_t last_contributor = inside ? n_contrib[pix_id] : 0;
    const uint32_t max_contributor = inside ? n_contrib[pix_id + HW] : 0;
    float dpx0 = 0, dpx1 = 0, dpx2 = 0, dn0 = 0, dn1 = 0, dn2 = 0, dL_dmax_depth = 0;
    if (inside) {
        dpx0 = dL_dpixels[0 * HW + pix_id]; dpx1 = dL_dpixels[1 * HW + pix_id]; dpx2 = dL_dpixels[2 * HW + pix_id];
        dn0 = dL_dpixels[3 * HW + pix_id]; dn1 = dL_dpixels[4 * HW + pix_id]; dn2 = dL_dpixels[5 * HW + pix_id];
        dL_dmax_depth = dL_dpixels[6 * HW + pix_id];
    }
    float bg_dot_dpixel = 0;
    bg_dot_dpixel += bg_color[0] * dpx0;
    bg_dot_dpixel += bg_color[1] * dpx1;
    bg_dot_dpixel += bg_color[2] * dpx2;
    // incoming gradients as the pairs the channel recurrences run on: (colour 0, 1), (colour 2, normal 2), (normal 0, 1)
    const f2 DP01 = { dpx0, dpx1 }, DP2N2 = { dpx2, dn2 }, DN01 = { dn0, dn1 };

    // tile-wide maximum of last_contributor: nothing behind it is ever used
    if (tid == 0) s_max_last = 0;
    __syncthreads();
    {
        uint32_t m = last_contributor;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if (lane == 0) atomicMax(&s_max_last, m);
    }
    __syncthreads();
    const uint32_t max_last = min(s_max_last, range.y - range.x);
    if (max_last == 0) { TILE_CLOCK_END(g_bw_tile_clock); return; }
    // this tile's block of the record pool: one record slot per staged list position (sum over the tiles = what gof_backward_query
    // reports, so a scratch sized from it always has room; a smaller one loses the tile's records: the caller that sized it from
    // earlier frames compares the staged count with its pool afterwards and repeats the backward, include/gof_hip.h)
    __shared__ uint32_t s_rec_base;
    if (tid == 0) {
        const uint32_t b0 = atomicAdd(rec_cursor, max_last);
        s_rec_base = (b0 + max_last <= rec_cap) ? b0 : POOL_NONE;        // no room: this tile's records are dropped (the cursor still counts them)
    }
    __syncthreads();
    const uint32_t rec_base = s_rec_base;

    // accum_rec / accum_normal_rec of backward.cu:824-837, 862-867 for the channel pairs (colour 0, 1), (colour 2, normal 2),
    // (normal 0, 1).  The reference folds the PREVIOUS pair into them at the start of a pair (last_alpha, last_color); here the
    // current pair is folded in at its end -- the same operation on the same operands, one iteration earlier, so last_alpha /
    // last_color / last_normal (7 more loop-carried registers) are not needed.
    f2 acc01 = { 0, 0 }, acc2n = { 0, 0 }, accn01 = { 0, 0 };
    const f2 DDEL = { (float)(0.5 * W), (float)(0.5 * H) };        // ddelx_dx, ddely_dy
    // mapped depth (2DGS NDC mapping, forward.cu:545): m(t) = (FAR t - FAR NEAR) / ((FAR-NEAR) t)
    const float MAP_A = (float)(GOF_FAR_PLANE / (GOF_FAR_PLANE - GOF_NEAR_PLANE));
    const float MAP_B = (float)(GOF_FAR_PLANE * GOF_NEAR_PLANE / (GOF_FAR_PLANE - GOF_NEAR_PLANE));

    // batches are aligned to the list start (so that they line up with the forward's mask words) and visited from
    // the back: batch kb covers list positions [kb * BATCH, min((kb + 1) * BATCH, max_last))
    const int nbatches = (int)((max_last + BATCH - 1) / BATCH);
#ifdef GOF_STATS
    __shared__ uint32_t s_stat_v[4];
    uint32_t stat_tile_visits = 0;
#endif
    for (int kb = nbatches - 1; kb >= 0; kb--) {
        __syncthreads();
        const uint32_t p0 = (uint32_t)kb * BATCH;
        const int n = (int)min((uint32_t)BATCH, max_last - p0);
        {
            // staging: one thread per entry reads the 64-byte record + the 2D conic and writes the pair layout
            if ((int)tid < n) {
                const uint32_t id = point_list[range.x + p0 + tid];
                const float4* src = reinterpret_cast<const float4*>(&rec[id]);
                const float4 a = src[0], b = src[1], c = src[2], d = src[3];
                const float4 co = conic[id];
                s_rec[0][tid] = f4{ a.x, a.y, a.y, a.w };
                s_rec[1][tid] = f4{ a.z, b.x, a.z, b.z };
                s_rec[2][tid] = f4{ b.x, b.w, b.y, c.x };
                s_rec[3][tid] = f4{ c.y, c.z, c.w, d.x };
                s_rec[4][tid] = f4{ d.y, 0.f, d.z, d.w };
                s_rec[5][tid] = f4{ co.x, co.z, co.y, co.y };
                // index of this (tile, Gaussian) instance: instances are numbered Gaussian by Gaussian (in id order), within a Gaussian
                // row by row over its tile rectangle: e = first(id) + (ty - miny) * w + (tx - minx)
                const uint2 rc = rect[id];
                s_inst[tid] = inst_off[id] + (ty - (rc.x >> 16)) * (rc.y & 0xFFFFu) + (tx - (rc.x & 0xFFFFu));
            }
        }
        // the pixel's contributor words of this batch: read by their own thread only -- registers, not LDS
        uint32_t cmw[BATCH / 32];
        {
            const int nw = (n + 31) >> 5;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)mask_entry[p0 >> 8]);
            // (POOL_NONE: the forward's pool had no room for this batch -- the caller repeats the frame before it uses these gradients,
            // gof_hip.h; nothing is read then)
            const uint32_t* const src = masks.pool + ((size_t)(c == POOL_NONE ? 0u : c) * 4u + (tid >> 6)) * MASK_SUBCHUNK_WORDS + ((p0 & 255u) >> 5) * 64u + lane;
#pragma unroll
            for (int q = 0; q < BATCH / 32; q++)
                cmw[q] = (q < nw && c != POOL_NONE) ? src[q * 64] : 0u;
        }
        __syncthreads();
        if (tid == 0) BSTAT_ADD(4, n);
#ifdef GOF_STATS
        uint32_t stat_visits = 0;
#endif

        // The wave walks the union of its pixels' contributors back to front (wave-uniform entry index: scalar bit walk over
        // the OR of the 64 mask words, LDS reads of the record are broadcasts).
        const uint32_t wave = tid >> 6;
#pragma unroll
        for (int w = BATCH / 32 - 1; w >= 0; w--) {
            if (w > ((n + 31) >> 5) - 1) continue;                 // (wave-uniform; the loop is unrolled so that cmw[] stays in registers)
            const uint32_t word = cmw[w];
            uint32_t todo = wave_or(word);
            uint32_t visited = todo;
#ifdef GOF_STATS
            stat_visits += (uint32_t)__popc(todo);
#endif
#ifdef GOF_STATS
            unsigned long long stat_prev_set = 0ull;
            {   // [1] trips a walk per 16-lane row would need for this word (each row its own union: the slowest row counts), [3] the wave's
                const uint32_t ru = row_or(word);
                const int c0 = __popc((uint32_t)__builtin_amdgcn_readlane((int)ru, 0)), c1 = __popc((uint32_t)__builtin_amdgcn_readlane((int)ru, 16));
                const int c2 = __popc((uint32_t)__builtin_amdgcn_readlane((int)ru, 32)), c3 = __popc((uint32_t)__builtin_amdgcn_readlane((int)ru, 48));
                if (lane == 0) { BSTAT_ADD(1, max(max(c0, c1), max(c2, c3))); BSTAT_ADD(3, __popc(todo)); }
            }
#endif
            while (todo) {
                const int b = 31 - __builtin_clz(todo);
                todo &= ~(1u << b);
                const bool contrib = (word >> b) & 1u;
                const int j = w * 32 + b;
                const uint32_t contributor = p0 + (uint32_t)j;       // 0-based list position (backward.cu:763)
                if (lane == 0) BSTAT_ADD(0, 1);
#ifdef GOF_STATS
                {
                    const unsigned long long cur_set = __ballot(contrib);
                    if (lane == 0) {
                        if (stat_prev_set != 0ull && (stat_prev_set & cur_set) == 0ull) { BSTAT_ADD(5, 1); stat_prev_set = 0ull; }   // merged: neither pairs again
                        else stat_prev_set = cur_set;
                        BSTAT_ADD(6, ((cur_set & 0xFFFFull) != 0) + ((cur_set & 0xFFFF0000ull) != 0) + ((cur_set & 0xFFFF00000000ull) != 0) + ((cur_set >> 48) != 0));
                        if (__popcll(cur_set) <= 32) BSTAT_ADD(7, 1);
                    }
                }
#endif

                float g[NGRAD];
                // (Measured and dropped, round 4: EVERY lane running the gradient block with alpha = G = 0 and harmless stand-ins for a
                // lane without a contribution -- T and the accumulators stay, all 16 values come out as zeros -- instead of 16 v_mov and
                // the divergent branch: 11 VALU instructions fewer per trip (ISA count 598 -> 572 cycles), blend_backward 1.065 vs
                // 1.067 ms at S1M, 1.379 vs 1.364 ms clustered: no difference.  profiles/r04_ab_call5_*.txt)
#pragma unroll
                for (int k = 0; k < NGRAD; k++) g[k] = 0.f;
                if (contrib) {
                    BSTAT_ADD(2, 1);
                    const f4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j], q3 = s_rec[3][j], q4 = s_rec[4][j], q5 = s_rec[5][j];
                    // fp32 prelude in the forward's operation order (min_value is ill-conditioned in it), two values per instruction
                    const f2 n01 = (q0.xy * RX + q0.zw * RY) + q1.xy;               // normal[0], normal[1]
                    const f2 n2b = (q1.zw * RX + q2.xy * RY) + q2.zw;               // normal[2], BB / 2
                    const f2 rn = RXY * n01;
                    PairEval p;
                    p.n0 = n01.x; p.n1 = n01.y; p.n2 = n2b.x;
                    p.AAf = (rn.x + rn.y) + n2b.x;
                    p.BBf = 2 * n2b.y;
                    const float wgt = q3.y;
                    pair_exact_backward_cc(q3.x, wgt, p);           // alpha, G, t, q to ~5e-7 of the forward's values (no decision depends on them here)
                    // Everything below is gradient arithmetic held to a tolerance (the reference rounds every term to fp32 before its
                    // atomicAdd and accumulates in arbitrary order), not to bit-identity: FMA contraction allowed, packed where two
                    // values see the same operation.  alpha, G, t, q above keep the forward's exact (uncontracted) prelude.
                    {
#pragma clang fp contract(fast)
                    const float G = p.G, alpha = p.alpha;
                    const f2 dxy = q4.zw - PXM;

                    const float t = p.t;
                    const float inv_t = __builtin_amdgcn_rcpf(t);
                    const float mapped_max_t = MAP_A - MAP_B * inv_t;
                    const float dmax_t_dd = MAP_B * inv_t * inv_t;
                    const f2 sq = n01 * n01;
                    const float len2 = (sq.x + sq.y) + p.n2 * p.n2 + 1e-7f;
                    const float inv_len = __builtin_amdgcn_rsqf(len2);
                    const f2 NINV = { -inv_len, -inv_len };
                    const f2 nn01 = n01 * NINV;
                    const f2 c2n = { q4.x, p.n2 * NINV.x };         // colour 2 | unit normal 2

                    // recurrence of backward.cu:816 (T = T / (1 - alpha)) with the hardware reciprocal (1 ulp) that the background
                    // term needs anyway: <= 1.5 ulp per step instead of 0.5, over ~70 steps -> 1e-5 relative at worst
                    const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                    T = T * inv_1ma;
                    const float dchannel_dcolor = alpha * T;
                    const f2 DCD = { dchannel_dcolor, dchannel_dcolor };

                    // the six channel terms of dL_dalpha (backward.cu:824-837, 862-867) on three register pairs, then this pair is
                    // folded into the accumulators for the next one (in front of it)
                    f2 da2 = (q3.zw - acc01) * DP01;
                    da2 += (c2n - acc2n) * DP2N2;
                    da2 += (nn01 - accn01) * DN01;
                    float dL_dalpha = da2.x + da2.y;
                    const f2 AL = { alpha, alpha }, OMA = { 1.f - alpha, 1.f - alpha };
                    acc01 = AL * q3.zw + OMA * acc01;
                    acc2n = AL * c2n + OMA * acc2n;
                    accn01 = AL * nn01 + OMA * accn01;
                    const f2 g01 = DCD * DP01;                      // dL_dcolour 0, 1
                    const f2 g2d = DCD * DP2N2;                     // dL_dcolour 2 | dL_dnormal_normalized 2
                    const f2 dnn01 = DCD * DN01;
                    g[0] = g01.x; g[1] = g01.y; g[2] = g2d.x;
                    const float dnn2 = g2d.y;

                    // distortion: only dL_dmax_t survives (dL_dweight is detached, backward.cu:848-852)
                    const float dL_dmax_t = 2.0f * (T * alpha) * (mapped_max_t * final_A - final_D) * dL_dreg * dmax_t_dd;

                    // d(-n/|n|): dL_dn = (-dnn + (dnn . n) n / |n|^2) / |n|
                    const f2 dd = dnn01 * n01;
                    const float dL_dlength = ((dd.x + dd.y) + dnn2 * p.n2) * (inv_len * inv_len);
                    const f2 DLEN = { dL_dlength, dL_dlength }, INV = { inv_len, inv_len };
                    f2 dLn01 = (DLEN * n01 - dnn01) * INV;
                    float dL_dn2 = (dL_dlength * p.n2 - dnn2) * inv_len;

                    float dL_dt = dL_dmax_t;
                    if (contributor == max_contributor - 1u) dL_dt += dL_dmax_depth;

                    dL_dalpha *= T;
                    dL_dalpha += (-T_final * inv_1ma) * bg_dot_dpixel;

                    const float dL_dG = wgt * dL_dalpha;
                    // dG_ddelx = -G dx con.x - G dy con.y, dG_ddely = -G dy con.z - G dx con.y   (backward.cu:897-909)
                    const f2 gd = dxy * f2{ G, G };
                    const f2 dG = gd * q5.xy + gd.yx * q5.zw;       // negated below
                    const f2 g34 = dG * DDEL * f2{ -dL_dG, -dL_dG };
                    g[3] = g34.x; g[4] = g34.y;
                    g[5] = fabsf(g34.x) + fabsf(g34.y);
                    g[6] = G * dL_dalpha;

                    // min_value = CC - (BB/AA)(BB/4), t = -BB/(2AA)
                    const float qf = p.qf;
                    const float dL_dmin_value = -0.5f * (dL_dG * G);
                    const float inv2AA = 0.5f * __builtin_amdgcn_rcpf(p.AAf);
                    const float dL_dA = dL_dmin_value * (qf * qf) * 0.25f + dL_dt * qf * inv2AA;
                    const float dL_dB2 = -(dL_dmin_value * qf) - 2.0f * (dL_dt * inv2AA);      // 2 dL_dB
                    dLn01 += f2{ dL_dA, dL_dA } * RXY;
                    dL_dn2 += dL_dA;
                    const f2 g710 = dLn01 * RXY;                    // dL_dv2g 0, 3
                    const f2 cr = dLn01 * RYX;
                    const f2 g911 = dLn01 + f2{ dL_dn2, dL_dn2 } * RXY;   // dL_dv2g 2, 4
                    const f2 g1314 = f2{ dL_dB2, dL_dB2 } * RXY;    // dL_dv2g 6, 7
                    g[7] = g710.x;
                    g[8] = cr.x + cr.y;
                    g[9] = g911.x;
                    g[10] = g710.y;
                    g[11] = g911.y;
                    g[12] = dL_dn2;
                    g[13] = g1314.x;
                    g[14] = g1314.y;
                    g[15] = dL_dB2;
                    // dL_dv2g[9] = dL_dmin_value = -0.5 wgt (G dL_dalpha) = -0.5 wgt g[6] with wgt a constant of the Gaussian: its total is
                    // formed from the total of g[6] at the flush, not reduced here
                    }
                }
                // The wave's own panel: no other wave touches it, and the LDS serves a wave's instructions in order -- the wave
                // barriers only keep the compiler from moving the accesses across (no instruction).
                {
                    float* const panel = &s_red[wave][0][0];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
#pragma unroll
                        for (int k = 0; k < NGRAD / 2; k++) panel[k * RED_STRIDE + lane] = g[h * (NGRAD / 2) + k];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const f4* const rp = reinterpret_cast<const f4*>(panel + (lane >> 3) * RED_STRIDE + 8u * (lane & 7u));
                        const f4 a = rp[0], b = rp[1];
                        float tot = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
                        tot = tot + dpp_get<0xB1>(tot);               // quad_perm [1,0,3,2]
                        tot = tot + dpp_get<0x4E>(tot);               // quad_perm [2,3,0,1]
                        tot = tot + dpp_get<0x141>(tot);              // row_half_mirror: the other quad of the 8 lanes
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if ((lane & 7u) == 0u) s_slab[wave][h * (NGRAD / 2) + (lane >> 3)][j] = tot;
                    }
                }
            }
            if (lane == 0) s_vis[wave][w] = visited;
        }
#ifdef GOF_STATS
        stat_tile_visits += stat_visits;
        if (lane == 0) s_stat_v[wave] = stat_visits;
        __syncthreads();
        if (tid == 0) { BSTAT_ADD(8, max(max(s_stat_v[0], s_stat_v[1]), max(s_stat_v[2], s_stat_v[3]))); BSTAT_ADD(9, 1); }
#endif
        __syncthreads();
        // flush: one entry per thread -- the slabs of the waves that visited it, summed in wave order, stored as the partial
        // gradient record of this (tile, Gaussian) instance.  Records live in a POOL (round 4: the scratch holds a record per STAGED
        // instance -- ~30 % of R at S1M -- not per instance): the tile took a block of max_last records when it started (one atomic
        // per tile: rec_base above), the entry at list position p writes record rec_base + p and leaves that + 1 in
        // slot_of[instance] (0 = no record), which gather_tile_partials reads where it read a validity byte before.  The values a
        // Gaussian receives do not depend on WHERE its records lie: bit-reproducible.
        if ((int)tid < n) {
            const uint32_t qw = tid >> 5, qb = 1u << (tid & 31u);
            const bool v0 = (s_vis[0][qw] & qb) != 0u, v1 = (s_vis[1][qw] & qb) != 0u, v2 = (s_vis[2][qw] & qb) != 0u, v3 = (s_vis[3][qw] & qb) != 0u;
            if ((v0 | v1 | v2 | v3) && rec_base != POOL_NONE) {
                const uint32_t slot = rec_base + p0 + tid;            // the tile's block of the pool, indexed by list position
                auto total = [&](int k) {
                    float x = v0 ? s_slab[0][k][tid] : 0.f;
                    x += v1 ? s_slab[1][k][tid] : 0.f;
                    x += v2 ? s_slab[2][k][tid] : 0.f;
                    x += v3 ? s_slab[3][k][tid] : 0.f;
                    return x;
                };
                float4* dst = part16 + (size_t)slot * 4;
#pragma unroll
                for (int q = 0; q < 4; q++) dst[q] = make_float4(total(4 * q), total(4 * q + 1), total(4 * q + 2), total(4 * q + 3));
                // (dL_dv2g[9]'s share of this record, -0.5 wgt x the total of value 6 with wgt a constant of the Gaussian, is formed by
                // gather_tile_partials from the record's value 6 -- until round 5 a 17th value stored here, in an array of its own: a
                // second random sector per record for the flush and for the gather)
                slot_of[s_inst[tid]] = slot + 1u;
            }
        }
    }
#ifdef GOF_STATS
    __syncthreads();
    if ((tid & 63u) == 0u) s_stat_v[tid >> 6] = stat_tile_visits;
    __syncthreads();
    if (tid == 0) BSTAT_ADD(10, max(max(s_stat_v[0], s_stat_v[1]), max(s_stat_v[2], s_stat_v[3])));
#endif
    TILE_CLOCK_END(g_bw_tile_clock);
}

// one workgroup per tile, popped as the workgroup starts: deepest walk first (pop_tile, gof_common.h; order by the forward's tile_cost)
#define GOF_BW_MIN_WAVES 5
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GOF_BW_MIN_WAVES, 8)))
blend_backward(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
               const float4* __restrict__ conic, const MaskPool masks, int W, int H, float focal_x, float focal_y,
               const float* __restrict__ bg_color, const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
               const float* __restrict__ dL_dpixels, const uint2* __restrict__ rect, const uint32_t* __restrict__ inst_off,
               float4* __restrict__ part16, uint32_t* __restrict__ slot_of, uint32_t* __restrict__ rec_cursor,
               uint32_t rec_cap, uint32_t gx, uint32_t ntiles,
               const uint32_t* __restrict__ tile_order, uint32_t* __restrict__ tile_queue, const uint32_t* __restrict__ tile_lens)
{
    __shared__ uint32_t s_tile;
    const uint32_t tile = pop_tile(tile_order, tile_queue, tile_lens, ntiles, &s_tile);
    if (tile >= ntiles) return;
    blend_backward_tile(tile, ranges, point_list, rec, conic, masks, W, H, focal_x, focal_y, bg_color, final_Ts, n_contrib, dL_dpixels, rect,
                        inst_off, part16, slot_of, rec_cursor, rec_cap, gx);
}

// Sum of the partial gradient records of every Gaussian over its tile instances, in ascending instance order (deterministic):
// writes dL_dcolors [P,3], dL_dmean2D [P,3], dL_dopacity [P], dL_dview2gaussian [P,10] completely (zeros for Gaussians no pixel
// blended), so none of them needs a memset.  A QUAD of lanes per Gaussian; instances are numbered in Gaussian-id order, so
// slot words, counts, offsets and outputs are all visited in ascending address order (the records themselves lie where the backward's
// waves happened to allocate them: one 64-byte line each); lane q of the quad owns values
// 4q .. 4q+3 of the record, so a record is ONE 64-byte line read by four adjacent lanes (a thread per Gaussian issued four fully
// address-divergent 16-byte loads per record and a divergent byte load per instance: 0.245 ms, bound by the address units, not by
// bytes); the slot words are read by the quad together (lane q takes instances k0 + q and k0 + 4 + q) and handed round by DPP.
// (Adding 0.0f for an absent record leaves every bit of the sum unchanged, so the result is that of the plain ordered loop.)
__global__ void __launch_bounds__(256)
gather_tile_partials(int P, const uint32_t* __restrict__ inst_off, const uint32_t* __restrict__ tiles_touched,
                     const float4* __restrict__ part16, const float4* __restrict__ conic_w, const uint32_t* __restrict__ slot_of,
                     float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors, float* __restrict__ dL_dv2g)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 2;
    const uint32_t q = (uint32_t)t & 3u;
    const bool live = i < P;                              // (no early return: the wave-wide section below wants all 64 lanes)
    const uint32_t g = (uint32_t)i;
    const uint32_t n = live ? tiles_touched[g] : 0u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // dL_dview2gaussian[9] = dL_dmin_value summed over the pairs = -0.5 wgt (G dL_dalpha) summed, wgt = opacity x the 3D filter's coefficient, a
    // constant of the Gaussian (preprocess_fwd leaves it in the spare word of the 2D conic): each record's share is -0.5 wgt x its
    // opacity sum (value 6: word z of the quarter lane 1 of the quad loads).  Until round 5 the flush stored that product as a 17th
    // value in an array of its own -- a second random sector per record for the flush and for this kernel; formed here from the same
    // operands in the same order, it and its sum (acc17, in lane 1) have the bits they had
    float acc17 = 0.f;
    float wm = (n ? -0.5f * conic_w[g].w : 0.f);          // (a culled Gaussian's conic is never written)
    constexpr uint32_t BIG = 64;          // Gaussians covering more tiles than this are summed by the whole wave (a near, huge splat can
                                          // cover thousands of tiles: left to its own quad it alone determined the kernel's duration)
    const size_t e0 = n ? inst_off[g] : 0;
    // 8 instances per trip: both validity loads first, then all (up to 8) record loads back to back, then the adds in instance order
#define GOF_GATHER_TRIP(E0, N, K0)                                                                                   \
    {                                                                                                                  \
        const uint32_t ka = (K0) + q, kb = (K0) + 4u + q;                                                              \
        const int va = (ka < (N)) ? (int)slot_of[(E0) + ka] : 0;                                                       \
        const int vb = (kb < (N)) ? (int)slot_of[(E0) + kb] : 0;                                                       \
        GOF_GATHER_TRIP_V(E0, K0, va, vb)                                                                              \
    }
#define GOF_GATHER_TRIP_V(E0, K0, va, vb)                                                                            \
    {                                                                                                                  \
        float4 r[8];                                                                                                   \
        GOF_GATHER_LOAD(E0, K0, 0, va, 0x00) GOF_GATHER_LOAD(E0, K0, 1, va, 0x55) GOF_GATHER_LOAD(E0, K0, 2, va, 0xAA) GOF_GATHER_LOAD(E0, K0, 3, va, 0xFF) \
        GOF_GATHER_LOAD(E0, K0, 4, vb, 0x00) GOF_GATHER_LOAD(E0, K0, 5, vb, 0x55) GOF_GATHER_LOAD(E0, K0, 6, vb, 0xAA) GOF_GATHER_LOAD(E0, K0, 7, vb, 0xFF) \
        _Pragma("unroll") for (int j = 0; j < 8; j++) { acc.x += r[j].x; acc.y += r[j].y; acc.z += r[j].z; acc.w += r[j].w; acc17 += wm * r[j].z; }      \
    }
#define GOF_GATHER_LOAD(E0, K0, J, V, CTRL)                                                                          \
        {                                                                                                              \
            const uint32_t sl = (uint32_t)__builtin_amdgcn_update_dpp(0, V, CTRL, 0xf, 0xf, false);   /* quad_perm [j,j,j,j]: slot + 1 of instance (E0) + (K0) + J, 0 = no record */ \
            const bool ok = sl != 0u;                                                                                  \
            const size_t e = (size_t)(sl - 1u);                                                                        \
            r[J] = ok ? part16[e * 4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);                                           \
        }
    if (n && n <= BIG) {
        // the slot words of the NEXT trip are requested before this trip's records: a trip then waits for one memory round trip,
        // not two (0.095 -> 0.086 ms; pipelining over several Gaussians per quad as well was measured slower, 0.097 ms)
        int va = (q < n) ? (int)slot_of[e0 + q] : 0;
        int vb = (4u + q < n) ? (int)slot_of[e0 + 4u + q] : 0;
        for (uint32_t k0 = 0; k0 < n; k0 += 8) {
            const uint32_t na = k0 + 8u + q, nb = k0 + 12u + q;
            const int va_next = (na < n) ? (int)slot_of[e0 + na] : 0;
            const int vb_next = (nb < n) ? (int)slot_of[e0 + nb] : 0;
            GOF_GATHER_TRIP_V(e0, k0, va, vb)
            va = va_next; vb = vb_next;
        }
    }
    // the big ones, one after the other, by all 16 quads of the wave: quad c takes trips c, c + 16, ...; the 16 partial sums are
    // added in a fixed tree (row rotations, then across the rows), i.e. still bit-reproducible
    uint64_t big = __ballot(n > BIG && q == 0u);
    const uint32_t lane = threadIdx.x & 63u;
    while (big) {
        const int owner = __builtin_ctzll(big);                      // lane 4 * (owner quad)
        big &= big - 1ull;
        const uint32_t bn = (uint32_t)__builtin_amdgcn_readlane((int)n, owner);
        const size_t be0 = ((size_t)(uint32_t)__builtin_amdgcn_readlane((int)(e0 >> 32), owner) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e0, owner);
        float4 keep = acc; float keep17 = acc17; const float keep_wm = wm;
        acc = make_float4(0.f, 0.f, 0.f, 0.f); acc17 = 0.f;
        wm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(keep_wm), owner));      // the owner's weight, for every quad that helps
        for (uint32_t k0 = 8u * (lane >> 2); k0 < bn; k0 += 8u * 16u) GOF_GATHER_TRIP(be0, bn, k0)
        wm = keep_wm;
        float v5[5] = { acc.x, acc.y, acc.z, acc.w, acc17 };
#pragma unroll
        for (int c = 0; c < 5; c++) {
            float x = v5[c];
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, false));   // row_ror:4  (quads of the row)
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));   // row_ror:8
            x += __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane ^ 16u) << 2), __float_as_int(x)));
            x += __int_as_float(__builtin_amdgcn_ds_bpermute((int)((lane ^ 32u) << 2), __float_as_int(x)));
            v5[c] = x;
        }
        const bool mine = (int)(lane & ~3u) == owner;
        acc = mine ? make_float4(v5[0], v5[1], v5[2], v5[3]) : keep;
        acc17 = mine ? v5[4] : keep17;
    }
#undef GOF_GATHER_LOAD
#undef GOF_GATHER_TRIP_V
#undef GOF_GATHER_TRIP
    // record layout (blend_backward's flush): [colour 0-2, mean2D 0 | mean2D 1-2, opacity, v2g 0 | v2g 1-4 | v2g 5-8]; v2g 9: acc17 of lane 1
    const float v2g9 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc17), 0x55, 0xf, 0xf, false));     // quad_perm [1,1,1,1]
    if (!live) return;
    const size_t o = (size_t)g;
    if (q == 0u) {
        dL_dcolors[o * 3 + 0] = acc.x; dL_dcolors[o * 3 + 1] = acc.y; dL_dcolors[o * 3 + 2] = acc.z;
        dL_dmean2D[o * 3 + 0] = acc.w;
        dL_dv2g[o * 10 + 9] = v2g9;
    } else if (q == 1u) {
        dL_dmean2D[o * 3 + 1] = acc.x; dL_dmean2D[o * 3 + 2] = acc.y;
        dL_dopacity[o] = acc.z;
        dL_dv2g[o * 10 + 0] = acc.w;
    } else {
        float* d = dL_dv2g + o * 10 + (q == 2u ? 1 : 5);
        d[0] = acc.x; d[1] = acc.y; d[2] = acc.z; d[3] = acc.w;
    }
}

#ifdef GOF_TILE_CLOCK
extern "C" int gof_debug_bw_tile_clock(unsigned long long* out, int ntiles)      // out[2][ntiles]: start, end
{
    (void)hipDeviceSynchronize();
    if (ntiles > (1 << 16)) return -1;
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bw_tile_clock), sizeof(unsigned long long) * ntiles, 0);
    (void)hipMemcpyFromSymbol(out + ntiles, HIP_SYMBOL(g_bw_tile_clock), sizeof(unsigned long long) * ntiles, sizeof(unsigned long long) * (1 << 16));
    return 0;
}
#endif
#ifdef GOF_STATS
extern "C" int gof_debug_bw_stats(unsigned long long* out12, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_bw_stats), sizeof(g_bw_stats));
    if (reset) { unsigned long long z[12] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bw_stats), z, sizeof(z)); }
    return 0;
}
#endif

} // namespace gof
