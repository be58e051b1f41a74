// blend_forward.hip -- K7, the tiled forward alpha-blend (replaces renderCUDA, reference
// forward.cu:409-612).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; wave w covers the 8x8 pixel quadrant (w & 1, w >> 1) (tile_pixel):
// a compact footprint meets fewer splats than the reference's 16x4 strip, and per-pixel results do not depend on the map.
//
// MI355X design -- the kernel is VALU-issue bound (two fp64 divisions and an exp per contributing pair), and a
// tile list is ~4x longer than the set of splats that actually reach a given pixel (at S1M a wave scans ~420 entries,
// a pixel keeps ~89 candidates, ~72 contribute).  A lock-step loop makes all 64 pixels of a wave pay the
// exact path whenever ANY of them needs it.  Instead, per staged batch of 256 entries:
//
//   phase 1 (cull scan): (a) lane = ENTRY, 64 entries at a time: the entry's conservative footprint box (the box of the
//     alpha >= 1/255 level-set ellipsoid, computed once per Gaussian in preprocess_fwd, see footprint_bbox) against the
//     wave's pixel rectangle -> a wave-uniform ballot of the entries that touch the wave at all; (b) lane = PIXEL, scalar
//     loop over those entries: the pixel's own box test, the fp32 prelude and the error-bounded cull
//     (pair_certainly_transparent); survivors are recorded as a 256-bit mask of ITS pixel in LDS (s_mask[word][thread]).
//   phase 2 (per-lane ordered consumption): every lane pops the next set bit of its own mask, reads
//     that entry's record from LDS with a per-lane address and runs the exact path (fp64 t /
//     min_value, exp, blend update).  Entries are consumed in ascending list order per pixel, so
//     the per-pixel operation sequence -- and therefore every output bit -- is unchanged; the number
//     of heavy iterations of a wave drops from "#entries any pixel passes" to "max #candidates of
//     one pixel" (~97 at S1M).  The entries that really contributed are written back into the mask and stored to the
//     binning workspace (contributor masks, cmask_base): the backward visits exactly those.
//
//  * tile-list entries are staged as whole 64-byte SplatRec lines (one aligned gather per entry,
//    colour included; the reference re-reads colour from global memory per contributing pair,
//    forward.cu:561).  LDS layout [4][256] float4: staging writes of 64 consecutive lanes are
//    contiguous (conflict-free), phase-1 reads are wave-uniform broadcasts.  The LDS copy carries the
//    entry's cull threshold next to view2gaussian (phase 1 reads 3 x 16 B).
//  * per-wave exit by ballot(done), workgroup exit by __syncthreads_and(done) (forward.cu:475-477).
//  * XCD-aware tile order (xcd_tile_id): neighbouring tiles, which gather the same records, run on
//    the same XCD and share its L2.
//
// Arithmetic: identical operation sequence to the oracle (fp32 products/sums in source order, fp64 for
// AA/BB/min_value, the mapped depth and the final distortion normalisation, forward.cu:504-557, 589):
// colour, depth, alpha, distortion, final_T and n_contrib are bit-identical to the oracle.  The one
// exception is the unit normal (hardware rsq instead of fp64 sqrt + 3 divisions): channels 3-5 agree
// to ~3e-7.
#include "gof_common.h"

namespace gof {

// scan/consume granularity inside a staged batch: smaller = finer early exit once a wave saturates,
// larger = better lane compaction in phase 2
#ifndef GOF_FW_CHUNK
#define GOF_FW_CHUNK 64      // measured A/B at S1M: 64 -> 1.066 ms, 128 -> 1.08 ms, 256 -> 1.14 ms
#endif
constexpr int FW_CHUNK = GOF_FW_CHUNK;

#ifdef GOF_STATS
// developer-only instrumentation (never in the shipped build): [0] scanned wave-entries, [1] candidate
// (lane, entry) pairs, [2] phase-2 wave iterations, [3] exact-pass pairs, [4] contributing pairs, [5] lane-iterations active
__device__ unsigned long long g_fw_stats[8];
#define STAT_ADD(i, v) atomicAdd(&g_fw_stats[i], (unsigned long long)(v))
#else
#define STAT_ADD(i, v)
#endif

__global__ void __launch_bounds__(256)
blend_forward(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
              const float4* __restrict__ bbox, int W, int H, float focal_x, float focal_y, const float* __restrict__ bg_color,
              float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
              uint32_t* __restrict__ cmask, uint32_t gx, uint32_t ntiles)
{
    const uint32_t tile = xcd_tile_id(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const uint32_t tx = tile % gx, ty = tile / gx;
    const uint32_t tid = threadIdx.x;
    uint32_t lx, ly;
    tile_pixel(tid, lx, ly);
    const uint32_t px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const float pixfx = (float)px + 0.5f, pixfy = (float)py + 0.5f;
    const float rx = (float)(((double)pixfx - W / 2.) / (double)focal_x);
    const float ry = (float)(((double)pixfy - H / 2.) / (double)focal_y);

    const uint2 range = ranges[tile];
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + TILE_PIX - 1) / TILE_PIX;

    // LDS record: q0 = v0..v3, q1 = v4..v7, q2 = {v8, v9, cull threshold, w}, q3 = {r, g, b, -}
    __shared__ float4 s_rec[4][TILE_PIX];
    __shared__ uint32_t s_mask[TILE_PIX / 32][TILE_PIX];
    __shared__ float4 s_box[TILE_PIX];
    uint32_t* const cm_tile = cmask + cmask_base(range.x, tile) * TILE_PIX;
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    // pixel rectangle of this wave (tile_pixel: wave w covers the 8x8 quadrant (w & 1, w >> 1)), inclusive bounds
    const float wave_x0 = (float)(tx * TILE_X + 8u * (wave & 1u)), wave_x1 = wave_x0 + 7.0f;
    const float wave_y0 = (float)(ty * TILE_Y + 8u * (wave >> 1)), wave_y1 = wave_y0 + 7.0f;

    bool done = !inside;
    float T = 1.0f;
    uint32_t last_contributor = 0, max_contributor = (uint32_t)-1;
    float C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0, Dp = 0, Al = 0;
    float dist1 = 0, dist2 = 0, distortion = 0;

    for (int i = 0; i < rounds; i++, toDo -= TILE_PIX) {
        if (__syncthreads_and(done)) break;
        const uint32_t k = range.x + (uint32_t)i * TILE_PIX + tid;
        if (k < range.y) {
            const uint32_t id = point_list[k];
            const float4* src = reinterpret_cast<const float4*>(&rec[id]);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            s_rec[0][tid] = a; s_rec[1][tid] = b;
            s_rec[2][tid] = make_float4(c.x, c.y, cull_log_threshold(c.z), c.z);
            s_rec[3][tid] = make_float4(c.w, d.x, d.y, 0.f);
            s_box[tid] = bbox[id];
        }
        __syncthreads();
        const int nwords_batch = (min(TILE_PIX, toDo) + 31) >> 5;
        if (__ballot(!done) == 0ull) {             // whole wave saturated: it only helps staging (and reports "no contributors")
            for (int q = 0; q < nwords_batch; q++) cm_tile[((size_t)i * (TILE_PIX / 32) + q) * TILE_PIX + tid] = 0u;
            continue;
        }

        const int n = min(TILE_PIX, toDo);
        const uint32_t base = (uint32_t)i * TILE_PIX;

        int words_valid = 0;                     // mask words of this batch that hold contributor bits (wave-uniform)
        for (int c0 = 0; c0 < n; c0 += FW_CHUNK) {
        if (__ballot(!done) == 0ull) break;
        const int cn = min(FW_CHUNK, n - c0);            // entries [c0, c0 + cn) of the staged batch
        const int w0 = c0 >> 5;
        const int nw = w0 + ((cn + 31) >> 5);

        // ---- phase 1: cull scan ----
        // (a) lane = ENTRY: 64 entries at a time against the wave's 8x8 pixel rectangle -> a wave-uniform 64-bit mask of the
        //     entries whose footprint box touches the wave at all (the others cost 1/64 instruction each instead of ~8);
        // (b) lane = PIXEL, scalar loop over the set bits: the pixel's own box test, the fp32 prelude and the error-bounded cull.
        for (int w = w0; w < nw; w += 2) {
            const int je = w * 32 + (int)lane;
            bool touch = false;
            if (je < n) {
                const float4 bx = s_box[je];
                touch = (wave_x1 >= bx.x) & (wave_x0 <= bx.y) & (wave_y1 >= bx.z) & (wave_y0 <= bx.w);
            }
            uint64_t m = __ballot(touch);
            uint32_t word_lo = 0, word_hi = 0;
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1ull;
                const int j = w * 32 + b;
                const float4 bx = s_box[j];
                const bool inbox = !done & (pxf >= bx.x) & (pxf <= bx.y) & (pyf >= bx.z) & (pyf <= bx.w);
                if (__ballot(inbox) == 0ull) continue;          // only saturated pixels of the wave lie inside
                const float4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j];
                const float v[10] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y };
                PairEval p;
                pair_prelude(v, rx, ry, p);
                const uint32_t pass = (inbox && !pair_certainly_transparent(p, q2.y, q2.z)) ? 1u : 0u;
                if (b < 32) word_lo |= pass << b; else word_hi |= pass << (b - 32);
            }
            s_mask[w][tid] = done ? 0u : word_lo;
            if (w + 1 < nw) s_mask[w + 1][tid] = done ? 0u : word_hi;
            if ((tid & 63) == 0) STAT_ADD(0, min(64, n - w * 32));
            STAT_ADD(1, done ? 0 : __popc(word_lo) + __popc(word_hi));
        }

        // ---- phase 2: every lane consumes its own candidates in list order ----
        int w = w0;
        uint32_t cur = s_mask[w0][tid];
        uint32_t cbits = 0;                              // contributor bits of word w (flushed when w advances)
        for (;;) {
            const bool more = !done && (cur != 0u || w + 1 < nw);
            if (__ballot(more) == 0ull) break;
            if ((tid & 63) == 0) STAT_ADD(2, 1);
            if (!more) continue;
            STAT_ADD(5, 1);
            if (cur == 0u) { s_mask[w][tid] = cbits; cbits = 0; w++; cur = s_mask[w][tid]; }   // a consumed word becomes its contributor word
            if (cur == 0u) continue;
            const int b = __ffs((int)cur) - 1;
            cur &= cur - 1u;
            const int j = w * 32 + b;
            const uint32_t contributor = base + (uint32_t)j + 1u;   // 1-based list position (forward.cu:497)

            const float4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j];
            const float v[10] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y };
            PairEval p;
            pair_prelude(v, rx, ry, p);
            pair_exact(v, q2.w, p);
            if (p.skip) continue;
            STAT_ADD(3, 1);
            const float alpha = p.alpha, t = p.t;
            const float test_T = T * (1 - alpha);
            if (test_T < 0.0001f) { done = true; continue; }
            STAT_ADD(4, 1);

            const float4 q3 = s_rec[3][j];
            const float max_t = t;
            const float mapped_max_t = (float)((GOF_FAR_PLANE * max_t - GOF_FAR_PLANE * GOF_NEAR_PLANE) / ((GOF_FAR_PLANE - GOF_NEAR_PLANE) * max_t));
            // unit normal -n / |n|: the reference takes an fp64 sqrt and three IEEE divisions (forward.cu:548-549);
            // here one v_rsq_f32 (<= 1 ulp).  Only the normal channels depend on it: they agree with the oracle to
            // ~3e-7 instead of bit for bit; every other output is unaffected.
            const float inv_len = __builtin_amdgcn_rsqf(p.n0 * p.n0 + p.n1 * p.n1 + p.n2 * p.n2 + 1e-7f);
            const float nn0 = -p.n0 * inv_len, nn1 = -p.n1 * inv_len, nn2 = -p.n2 * inv_len;

            const float A = 1 - T;
            const float error = mapped_max_t * mapped_max_t * A + dist2 - 2 * mapped_max_t * dist1;
            distortion += error * alpha * T;
            dist1 += mapped_max_t * alpha * T;
            dist2 += mapped_max_t * mapped_max_t * alpha * T;

            C0 += q3.x * alpha * T;
            C1 += q3.y * alpha * T;
            C2 += q3.z * alpha * T;
            N0 += nn0 * alpha * T;
            N1 += nn1 * alpha * T;
            N2 += nn2 * alpha * T;
            if (T > 0.5f) { Dp = t; max_contributor = contributor; }
            Al += alpha * T;
            T = test_T;
            last_contributor = contributor;
            cbits |= 1u << b;
        }
        s_mask[w][tid] = cbits;
        for (int q = w + 1; q < nw; q++) s_mask[q][tid] = 0u;       // candidate words this pixel never reached (it saturated)
        words_valid = nw;
        }   // chunk
        for (int q = 0; q < nwords_batch; q++)
            cm_tile[((size_t)i * (TILE_PIX / 32) + q) * TILE_PIX + tid] = (q < words_valid) ? s_mask[q][tid] : 0u;
    }

    if (inside) {
        const size_t HW = (size_t)W * H;
        const float distortion_before_normalized = distortion;
        distortion = (float)((double)distortion / ((double)((1 - T) * (1 - T)) + 1e-7));
        final_T[pix_id] = T;
        final_T[pix_id + HW] = dist1;
        final_T[pix_id + 2 * HW] = dist2;
        final_T[pix_id + 3 * HW] = distortion_before_normalized;
        n_contrib[pix_id] = last_contributor;
        n_contrib[pix_id + HW] = max_contributor;
        out_color[0 * HW + pix_id] = C0 + T * bg_color[0];
        out_color[1 * HW + pix_id] = C1 + T * bg_color[1];
        out_color[2 * HW + pix_id] = C2 + T * bg_color[2];
        out_color[3 * HW + pix_id] = N0;
        out_color[4 * HW + pix_id] = N1;
        out_color[5 * HW + pix_id] = N2;
        out_color[6 * HW + pix_id] = Dp;
        out_color[7 * HW + pix_id] = Al;
        out_color[8 * HW + pix_id] = distortion;
    }
}

#ifdef GOF_STATS
extern "C" int gof_debug_fw_stats(unsigned long long* out8, int reset)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fw_stats), sizeof(g_fw_stats));
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_fw_stats), z, sizeof(z)); }
    return 0;
}
#endif

} // namespace gof
