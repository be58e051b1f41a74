// blend_backward.hip -- K8, the per-pixel backward of the blend (replaces backward renderCUDA,
// reference backward.cu:634-955).
//
// Same tile / pixel decomposition as blend_forward.  Each tile list is traversed back to front.
//
// MI355X design:
//  * The reference issues 17 global fp32 atomicAdd per contributing (pixel, splat) pair
//    (backward.cu:836, 905-912, 943-952).  Here all 64 lanes of a wave (an 8x8 pixel quadrant) look at the SAME splat
//    at the same time, so the 17 partial gradients are reduced in three levels:
//      1. in registers, per 16-lane row (= 8x2 pixels): a TRANSPOSED (butterfly) reduction of 16 of the values -- the
//         two quad exchange steps halve the number of live values, two row rotations finish -- plus a plain 4-step
//         DPP sum of the 17th (row_sum16_transposed / row_sum);
//      2. lanes 0..3 of every row that contributed add their 4 sums each (and lane 15 the 17th) into a per-batch LDS
//         accumulator s_acc[17][BATCH] with ds_add_f32: 5 LDS instructions per splat instead of 17;
//      3. after the batch, thread j flushes entry j with ONE global atomic per component and only if
//         some pixel of the tile contributed: <= 17 atomics per (tile, splat) instead of up to
//         17 x 256, issued 64 lanes wide.
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
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace gof {

constexpr int NGRAD = 17;   // colour 3, mean2D 3, opacity 1, view2gaussian 10
// tile-list entries staged per batch.  128 (not 256) keeps the LDS footprint at ~21 KB so that
// occupancy is bounded by registers (4 waves/SIMD), not by LDS.
#ifndef GOF_BW_BATCH
#define GOF_BW_BATCH 128
#endif
constexpr int BATCH = GOF_BW_BATCH;
static_assert(BATCH == 128 || BATCH == 256, "BATCH must be 128 or 256");

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
// Transposed ("butterfly") row reduction of 16 values: each exchange step halves the number of live values instead of summing
// every value at every step.  After the two quad steps lane (b1 b0) holds, for m = 0..3, the quad sum of value 4m + 2 b1 + b0;
// two row rotations complete the sum over the row's four quads.  44 VALU instead of 64, and the 16 results sit in 4 lanes
// (4 each) instead of 16 values in one lane: 4 LDS adds instead of 16.
__device__ __forceinline__ void row_sum16_transposed(const float* g, bool b0, bool b1, float w4[4])
{
    float u[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const float keep = b0 ? g[2 * m + 1] : g[2 * m];
        const float send = b0 ? g[2 * m] : g[2 * m + 1];
        u[m] = keep + dpp_get<0xB1>(send);          // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const float keep = b1 ? u[2 * m + 1] : u[2 * m];
        const float send = b1 ? u[2 * m] : u[2 * m + 1];
        float w = keep + dpp_get<0x4E>(send);       // quad_perm [2,3,0,1]
        w = w + dpp_get<0x124>(w);                  // row_ror:4
        w = w + dpp_get<0x128>(w);                  // row_ror:8
        w4[m] = w;
    }
}

__global__ void __launch_bounds__(256)
blend_backward(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
               const float4* __restrict__ conic, const uint32_t* __restrict__ cmask, int W, int H, float focal_x, float focal_y,
               const float* __restrict__ bg_color, const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
               const float* __restrict__ dL_dpixels, float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity,
               float* __restrict__ dL_dcolors, float* __restrict__ dL_dv2g, uint32_t gx, uint32_t ntiles)
{
    const uint32_t tile = xcd_tile_id(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
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
    const float pxm = (float)px, pym = (float)py;   // pixf - 0.5 (backward.cu:770), exact

    const uint2 range = ranges[tile];
    const uint32_t* const cm_tile = cmask + cmask_base(range.x, tile) * TILE_PIX;

    __shared__ float4 s_rec[4][BATCH];
    __shared__ float4 s_conic[BATCH];
    __shared__ uint32_t s_id[BATCH];
    __shared__ float s_acc[NGRAD][BATCH];
    __shared__ uint32_t s_touched[BATCH];
    __shared__ uint32_t s_cm[BATCH / 32][TILE_PIX];
    __shared__ uint32_t s_max_last;

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
    if (max_last == 0) return;

    float acc0 = 0, acc1 = 0, acc2 = 0;          // accum_rec
    float lc0 = 0, lc1 = 0, lc2 = 0;             // last_color
    float an0 = 0, an1 = 0, an2 = 0;             // accum_normal_rec
    float ln0 = 0, ln1 = 0, ln2 = 0;             // last_normal
    float last_alpha = 0;
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);
    // mapped depth (2DGS NDC mapping, forward.cu:545): m(t) = (FAR t - FAR NEAR) / ((FAR-NEAR) t)
    const float MAP_A = (float)(GOF_FAR_PLANE / (GOF_FAR_PLANE - GOF_NEAR_PLANE));
    const float MAP_B = (float)(GOF_FAR_PLANE * GOF_NEAR_PLANE / (GOF_FAR_PLANE - GOF_NEAR_PLANE));

    // batches are aligned to the list start (so that they line up with the forward's mask words) and visited from
    // the back: batch kb covers list positions [kb * BATCH, min((kb + 1) * BATCH, max_last))
    const int nbatches = (int)((max_last + BATCH - 1) / BATCH);
    for (int kb = nbatches - 1; kb >= 0; kb--) {
        __syncthreads();
        const uint32_t p0 = (uint32_t)kb * BATCH;
        const int n = (int)min((uint32_t)BATCH, max_last - p0);
        {
            // staging: BATCH entries by 256 threads -> (256 / BATCH) threads share one 64-byte record
            constexpr int TPE = TILE_PIX / BATCH;            // threads per entry: 1 or 2
            constexpr int F4 = 4 / TPE;                      // float4 per thread
            const uint32_t e = tid / TPE, part = tid % TPE;
            if ((int)e < n) {
                const uint32_t id = point_list[range.x + p0 + e];
                const float4* src = reinterpret_cast<const float4*>(&rec[id]) + part * F4;
#pragma unroll
                for (int q = 0; q < F4; q++) s_rec[part * F4 + q][e] = src[q];
                if (part == 0) {
                    s_conic[e] = conic[id];
                    s_id[e] = id;
                }
            }
            const int nw = (n + 31) >> 5;
#pragma unroll
            for (int q = 0; q < BATCH / 32; q++)
                s_cm[q][tid] = (q < nw) ? cm_tile[((size_t)(p0 >> 5) + q) * TILE_PIX + tid] : 0u;
            for (int k = tid; k < NGRAD * BATCH; k += TILE_PIX) (&s_acc[0][0])[k] = 0.f;
            if (tid < BATCH) s_touched[tid] = 0;
        }
        __syncthreads();

        for (int w = ((n + 31) >> 5) - 1; w >= 0; w--) {
            const uint32_t word = s_cm[w][tid];
            if (__ballot(word != 0u) == 0ull) continue;            // no pixel of this wave has a contributor in these 32 entries
            for (int b = 31; b >= 0; b--) {
                const bool contrib = (word >> b) & 1u;
                if (__ballot(contrib) == 0ull) continue;
                const int j = w * 32 + b;
                const uint32_t contributor = p0 + (uint32_t)j;       // 0-based list position (backward.cu:763)

                float g[NGRAD];
#pragma unroll
                for (int k = 0; k < NGRAD; k++) g[k] = 0.f;
                if (contrib) {
                    const float4 a = s_rec[0][j], bq = s_rec[1][j], c = s_rec[2][j], d = s_rec[3][j];
                    const float v[10] = { a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w, c.x, c.y };
                    const float wgt = c.z;
                    PairEval p;
                    pair_prelude(v, rx, ry, p);
                    pair_exact_backward(v, wgt, p);                 // alpha, G, t, q to ~5e-7 of the forward's values (no decision depends on them here)
                    // Everything below is gradient arithmetic held to a tolerance (the reference rounds every term to fp32 before its
                    // atomicAdd and accumulates in arbitrary order), not to bit-identity: let the compiler contract mul+add into FMA
                    // here.  alpha, G, t, q above keep the forward's exact (uncontracted) arithmetic -- they are ill-conditioned.
                    {
#pragma clang fp contract(fast)
                    const float4 con = s_conic[j];
                    const float G = p.G, alpha = p.alpha;
                    const float dx = d.z - pxm, dy = d.w - pym;

                    const float t = p.t;
                    const float inv_t = __builtin_amdgcn_rcpf(t);
                    const float mapped_max_t = MAP_A - MAP_B * inv_t;
                    const float dmax_t_dd = MAP_B * inv_t * inv_t;
                    const float len2 = p.n0 * p.n0 + p.n1 * p.n1 + p.n2 * p.n2 + 1e-7f;
                    const float inv_len = __builtin_amdgcn_rsqf(len2);
                    const float nn0 = -p.n0 * inv_len, nn1 = -p.n1 * inv_len, nn2 = -p.n2 * inv_len;

                    // recurrence of backward.cu:816 (T = T / (1 - alpha)) with the hardware reciprocal (1 ulp) that the background
                    // term needs anyway: <= 1.5 ulp per step instead of 0.5, over ~70 steps -> 1e-5 relative at worst
                    const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                    T = T * inv_1ma;
                    const float dchannel_dcolor = alpha * T;

                    float dL_dalpha = 0.0f;
                    {
                        const float c0 = c.w, c1 = d.x, c2 = d.y;
                        acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = c0;
                        dL_dalpha += (c0 - acc0) * dpx0;
                        g[0] = dchannel_dcolor * dpx0;
                        acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = c1;
                        dL_dalpha += (c1 - acc1) * dpx1;
                        g[1] = dchannel_dcolor * dpx1;
                        acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = c2;
                        dL_dalpha += (c2 - acc2) * dpx2;
                        g[2] = dchannel_dcolor * dpx2;
                    }
                    // distortion: only dL_dmax_t survives (dL_dweight is detached, backward.cu:848-852)
                    const float dL_dmax_t = 2.0f * (T * alpha) * (mapped_max_t * final_A - final_D) * dL_dreg * dmax_t_dd;

                    float dnn0, dnn1, dnn2;
                    an0 = last_alpha * ln0 + (1.f - last_alpha) * an0; ln0 = nn0;
                    dL_dalpha += (nn0 - an0) * dn0;
                    dnn0 = dchannel_dcolor * dn0;
                    an1 = last_alpha * ln1 + (1.f - last_alpha) * an1; ln1 = nn1;
                    dL_dalpha += (nn1 - an1) * dn1;
                    dnn1 = dchannel_dcolor * dn1;
                    an2 = last_alpha * ln2 + (1.f - last_alpha) * an2; ln2 = nn2;
                    dL_dalpha += (nn2 - an2) * dn2;
                    dnn2 = dchannel_dcolor * dn2;

                    // d(-n/|n|): dL_dn = (-dnn + (dnn . n) n / |n|^2) / |n|
                    const float dL_dlength = (dnn0 * p.n0 + dnn1 * p.n1 + dnn2 * p.n2) * (inv_len * inv_len);
                    float dL_dn0 = (-dnn0 + dL_dlength * p.n0) * inv_len;
                    float dL_dn1 = (-dnn1 + dL_dlength * p.n1) * inv_len;
                    float dL_dn2 = (-dnn2 + dL_dlength * p.n2) * inv_len;

                    float dL_dt = dL_dmax_t;
                    if (contributor == max_contributor - 1u) dL_dt += dL_dmax_depth;

                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final * inv_1ma) * bg_dot_dpixel;

                    const float dL_dG = wgt * dL_dalpha;
                    const float gdx = G * dx;
                    const float gdy = G * dy;
                    const float dG_ddelx = -gdx * con.x - gdy * con.y;
                    const float dG_ddely = -gdy * con.z - gdx * con.y;
                    g[3] = dL_dG * dG_ddelx * ddelx_dx;
                    g[4] = dL_dG * dG_ddely * ddely_dy;
                    g[5] = fabsf(g[3]) + fabsf(g[4]);
                    g[6] = G * dL_dalpha;

                    // min_value = CC - (BB/AA)(BB/4), t = -BB/(2AA)
                    const float qf = (float)p.q;
                    const float dL_dmin_value = -0.5f * (dL_dG * G);
                    const float inv2AA = 0.5f * __builtin_amdgcn_rcpf(p.AAf);
                    const float dL_dA = dL_dmin_value * (qf * qf) * 0.25f + dL_dt * qf * inv2AA;
                    const float dL_dB = -0.5f * (dL_dmin_value * qf) - dL_dt * inv2AA;
                    dL_dn0 += dL_dA * rx;
                    dL_dn1 += dL_dA * ry;
                    dL_dn2 += dL_dA;
                    g[7] = dL_dn0 * rx;
                    g[8] = dL_dn0 * ry + dL_dn1 * rx;
                    g[9] = dL_dn0 + dL_dn2 * rx;
                    g[10] = dL_dn1 * ry;
                    g[11] = dL_dn1 + dL_dn2 * ry;
                    g[12] = dL_dn2;
                    g[13] = dL_dB * 2 * rx;
                    g[14] = dL_dB * 2 * ry;
                    g[15] = dL_dB * 2;
                    g[16] = dL_dmin_value;
                    }
                }
                // rows (16 lanes = 8x2 pixels) in which no pixel contributed hold exact zeros: skip their LDS adds
                const uint64_t cmask64 = __ballot(contrib);
                const bool row_hit = ((cmask64 >> (lane & 48u)) & 0xFFFFull) != 0ull;
                float w4[4];
                row_sum16_transposed(g, (lane & 1u) != 0u, (lane & 2u) != 0u, w4);
                const float g16 = row_sum(g[16]);
                if ((lane & 12u) == 0u && row_hit) {            // lanes 0..3 of the row: values 4m + (lane & 3)
                    float* dst = &s_acc[lane & 3u][j];
#pragma unroll
                    for (int m = 0; m < 4; m++) unsafeAtomicAdd(dst + (size_t)m * 4 * BATCH, w4[m]);
                }
                if ((lane & 15u) == 15u && row_hit) {
                    unsafeAtomicAdd(&s_acc[16][j], g16);
                    s_touched[j] = 1u;
                }
            }
        }
        __syncthreads();
        // flush: one entry per thread, one global atomic per component, 64 lanes wide
        if ((int)tid < n && s_touched[tid]) {
            const size_t id = s_id[tid];
            unsafeAtomicAdd(&dL_dcolors[id * 3 + 0], s_acc[0][tid]);
            unsafeAtomicAdd(&dL_dcolors[id * 3 + 1], s_acc[1][tid]);
            unsafeAtomicAdd(&dL_dcolors[id * 3 + 2], s_acc[2][tid]);
            unsafeAtomicAdd(&dL_dmean2D[id * 3 + 0], s_acc[3][tid]);
            unsafeAtomicAdd(&dL_dmean2D[id * 3 + 1], s_acc[4][tid]);
            unsafeAtomicAdd(&dL_dmean2D[id * 3 + 2], s_acc[5][tid]);
            unsafeAtomicAdd(&dL_dopacity[id], s_acc[6][tid]);
            float* gv = dL_dv2g + id * 10;
#pragma unroll
            for (int k = 0; k < 10; k++) unsafeAtomicAdd(gv + k, s_acc[7 + k][tid]);
        }
    }
}

} // namespace gof
