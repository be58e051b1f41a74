// blend_backward.hip -- K8, the per-pixel backward of the blend (replaces backward renderCUDA,
// reference backward.cu:634-955).
//
// Same tile / pixel decomposition as blend_forward.  Each tile list is traversed back to front.
//
// MI355X design:
//  * The reference issues 17 global fp32 atomicAdd per contributing (pixel, splat) pair
//    (backward.cu:836, 905-912, 943-952).  Here all 64 lanes of a wave look at the SAME splat
//    at the same time, so the 17 partial gradients are first summed across the wave with DPP
//    adds in registers (quad_perm / row_mirror / row_bcast: no LDS traffic) and only lane 63
//    issues one hardware global_atomic_add_f32 per component: <= 17 atomics per (wave, splat)
//    instead of up to 17 x 64.
//  * Entries behind the last contributor of EVERY pixel of the tile are never staged: the
//    traversal starts at max-over-tile(last_contributor) (the reference stages the full list
//    and skips per pixel, backward.cu:763-765).
//  * Staging as in the forward: whole 64-byte SplatRec lines + the 16-byte conic, LDS layout
//    [q][256] for conflict-free writes and broadcast reads.
//
// Gradient semantics reproduced exactly (they are the training signal): dL_dweight is detached
// (backward.cu:851-852) so only dL_dmax_t carries the distortion gradient; the alpha channel's
// incoming gradient (channel 7) is never read; a 0.99-clamped alpha still back-propagates;
// the depth gradient goes only to contributor == max_contributor-1 (backward.cu:880-882);
// dL_dmean2D.z accumulates |.| (backward.cu:908-909).
#include "gof_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace gof {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
// sum over the 64 lanes of a wave; the total is valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);    // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);    // row_mirror
    v = dpp_add<0x142, 0xa>(v);    // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xc>(v);    // row_bcast:31 -> rows 2,3
    return v;
}

__global__ void __launch_bounds__(256)
blend_backward(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
               const float4* __restrict__ conic, int W, int H, float focal_x, float focal_y, const float* __restrict__ bg_color,
               const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
               float* __restrict__ dL_dmean2D, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
               float* __restrict__ dL_dv2g, uint32_t gx, uint32_t ntiles)
{
    const uint32_t tile = xcd_tile_id(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const uint32_t tx = tile % gx, ty = tile / gx;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t px = tx * TILE_X + (tid % TILE_X), py = ty * TILE_Y + (tid / TILE_X);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const size_t HW = (size_t)W * H;
    const float pixfx = (float)px + 0.5f, pixfy = (float)py + 0.5f;
    const float rx = (float)(((double)pixfx - W / 2.) / (double)focal_x);
    const float ry = (float)(((double)pixfy - H / 2.) / (double)focal_y);
    const float pxm = (float)((double)pixfx - 0.5), pym = (float)((double)pixfy - 0.5);   // exact: pix + 0.5 - 0.5

    const uint2 range = ranges[tile];

    __shared__ float4 s_rec[4][TILE_PIX];
    __shared__ float4 s_conic[TILE_PIX];
    __shared__ uint32_t s_id[TILE_PIX];
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

    int toDo = (int)max_last;
    const int rounds = (toDo + TILE_PIX - 1) / TILE_PIX;
    uint32_t contributor = max_last;   // entries [max_last, toDo_full) are skipped by every pixel
    const uint32_t list_end = range.x + max_last;

    float acc0 = 0, acc1 = 0, acc2 = 0;          // accum_rec
    float lc0 = 0, lc1 = 0, lc2 = 0;             // last_color
    float an0 = 0, an1 = 0, an2 = 0;             // accum_normal_rec
    float ln0 = 0, ln1 = 0, ln2 = 0;             // last_normal
    float last_alpha = 0;
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);

    for (int i = 0; i < rounds; i++, toDo -= TILE_PIX) {
        __syncthreads();
        const uint32_t progress = (uint32_t)i * TILE_PIX + tid;
        if (range.x + progress < list_end) {
            const uint32_t id = point_list[list_end - progress - 1];
            const float4* src = reinterpret_cast<const float4*>(&rec[id]);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            s_rec[0][tid] = a; s_rec[1][tid] = b; s_rec[2][tid] = c; s_rec[3][tid] = d;
            s_conic[tid] = conic[id];
            s_id[tid] = id;
        }
        __syncthreads();

        const int n = min(TILE_PIX, toDo);
        for (int j = 0; j < n; j++) {
            contributor--;
            const bool active = inside && (contributor < last_contributor);
            if (__ballot(active) == 0ull) continue;

            float g_c0 = 0, g_c1 = 0, g_c2 = 0, g_mx = 0, g_my = 0, g_mz = 0, g_op = 0;
            float g_v0 = 0, g_v1 = 0, g_v2 = 0, g_v3 = 0, g_v4 = 0, g_v5 = 0, g_v6 = 0, g_v7 = 0, g_v8 = 0, g_v9 = 0;
            bool contrib = false;
            if (active) {
                const float4 a = s_rec[0][j], b = s_rec[1][j], c = s_rec[2][j];
                const float v[10] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y };
                const float w = c.z;
                PairEval p;
                eval_pair(v, w, rx, ry, p);
                if (!p.skip) {
                    contrib = true;
                    const float4 d = s_rec[3][j];
                    const float4 con = s_conic[j];
                    const float G = p.G, alpha = p.alpha;
                    const double AA = p.AA, BB = p.BB;
                    const float dx = d.z - pxm, dy = d.w - pym;

                    const float max_t = p.t;
                    const float mapped_max_t = (float)((GOF_FAR_PLANE * max_t - GOF_FAR_PLANE * GOF_NEAR_PLANE) / ((GOF_FAR_PLANE - GOF_NEAR_PLANE) * max_t));
                    const float dmax_t_dd = (float)((GOF_FAR_PLANE * GOF_NEAR_PLANE) / ((GOF_FAR_PLANE - GOF_NEAR_PLANE) * max_t * max_t));
                    const float length = (float)sqrt((double)(p.n0 * p.n0 + p.n1 * p.n1 + p.n2 * p.n2) + 1e-7);
                    const float nn0 = -p.n0 / length, nn1 = -p.n1 / length, nn2 = -p.n2 / length;

                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;

                    float dL_dalpha = 0.0f;
                    {
                        const float c0 = c.w, c1 = d.x, c2 = d.y;
                        acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = c0;
                        dL_dalpha += (c0 - acc0) * dpx0;
                        g_c0 = dchannel_dcolor * dpx0;
                        acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = c1;
                        dL_dalpha += (c1 - acc1) * dpx1;
                        g_c1 = dchannel_dcolor * dpx1;
                        acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = c2;
                        dL_dalpha += (c2 - acc2) * dpx2;
                        g_c2 = dchannel_dcolor * dpx2;
                    }
                    float dL_dmax_t = 0.0f;
                    dL_dmax_t += 2.0f * (T * alpha) * (mapped_max_t * final_A - final_D) * dL_dreg * dmax_t_dd;
                    // dL_dweight == 0 and last_dL_dT == 0 (backward.cu:852-858): "dL_dalpha += 0 - 0"
                    dL_dalpha += 0.f - 0.f;

                    float dnn0, dnn1, dnn2;
                    an0 = last_alpha * ln0 + (1.f - last_alpha) * an0; ln0 = nn0;
                    dL_dalpha += (nn0 - an0) * dn0;
                    dnn0 = alpha * T * dn0;
                    an1 = last_alpha * ln1 + (1.f - last_alpha) * an1; ln1 = nn1;
                    dL_dalpha += (nn1 - an1) * dn1;
                    dnn1 = alpha * T * dn1;
                    an2 = last_alpha * ln2 + (1.f - last_alpha) * an2; ln2 = nn2;
                    dL_dalpha += (nn2 - an2) * dn2;
                    dnn2 = alpha * T * dn2;

                    float dL_dlength = (dnn0 * p.n0 + dnn1 * p.n1 + dnn2 * p.n2);
                    dL_dlength *= 1.f / (length * length);
                    float dL_dn0 = (-dnn0 + dL_dlength * p.n0) / length;
                    float dL_dn1 = (-dnn1 + dL_dlength * p.n1) / length;
                    float dL_dn2 = (-dnn2 + dL_dlength * p.n2) / length;

                    float dL_dt = dL_dmax_t;
                    if (contributor == max_contributor - 1u) dL_dt += dL_dmax_depth;

                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                    const float dL_dG = w * dL_dalpha;
                    const float gdx = G * dx;
                    const float gdy = G * dy;
                    const float dG_ddelx = -gdx * con.x - gdy * con.y;
                    const float dG_ddely = -gdy * con.z - gdx * con.y;
                    g_mx = dL_dG * dG_ddelx * ddelx_dx;
                    g_my = dL_dG * dG_ddely * ddely_dy;
                    g_mz = fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy);
                    g_op = G * dL_dalpha;

                    const float dL_dpower = dL_dG * G;
                    const float dL_dmin_value = dL_dpower * -0.5f;
                    double dL_dA = dL_dmin_value * (BB / AA) * (BB / AA) / 4.f;
                    double dL_dB = dL_dmin_value * -BB / (2 * AA);
                    const double dL_dC = dL_dmin_value * 1.0f;
                    dL_dA += dL_dt * BB / (2 * AA * AA);
                    dL_dB += dL_dt * -1.f / (2 * AA);

                    dL_dn0 = (float)((double)dL_dn0 + dL_dA * rx);
                    dL_dn1 = (float)((double)dL_dn1 + dL_dA * ry);
                    dL_dn2 = (float)((double)dL_dn2 + dL_dA);

                    g_v0 = dL_dn0 * rx;
                    g_v1 = dL_dn0 * ry + dL_dn1 * rx;
                    g_v2 = dL_dn0 + dL_dn2 * rx;
                    g_v3 = dL_dn1 * ry;
                    g_v4 = dL_dn1 + dL_dn2 * ry;
                    g_v5 = dL_dn2;
                    g_v6 = (float)(dL_dB * 2 * rx);
                    g_v7 = (float)(dL_dB * 2 * ry);
                    g_v8 = (float)(dL_dB * 2);
                    g_v9 = (float)dL_dC;
                }
            }
            if (__ballot(contrib) == 0ull) continue;

            g_c0 = wave_sum_to_lane63(g_c0); g_c1 = wave_sum_to_lane63(g_c1); g_c2 = wave_sum_to_lane63(g_c2);
            g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my); g_mz = wave_sum_to_lane63(g_mz);
            g_op = wave_sum_to_lane63(g_op);
            g_v0 = wave_sum_to_lane63(g_v0); g_v1 = wave_sum_to_lane63(g_v1); g_v2 = wave_sum_to_lane63(g_v2);
            g_v3 = wave_sum_to_lane63(g_v3); g_v4 = wave_sum_to_lane63(g_v4); g_v5 = wave_sum_to_lane63(g_v5);
            g_v6 = wave_sum_to_lane63(g_v6); g_v7 = wave_sum_to_lane63(g_v7); g_v8 = wave_sum_to_lane63(g_v8);
            g_v9 = wave_sum_to_lane63(g_v9);
            if (lane == 63) {
                const size_t id = s_id[j];
                unsafeAtomicAdd(&dL_dcolors[id * 3 + 0], g_c0);
                unsafeAtomicAdd(&dL_dcolors[id * 3 + 1], g_c1);
                unsafeAtomicAdd(&dL_dcolors[id * 3 + 2], g_c2);
                unsafeAtomicAdd(&dL_dmean2D[id * 3 + 0], g_mx);
                unsafeAtomicAdd(&dL_dmean2D[id * 3 + 1], g_my);
                unsafeAtomicAdd(&dL_dmean2D[id * 3 + 2], g_mz);
                unsafeAtomicAdd(&dL_dopacity[id], g_op);
                float* gv = dL_dv2g + id * 10;
                unsafeAtomicAdd(gv + 0, g_v0); unsafeAtomicAdd(gv + 1, g_v1); unsafeAtomicAdd(gv + 2, g_v2);
                unsafeAtomicAdd(gv + 3, g_v3); unsafeAtomicAdd(gv + 4, g_v4); unsafeAtomicAdd(gv + 5, g_v5);
                unsafeAtomicAdd(gv + 6, g_v6); unsafeAtomicAdd(gv + 7, g_v7); unsafeAtomicAdd(gv + 8, g_v8);
                unsafeAtomicAdd(gv + 9, g_v9);
            }
        }
    }
}

} // namespace gof
