// blend_forward.hip -- K7, the tiled forward alpha-blend (replaces renderCUDA, reference
// forward.cu:409-612).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; lane l of wave w owns pixel
// (x = l % 16, y = 4*w + l / 16) -- the same thread_rank -> pixel map as the reference, so
// per-pixel results do not depend on the decomposition.
//
// MI355X design:
//  * tile-list entries are staged 256 at a time into LDS as whole 64-byte SplatRec lines
//    (view2gaussian + opacity + colour + 2D mean): one aligned 64-B gather per entry, colour
//    included (the reference re-reads colour from global memory per contributing pair,
//    forward.cu:561).  LDS layout is [4][256] float4, so the staging ds_write_b128 of 64
//    consecutive lanes are contiguous (conflict-free) and the inner-loop reads are wave-uniform
//    broadcasts.
//  * per-wave early exit: a wave whose 64 pixels are all saturated skips the batch
//    (ballot over `done`), the workgroup exits when all 4 waves are done (forward.cu:475-477).
//  * conservative fp32 cull (pair_certainly_transparent): a wave whose lanes are all CERTAINLY below
//    alpha = 1/255 for this splat skips the exact fp64 division / exp path entirely; any lane that is
//    not certainly transparent takes the exact path, so the result is unchanged (bit-exact).
//  * a wave skips the heavy "contributing" path when no lane passes the alpha test.
//  * XCD-aware tile order (xcd_tile_id): neighbouring tiles, which gather the same records,
//    run on the same XCD and share its L2.
//
// Arithmetic: identical operation sequence to the oracle (fp32 products/sums in source order,
// fp64 for AA/BB/min_value, mapped depth, normal length and the final distortion normalisation,
// as forward.cu:504-557, 589), so every output is expected to be bit-identical to the oracle.
#include "gof_common.h"

namespace gof {

__global__ void __launch_bounds__(256)
blend_forward(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
              int W, int H, float focal_x, float focal_y, const float* __restrict__ bg_color,
              float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
              uint32_t gx, uint32_t ntiles)
{
    const uint32_t tile = xcd_tile_id(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    const uint32_t tx = tile % gx, ty = tile / gx;
    const uint32_t tid = threadIdx.x;
    const uint32_t px = tx * TILE_X + (tid % TILE_X), py = ty * TILE_Y + (tid / TILE_X);
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const float pixfx = (float)px + 0.5f, pixfy = (float)py + 0.5f;
    const float rx = (float)(((double)pixfx - W / 2.) / (double)focal_x);
    const float ry = (float)(((double)pixfy - H / 2.) / (double)focal_y);

    const uint2 range = ranges[tile];
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + TILE_PIX - 1) / TILE_PIX;

    __shared__ float4 s_rec[4][TILE_PIX];
    __shared__ float s_thr[TILE_PIX];

    bool done = !inside;
    float T = 1.0f;
    uint32_t contributor = 0, last_contributor = 0, max_contributor = (uint32_t)-1;
    float C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0, Dp = 0, Al = 0;
    float dist1 = 0, dist2 = 0, distortion = 0;

    for (int i = 0; i < rounds; i++, toDo -= TILE_PIX) {
        if (__syncthreads_and(done)) break;
        const uint32_t k = range.x + (uint32_t)i * TILE_PIX + tid;
        if (k < range.y) {
            const uint32_t id = point_list[k];
            const float4* src = reinterpret_cast<const float4*>(&rec[id]);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            s_rec[0][tid] = a; s_rec[1][tid] = b; s_rec[2][tid] = c; s_rec[3][tid] = d;
            s_thr[tid] = cull_log_threshold(c.z);
        }
        __syncthreads();
        if (__ballot(!done) == 0ull) continue;   // whole wave saturated: help staging only

        const int n = min(TILE_PIX, toDo);
        for (int j = 0; j < n; j++) {
            if (__ballot(!done) == 0ull) break;      // wave-uniform: every pixel of this wave is saturated
            if (done) continue;
            contributor++;
            const float4 a = s_rec[0][j], b = s_rec[1][j], c = s_rec[2][j];
            const float v[10] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y };
            PairEval p;
            pair_prelude(v, rx, ry, p);
            if (pair_certainly_transparent(p, c.y, s_thr[j])) continue;
            pair_exact(v, c.z, p);
            if (p.skip) continue;
            const float alpha = p.alpha, t = p.t;
            const float test_T = T * (1 - alpha);
            if (test_T < 0.0001f) { done = true; continue; }

            const float4 d = s_rec[3][j];
            const float max_t = t;
            const float mapped_max_t = (float)((GOF_FAR_PLANE * max_t - GOF_FAR_PLANE * GOF_NEAR_PLANE) / ((GOF_FAR_PLANE - GOF_NEAR_PLANE) * max_t));
            const float length = (float)sqrt((double)(p.n0 * p.n0 + p.n1 * p.n1 + p.n2 * p.n2) + 1e-7);
            const float nn0 = -p.n0 / length, nn1 = -p.n1 / length, nn2 = -p.n2 / length;

            const float A = 1 - T;
            const float error = mapped_max_t * mapped_max_t * A + dist2 - 2 * mapped_max_t * dist1;
            distortion += error * alpha * T;
            dist1 += mapped_max_t * alpha * T;
            dist2 += mapped_max_t * mapped_max_t * alpha * T;

            C0 += c.w * alpha * T;
            C1 += d.x * alpha * T;
            C2 += d.y * alpha * T;
            N0 += nn0 * alpha * T;
            N1 += nn1 * alpha * T;
            N2 += nn2 * alpha * T;
            if (T > 0.5f) { Dp = t; max_contributor = contributor; }
            Al += alpha * T;
            T = test_T;
            last_contributor = contributor;
        }
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

} // namespace gof
