// train_epilogue.hip -- the per-iteration work of the reference's train.py around the rasterizer
// (SURVEY.md 8(f) item 2): SSIM forward/backward, depth -> points -> normals forward/backward, multi-tensor Adam.
// C ABI in include/gof_train_hip.h.  All three are streaming, HBM-bound kernels (no MFMA: 11-tap separable
// stencils and elementwise updates); the reference runs them as 30-60 torch launches per iteration.
//
//   utils/loss_utils.py:43-63   _ssim          -> ssim_fwd_kernel / ssim_bwd_kernel (16x16 tile + 5-pixel halo in LDS,
//                                                 separable 11-tap passes; backward = the same convolution of three
//                                                 partial-derivative maps, the window is symmetric and zero-padded)
//   utils/depth_utils.py:6-35   depth_to_normal-> depth_normal_fwd_kernel / depth_normal_bwd_kernel (gather form: each
//                                                 pixel recomputes the 4 neighbouring normals it contributes to)
//   torch/optim/adam.py         _multi_tensor_adam (as configured at scene/gaussian_model.py:360) -> adam_kernel
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include "../../include/gof_hip.h"
#include "../../include/gof_train_hip.h"
#include "gof_common.h"

namespace gof {

constexpr int SSIM_R = GOF_SSIM_WINDOW / 2;     // 5
constexpr int SSIM_T = 16;                      // tile edge
constexpr int SSIM_E = SSIM_T + 2 * SSIM_R;     // 26: tile + halo
struct SsimWindow { float w[GOF_SSIM_WINDOW]; };

__device__ __forceinline__ float block_sum_256(float v, float* s_red)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);   // fixed order: deterministic
}

// loss_utils.py:43-58.  grid (ceil(W/16), ceil(H/16), planes), block 256.
// WITH_L1 (gof_train_loss): the tile also sums |img1 - img2| (loss_utils.py:17-18, train.py:156) from the pixels it holds anyway.
template <bool WITH_L1>
__global__ void __launch_bounds__(256)
ssim_fwd_kernel(int W, int H, const float* __restrict__ img1, const float* __restrict__ img2, SsimWindow win,
                float* __restrict__ partial, float* __restrict__ dmaps, int planes, float* __restrict__ partial_l1)
{
    __shared__ float s_x[SSIM_E][SSIM_E + 1];
    __shared__ float s_y[SSIM_E][SSIM_E + 1];
    __shared__ float s_h[5][SSIM_E][SSIM_T];
    __shared__ float s_red[4];
    const int plane = blockIdx.z;
    const size_t base = (size_t)plane * H * W;
    const int x0 = blockIdx.x * SSIM_T - SSIM_R, y0 = blockIdx.y * SSIM_T - SSIM_R;
    for (int i = threadIdx.x; i < SSIM_E * SSIM_E; i += 256) {
        const int r = i / SSIM_E, c = i - r * SSIM_E;
        const int gx = x0 + c, gy = y0 + r;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;          // zero padding (loss_utils.py:44)
        s_x[r][c] = in ? img1[base + (size_t)gy * W + gx] : 0.0f;
        s_y[r][c] = in ? img2[base + (size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SSIM_E * SSIM_T; i += 256) {           // horizontal 11-tap pass of the 5 moments
        const int r = i / SSIM_T, c = i - r * SSIM_T;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < GOF_SSIM_WINDOW; k++) {
            const float x = s_x[r][c + k], y = s_y[r][c + k], w = win.w[k];
            a += w * x; b += w * y; aa += w * (x * x); bb += w * (y * y); ab += w * (x * y);
        }
        s_h[0][r][c] = a; s_h[1][r][c] = b; s_h[2][r][c] = aa; s_h[3][r][c] = bb; s_h[4][r][c] = ab;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float mu1 = 0.f, mu2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
    for (int k = 0; k < GOF_SSIM_WINDOW; k++) {
        const float w = win.w[k];
        mu1 += w * s_h[0][ty + k][tx]; mu2 += w * s_h[1][ty + k][tx];
        s11 += w * s_h[2][ty + k][tx]; s22 += w * s_h[3][ty + k][tx]; s12 += w * s_h[4][ty + k][tx];
    }
    const int gx = blockIdx.x * SSIM_T + tx, gy = blockIdx.y * SSIM_T + ty;
    const bool inside = gx < W && gy < H;
    const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);   // loss_utils.py:54-55
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
    const float sigma1_sq = s11 - mu1_sq, sigma2_sq = s22 - mu2_sq, sigma12 = s12 - mu1_mu2;
    const float A1 = 2.0f * mu1_mu2 + C1, A2 = 2.0f * sigma12 + C2;
    const float B1 = mu1_sq + mu2_sq + C1, B2 = sigma1_sq + sigma2_sq + C2;
    const float den = B1 * B2;
    const float m = (A1 * A2) / den;                                     // loss_utils.py:57
    if (dmaps && inside) {
        // (mu1, E[x^2], E[xy]) as independent variables: sigma1_sq = E[x^2] - mu1^2, sigma12 = E[xy] - mu1 mu2
        const size_t pix = (size_t)gy * W + gx, plane_sz = (size_t)H * W;
        const float inv_den = 1.0f / den;
        const float dm_dmu1 = 2.0f * mu2 * (A2 - A1) * inv_den - 2.0f * mu1 * m * (1.0f / B1 - 1.0f / B2);
        const float dm_ds11 = -m / B2;
        const float dm_ds12 = 2.0f * A1 * inv_den;
        dmaps[((size_t)0 * planes + plane) * plane_sz + pix] = dm_dmu1;
        dmaps[((size_t)1 * planes + plane) * plane_sz + pix] = dm_ds11;
        dmaps[((size_t)2 * planes + plane) * plane_sz + pix] = dm_ds12;
    }
    const float tot = block_sum_256(inside ? m : 0.0f, s_red);
    const size_t tile = ((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) partial[tile] = tot;
    if (WITH_L1) {
        __shared__ float s_red1[4];
        const float l1 = block_sum_256(inside ? fabsf(s_x[ty + SSIM_R][tx + SSIM_R] - s_y[ty + SSIM_R][tx + SSIM_R]) : 0.0f, s_red1);
        if (threadIdx.x == 0) partial_l1[tile] = l1;
    }
}

// one block per plane: fixed-order sum of the tile partials (deterministic .mean(), loss_utils.py:60-63)
__global__ void __launch_bounds__(256)
ssim_reduce_kernel(const float* __restrict__ partial, int tiles, float* __restrict__ plane_sums)
{
    __shared__ float s_red[4];
    const float* p = partial + (size_t)blockIdx.x * tiles;
    float v = 0.f;
    for (int i = threadIdx.x; i < tiles; i += 256) v += p[i];
    const float tot = block_sum_256(v, s_red);
    if (threadIdx.x == 0) plane_sums[blockIdx.x] = tot;
}

// WITH_L1 (gof_train_loss): every plane is scaled by `ssim_scale` and the L1 term's l1_scale * sign(img1 - img2) is added.
template <bool WITH_L1>
__global__ void __launch_bounds__(256)
ssim_bwd_kernel(int W, int H, const float* __restrict__ img1, const float* __restrict__ img2, SsimWindow win,
                const float* __restrict__ dmaps, const float* __restrict__ plane_scale, float* __restrict__ dL_dimg1, int planes,
                float ssim_scale, float l1_scale)
{
    __shared__ float s_m[3][SSIM_E][SSIM_E + 1];
    __shared__ float s_h[3][SSIM_E][SSIM_T];
    const int plane = blockIdx.z;
    const size_t plane_sz = (size_t)H * W, base = (size_t)plane * plane_sz;
    const int x0 = blockIdx.x * SSIM_T - SSIM_R, y0 = blockIdx.y * SSIM_T - SSIM_R;
    for (int i = threadIdx.x; i < SSIM_E * SSIM_E; i += 256) {
        const int r = i / SSIM_E, c = i - r * SSIM_E;
        const int gx = x0 + c, gy = y0 + r;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const size_t pix = (size_t)gy * W + gx;
#pragma unroll
        for (int q = 0; q < 3; q++) s_m[q][r][c] = in ? dmaps[((size_t)q * planes + plane) * plane_sz + pix] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SSIM_E * SSIM_T; i += 256) {
        const int r = i / SSIM_T, c = i - r * SSIM_T;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < GOF_SSIM_WINDOW; k++) {
            const float w = win.w[k];
            a += w * s_m[0][r][c + k]; b += w * s_m[1][r][c + k]; d += w * s_m[2][r][c + k];
        }
        s_h[0][r][c] = a; s_h[1][r][c] = b; s_h[2][r][c] = d;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int gx = blockIdx.x * SSIM_T + tx, gy = blockIdx.y * SSIM_T + ty;
    if (gx >= W || gy >= H) return;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < GOF_SSIM_WINDOW; k++) {
        const float w = win.w[k];
        a += w * s_h[0][ty + k][tx]; b += w * s_h[1][ty + k][tx]; d += w * s_h[2][ty + k][tx];
    }
    const size_t pix = base + (size_t)gy * W + gx;
    const float x = img1[pix], y = img2[pix];
    if (WITH_L1) {
        const float df = x - y;
        const float sg = df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : df);      // torch.sign: 0 at 0, NaN stays NaN
        dL_dimg1[pix] = ssim_scale * (a + 2.0f * x * b + y * d) + l1_scale * sg;
    } else {
        dL_dimg1[pix] = plane_scale[plane] * (a + 2.0f * x * b + y * d);
    }
}


// ---- l1_loss (utils/loss_utils.py:17-18): mean |a - b| and its gradient ---------------------------------------------------
constexpr int L1_BLOCK_ELEMS = 8192;       // 256 threads x 8 float4
__global__ void __launch_bounds__(256)
l1_fwd_kernel(uint64_t n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partial)
{
    __shared__ float s_red[4];
    const uint64_t start = (uint64_t)blockIdx.x * L1_BLOCK_ELEMS;
    float acc = 0.0f;
    const bool vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0 && start + L1_BLOCK_ELEMS <= n;
    if (vec) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint64_t i = start + (uint64_t)(it * 256 + threadIdx.x) * 4;
            const float4 x = *reinterpret_cast<const float4*>(a + i), y = *reinterpret_cast<const float4*>(b + i);
            acc += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
        }
    } else {
        for (uint64_t i = start + threadIdx.x; i < start + L1_BLOCK_ELEMS && i < n; i += 256) acc += fabsf(a[i] - b[i]);
    }
    const float tot = block_sum_256(acc, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// one block: fixed-order sum of the partials, times 1/n
__global__ void __launch_bounds__(256)
l1_final_kernel(const float* __restrict__ partial, int nb, float inv_n, float* __restrict__ out)
{
    __shared__ float s_red[4];
    float v = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) v += partial[i];
    const float tot = block_sum_256(v, s_red);
    if (threadIdx.x == 0) out[0] = tot * inv_n;
}

// d mean|a - b| / da = sign(a - b) / n, scaled by the incoming gradient (a device scalar)
__global__ void __launch_bounds__(256)
l1_bwd_kernel(uint64_t n, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g_out, float inv_n,
              float* __restrict__ dL_da)
{
    const float g = g_out[0] * inv_n;
    const uint64_t start = (uint64_t)blockIdx.x * L1_BLOCK_ELEMS;
    for (uint64_t i = start + threadIdx.x; i < start + L1_BLOCK_ELEMS && i < n; i += 256) {
        const float d = a[i] - b[i];
        dL_da[i] = g * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : d));          // torch.sign: 0 at 0
    }
}

// ---- depth -> points -> normals ----------------------------------------------------------------------
struct DepthCam { float r[9]; float o[3]; };   // c2w rotation (row-major) and origin

// c2w = inverse(world_view_transform^T) (depth_utils.py:7).  The matrix is an arbitrary 4x4 in the reference
// (torch .inverse()); evaluated here with the adjugate of the upper-left 3x3 and the affine translation,
// which is what that inverse reduces to for the [R|t; 0 0 0 1] matrices cameras.py builds.
__device__ __forceinline__ DepthCam load_depth_cam(const float* __restrict__ wvt)
{
    // wvt[i*4+j] = W2C[j][i]; W2C = [R t]
    const float a00 = wvt[0], a01 = wvt[4], a02 = wvt[8],  t0 = wvt[12];
    const float a10 = wvt[1], a11 = wvt[5], a12 = wvt[9],  t1 = wvt[13];
    const float a20 = wvt[2], a21 = wvt[6], a22 = wvt[10], t2 = wvt[14];
    const float c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const float det = a00 * c00 + a01 * c01 + a02 * c02;
    const float id = 1.0f / det;
    DepthCam c;
    c.r[0] = c00 * id;                       c.r[1] = (a02 * a21 - a01 * a22) * id;   c.r[2] = (a01 * a12 - a02 * a11) * id;
    c.r[3] = c01 * id;                       c.r[4] = (a00 * a22 - a02 * a20) * id;   c.r[5] = (a02 * a10 - a00 * a12) * id;
    c.r[6] = c02 * id;                       c.r[7] = (a01 * a20 - a00 * a21) * id;   c.r[8] = (a00 * a11 - a01 * a10) * id;
    c.o[0] = -(c.r[0] * t0 + c.r[1] * t1 + c.r[2] * t2);
    c.o[1] = -(c.r[3] * t0 + c.r[4] * t1 + c.r[5] * t2);
    c.o[2] = -(c.r[6] * t0 + c.r[7] * t1 + c.r[8] * t2);
    return c;
}

// V3 comes from gof_common.h
__device__ __forceinline__ V3 v3sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 v3add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 v3cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float v3dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// depth_utils.py:16-18: rays_d = (x+0.5, y+0.5, 1) @ inv(K)^T @ c2w[:3,:3]^T with K = [[fx,0,W/2],[0,fy,H/2],[0,0,1]]
__device__ __forceinline__ V3 ray_dir(const DepthCam& c, int x, int y, float inv_fx, float inv_fy, float ncx, float ncy)
{
    const float vx = ((float)x + 0.5f) * inv_fx + ncx;
    const float vy = ((float)y + 0.5f) * inv_fy + ncy;
    return {c.r[0] * vx + c.r[1] * vy + c.r[2], c.r[3] * vx + c.r[4] * vy + c.r[5], c.r[6] * vx + c.r[7] * vy + c.r[8]};
}
__device__ __forceinline__ V3 point_at(const DepthCam& c, const float* __restrict__ depth, int W, int x, int y,
                                       float inv_fx, float inv_fy, float ncx, float ncy)
{
    const V3 d = ray_dir(c, x, y, inv_fx, inv_fy, ncx, ncy);
    const float z = depth[(size_t)y * W + x];
    return {z * d.x + c.o[0], z * d.y + c.o[1], z * d.z + c.o[2]};      // depth_utils.py:20
}

__global__ void __launch_bounds__(256)
depth_normal_fwd_kernel(int W, int H, const float* __restrict__ depth, const float* __restrict__ wvt, float inv_fx, float inv_fy,
                        float ncx, float ncy, float* __restrict__ normals, float* __restrict__ points)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const DepthCam c = load_depth_cam(wvt);
    const size_t pix = (size_t)y * W + x;
    if (points) {
        const V3 p = point_at(c, depth, W, x, y, inv_fx, inv_fy, ncx, ncy);
        points[pix * 3 + 0] = p.x; points[pix * 3 + 1] = p.y; points[pix * 3 + 2] = p.z;
    }
    V3 n = {0.f, 0.f, 0.f};
    if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1) {                   // depth_utils.py:31-34
        const V3 dx = v3sub(point_at(c, depth, W, x, y + 1, inv_fx, inv_fy, ncx, ncy), point_at(c, depth, W, x, y - 1, inv_fx, inv_fy, ncx, ncy));
        const V3 dy = v3sub(point_at(c, depth, W, x + 1, y, inv_fx, inv_fy, ncx, ncy), point_at(c, depth, W, x - 1, y, inv_fx, inv_fy, ncx, ncy));
        const V3 cr = v3cross(dx, dy);
        const float len = sqrtf(v3dot(cr, cr));
        const float dn = fmaxf(len, 1e-12f);                            // F.normalize eps
        n = {cr.x / dn, cr.y / dn, cr.z / dn};
    }
    normals[pix * 3 + 0] = n.x; normals[pix * 3 + 1] = n.y; normals[pix * 3 + 2] = n.z;
}

// For interior pixel q: returns dL/d(dx[q]) and dL/d(dy[q]) given dL/d(normal[q]).
__device__ __forceinline__ void normal_pixel_grads(const DepthCam& c, const float* __restrict__ depth, const float* __restrict__ gN,
                                                   int W, int qx, int qy, float inv_fx, float inv_fy, float ncx, float ncy,
                                                   V3& g_dx, V3& g_dy)
{
    const V3 dx = v3sub(point_at(c, depth, W, qx, qy + 1, inv_fx, inv_fy, ncx, ncy), point_at(c, depth, W, qx, qy - 1, inv_fx, inv_fy, ncx, ncy));
    const V3 dy = v3sub(point_at(c, depth, W, qx + 1, qy, inv_fx, inv_fy, ncx, ncy), point_at(c, depth, W, qx - 1, qy, inv_fx, inv_fy, ncx, ncy));
    const V3 cr = v3cross(dx, dy);
    const size_t q = ((size_t)qy * W + qx) * 3;
    const V3 g = {gN[q], gN[q + 1], gN[q + 2]};
    const float len = sqrtf(v3dot(cr, cr));
    V3 G;
    if (len >= 1e-12f) {                                                // v / max(|v|, eps): gradient through the norm
        const float il = 1.0f / len;
        const V3 n = {cr.x * il, cr.y * il, cr.z * il};
        const float ng = v3dot(n, g);
        G = {(g.x - n.x * ng) * il, (g.y - n.y * ng) * il, (g.z - n.z * ng) * il};
    } else {                                                            // clamped denominator: constant 1/eps
        G = {g.x / 1e-12f, g.y / 1e-12f, g.z / 1e-12f};
    }
    g_dx = v3cross(dy, G);                                              // d(dx x dy).G / d dx
    g_dy = v3cross(G, dx);
}

__global__ void __launch_bounds__(256)
depth_normal_bwd_kernel(int W, int H, const float* __restrict__ depth, const float* __restrict__ wvt, float inv_fx, float inv_fy,
                        float ncx, float ncy, const float* __restrict__ gN, const float* __restrict__ gP, float* __restrict__ dL_ddepth)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const DepthCam c = load_depth_cam(wvt);
    const size_t pix = (size_t)y * W + x;
    V3 g = {0.f, 0.f, 0.f};
    if (gP) g = {gP[pix * 3], gP[pix * 3 + 1], gP[pix * 3 + 2]};
    V3 a, b;
    // dx[q] = P[q + row] - P[q - row]; dy[q] = P[q + col] - P[q - col]; q interior
    const bool col_in = x >= 1 && x < W - 1, row_in = y >= 1 && y < H - 1;
    if (col_in && y - 1 >= 1)     { normal_pixel_grads(c, depth, gN, W, x, y - 1, inv_fx, inv_fy, ncx, ncy, a, b); g = v3add(g, a); }
    if (col_in && y + 1 < H - 1)  { normal_pixel_grads(c, depth, gN, W, x, y + 1, inv_fx, inv_fy, ncx, ncy, a, b); g = v3sub(g, a); }
    if (row_in && x - 1 >= 1)     { normal_pixel_grads(c, depth, gN, W, x - 1, y, inv_fx, inv_fy, ncx, ncy, a, b); g = v3add(g, b); }
    if (row_in && x + 1 < W - 1)  { normal_pixel_grads(c, depth, gN, W, x + 1, y, inv_fx, inv_fy, ncx, ncy, a, b); g = v3sub(g, b); }
    dL_ddepth[pix] = v3dot(ray_dir(c, x, y, inv_fx, inv_fy, ncx, ncy), g);
}


// ---- the loss of train.py:150-188 in one pass over the rendering (gof_train_loss) ---------------------------------
// Geometry terms: distortion_loss = mean(rendering[8]) (train.py:164-167); depth-normal consistency (train.py:170-182):
// depth_normal = depth_to_normal(view, rendering[6]), render_normal = F.normalize(rendering[3:6], dim=0) rotated to world
// space by c2w[:3,:3], error = 1 - <render_normal_world, depth_normal>.  One thread per pixel: the two tile sums, the
// gradient w.r.t. channels 3-5 (through the rotation and the normalisation), the constant gradient of channels 7-8, and
// dL/d(depth_normal) for the gather-form depth backward that follows.  g_dn = lambda_depth_normal / (H W), g_dist likewise.
__global__ void __launch_bounds__(256)
loss_geom_kernel(int W, int H, const float* __restrict__ rendering, const float* __restrict__ wvt, float inv_fx, float inv_fy,
                 float ncx, float ncy, float g_dn, float g_dist, float* __restrict__ partial_dn, float* __restrict__ partial_dist,
                 float* __restrict__ gN, float* __restrict__ dL)
{
    __shared__ float s_red[4];
    __shared__ float s_red1[4];
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    const bool inside = x < W && y < H;
    const size_t plane = (size_t)H * W;
    float err = 0.0f, dist = 0.0f;
    if (inside) {
        const DepthCam c = load_depth_cam(wvt);
        const float* depth = rendering + 6 * plane;
        const size_t pix = (size_t)y * W + x;
        V3 n = {0.f, 0.f, 0.f};
        if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1) {                   // depth_utils.py:31-34
            const V3 dx = v3sub(point_at(c, depth, W, x, y + 1, inv_fx, inv_fy, ncx, ncy), point_at(c, depth, W, x, y - 1, inv_fx, inv_fy, ncx, ncy));
            const V3 dy = v3sub(point_at(c, depth, W, x + 1, y, inv_fx, inv_fy, ncx, ncy), point_at(c, depth, W, x - 1, y, inv_fx, inv_fy, ncx, ncy));
            const V3 cr = v3cross(dx, dy);
            const float dn = fmaxf(sqrtf(v3dot(cr, cr)), 1e-12f);
            n = {cr.x / dn, cr.y / dn, cr.z / dn};
        }
        const V3 r = {rendering[3 * plane + pix], rendering[4 * plane + pix], rendering[5 * plane + pix]};
        const float len = sqrtf(v3dot(r, r));
        const float den = fmaxf(len, 1e-12f);                               // F.normalize(p=2, dim=0), train.py:175
        const V3 rn = {r.x / den, r.y / den, r.z / den};
        const V3 wn = {c.r[0] * rn.x + c.r[1] * rn.y + c.r[2] * rn.z,      // c2w[:3,:3] @ render_normal, train.py:177-179
                       c.r[3] * rn.x + c.r[4] * rn.y + c.r[5] * rn.z,
                       c.r[6] * rn.x + c.r[7] * rn.y + c.r[8] * rn.z};
        err = 1.0f - v3dot(wn, n);                                          // train.py:181
        dist = rendering[8 * plane + pix];
        if (dL) {
            gN[pix * 3 + 0] = -g_dn * wn.x; gN[pix * 3 + 1] = -g_dn * wn.y; gN[pix * 3 + 2] = -g_dn * wn.z;
            const V3 gw = {-g_dn * n.x, -g_dn * n.y, -g_dn * n.z};
            const V3 g = {c.r[0] * gw.x + c.r[3] * gw.y + c.r[6] * gw.z,   // rotation transposed
                          c.r[1] * gw.x + c.r[4] * gw.y + c.r[7] * gw.z,
                          c.r[2] * gw.x + c.r[5] * gw.y + c.r[8] * gw.z};
            V3 gr;
            if (len >= 1e-12f) {
                const float il = 1.0f / len, ng = v3dot(rn, g);
                gr = {(g.x - rn.x * ng) * il, (g.y - rn.y * ng) * il, (g.z - rn.z * ng) * il};
            } else {
                gr = {g.x / 1e-12f, g.y / 1e-12f, g.z / 1e-12f};            // clamped denominator: constant 1/eps
            }
            dL[3 * plane + pix] = gr.x; dL[4 * plane + pix] = gr.y; dL[5 * plane + pix] = gr.z;
            dL[7 * plane + pix] = 0.0f;                                     // the alpha channel is not part of the loss
            dL[8 * plane + pix] = g_dist;
        }
    }
    const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const float e = block_sum_256(err, s_red);
    const float d = block_sum_256(dist, s_red1);
    if (threadIdx.x == 0) { partial_dn[tile] = e; partial_dist[tile] = d; }
}

// one block of 1024 threads: fixed-order sums of the four partial arrays, then the scalar composition of train.py:161, 188 in fp32
// terms = {loss, Ll1, ssim, rgb_loss, depth_normal_loss, distortion_loss}
__global__ void __launch_bounds__(1024)
loss_final_kernel(const float* __restrict__ p_ssim, const float* __restrict__ p_l1, const float* __restrict__ p_dn,
                  const float* __restrict__ p_dist, int tiles, float n_rgb, float n_pix, float w_l1, float lambda_dssim,
                  float lambda_dn, float lambda_dist, float* __restrict__ terms)
{
    __shared__ float s_red[4][16];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < 3 * tiles; i += 1024) { v[0] += p_ssim[i]; v[1] += p_l1[i]; }
    for (int i = threadIdx.x; i < tiles; i += 1024) { v[2] += p_dn[i]; v[3] += p_dist[i]; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) s_red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float x = 0.f;
            for (int w = 0; w < 16; w++) x += s_red[k][w];       // fixed order: deterministic
            tot[k] = x;
        }
        const float Ll1 = tot[1] / n_rgb, ssim = tot[0] / n_rgb, dn = tot[2] / n_pix, dist = tot[3] / n_pix;
        const float rgb = w_l1 * Ll1 + lambda_dssim * (1.0f - ssim);
        terms[0] = rgb + dn * lambda_dn + dist * lambda_dist;
        terms[1] = Ll1; terms[2] = ssim; terms[3] = rgb; terms[4] = dn; terms[5] = dist;
    }
}

// ---- Adam --------------------------------------------------------------------------------------------
constexpr int ADAM_BLOCK_ELEMS = 4096;     // 256 threads x 4 float4
#ifndef GOF_ADAM_NT
#define GOF_ADAM_NT 1                      // nontemporal loads / stores in the vector body (adam_kernel below)
#endif
struct AdamArgs {
    GofAdamTensor t[GOF_ADAM_MAX_TENSORS];
    uint32_t block_end[GOF_ADAM_MAX_TENSORS];   // exclusive prefix of blocks per tensor
    int32_t n;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float w2,
                                         float step_size, float bc2_sqrt, float eps)
{
    m = m + w1 * (g - m);                          // _foreach_lerp_(exp_avgs, grads, 1 - beta1), weight < 0.5
    v = v * beta2;                                 // _foreach_mul_(exp_avg_sqs, beta2)
    v = v + (w2 * g) * g;                          // _foreach_addcmul_(exp_avg_sqs, grads, grads, 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps; // sqrt, div_(bias_correction2_sqrt), add_(eps)
    p = p + (step_size * m) / denom;               // _foreach_addcdiv_(params, exp_avgs, denom, step_size)
}

__global__ void __launch_bounds__(256)
adam_kernel(AdamArgs a, float w1, float beta2, float w2, float eps)
{
    int ti = 0;
    while (ti < a.n - 1 && blockIdx.x >= a.block_end[ti]) ti++;
    const GofAdamTensor t = a.t[ti];
    const uint32_t b0 = ti ? a.block_end[ti - 1] : 0u;
    const uint64_t start = (uint64_t)(blockIdx.x - b0) * ADAM_BLOCK_ELEMS;
    const bool vec = ((((uintptr_t)t.param) | ((uintptr_t)t.grad) | ((uintptr_t)t.exp_avg) | ((uintptr_t)t.exp_avg_sq)) & 15) == 0;
    if (vec && start + ADAM_BLOCK_ELEMS <= t.n) {
        // all sixteen 16-byte loads of the thread first, then the arithmetic and the twelve stores: the four tensors come through
        // plain pointers of a struct (they may alias as far as the compiler knows), so a store of one iteration kept the loads of the
        // next behind it -- 64 bytes per lane in flight instead of 256 (round 5: 5.0 -> see profiles/r05_*; same bits).
        // Round 6: NONTEMPORAL loads and stores -- every byte is touched once per step and the four tensors together (0.94 GB at 1M
        // Gaussians, 5.7 GB at 6M) are far beyond L2 + MALL: 9.91 GB in 1.817 -> 1.694 ms (5.46 -> 5.85 TB/s) at 6M x 59 floats,
        // 1.65 GB in 0.2715 -> 0.2523 ms (6.09 -> 6.55 TB/s) at 1M x 59, interleaved with eight other forms of the loop (stores only:
        // nothing; the gradient load only: half of it; 2 or 8 float4 per tensor and thread, persistent workgroups: nothing or worse --
        // tests/devtools/microbench/adam_stream.hip, profiles/r06_adam_stream.txt).  GOF_ADAM_NT=2 keeps the parameter store cached.
        typedef float v4f __attribute__((ext_vector_type(4)));
#if GOF_ADAM_NT
#define GOF_ADAM_LD(PTR) __builtin_nontemporal_load(reinterpret_cast<const v4f*>(PTR))
#define GOF_ADAM_ST(PTR, X) __builtin_nontemporal_store(X, reinterpret_cast<v4f*>(PTR))
#else
#define GOF_ADAM_LD(PTR) (*reinterpret_cast<const v4f*>(PTR))
#define GOF_ADAM_ST(PTR, X) (*reinterpret_cast<v4f*>(PTR) = (X))
#endif
        v4f p[4], g[4], m[4], v[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const uint64_t i = start + (uint64_t)(it * 256 + threadIdx.x) * 4;
            p[it] = GOF_ADAM_LD(t.param + i);
            g[it] = GOF_ADAM_LD(t.grad + i);
            m[it] = GOF_ADAM_LD(t.exp_avg + i);
            v[it] = GOF_ADAM_LD(t.exp_avg_sq + i);
        }
#pragma unroll
        for (int it = 0; it < 4; it++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float pp = p[it][c], mm = m[it][c], vv = v[it][c];
                adam_one(pp, g[it][c], mm, vv, w1, beta2, w2, t.step_size, t.bias_correction2_sqrt, eps);
                p[it][c] = pp; m[it][c] = mm; v[it][c] = vv;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const uint64_t i = start + (uint64_t)(it * 256 + threadIdx.x) * 4;
#if GOF_ADAM_NT == 2
            *reinterpret_cast<v4f*>(t.param + i) = p[it];
#else
            GOF_ADAM_ST(t.param + i, p[it]);
#endif
            GOF_ADAM_ST(t.exp_avg + i, m[it]);
            GOF_ADAM_ST(t.exp_avg_sq + i, v[it]);
        }
#undef GOF_ADAM_LD
#undef GOF_ADAM_ST
    } else {
        for (uint64_t i = start + threadIdx.x; i < start + ADAM_BLOCK_ELEMS && i < t.n; i += 256) {
            float p = t.param[i], m = t.exp_avg[i], v = t.exp_avg_sq[i];
            adam_one(p, t.grad[i], m, v, w1, beta2, w2, t.step_size, t.bias_correction2_sqrt, eps);
            t.param[i] = p; t.exp_avg[i] = m; t.exp_avg_sq[i] = v;
        }
    }
}

} // namespace gof

using namespace gof;

extern "C" {

size_t gof_ssim_scratch_bytes(int32_t planes, int32_t W, int32_t H)
{
    if (planes <= 0 || W <= 0 || H <= 0) return 256;
    const size_t tiles = (size_t)((W + SSIM_T - 1) / SSIM_T) * ((H + SSIM_T - 1) / SSIM_T);
    return ((size_t)planes * tiles * sizeof(float) + 255) & ~(size_t)255;
}

static int ssim_check(int32_t planes, int32_t W, int32_t H, const void* a, const void* b, const void* w)
{
    if (planes <= 0 || W <= 0 || H <= 0) { set_error("bad planes/W/H (%d, %d, %d)", planes, W, H); return GOF_E_INVALID; }
    if (planes > 65535) { set_error("too many planes (%d)", planes); return GOF_E_INVALID; }
    if (!a || !b || !w) { set_error("a required pointer is NULL"); return GOF_E_INVALID; }
    return GOF_OK;
}

int gof_ssim_forward(int32_t planes, int32_t W, int32_t H, const float* img1, const float* img2, const float* window_host,
                     float* plane_sums, float* dmaps, void* scratch, size_t scratch_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int e = ssim_check(planes, W, H, img1, img2, window_host)) return e;
    if (!plane_sums || !scratch) { set_error("plane_sums / scratch is NULL"); return GOF_E_INVALID; }
    if (scratch_bytes < gof_ssim_scratch_bytes(planes, W, H)) { set_error("ssim scratch too small"); return GOF_E_WORKSPACE; }
    SsimWindow win;
    memcpy(win.w, window_host, sizeof(win.w));
    const dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, planes);
    float* partial = static_cast<float*>(scratch);
    { GOF_PROFILE("ssim_forward", stream);
      hipLaunchKernelGGL(ssim_fwd_kernel<false>, grid, dim3(256), 0, stream, W, H, img1, img2, win, partial, dmaps, planes, (float*)nullptr);
      GOF_LAUNCH_CHECK(stream, 0);
      hipLaunchKernelGGL(ssim_reduce_kernel, dim3(planes), dim3(256), 0, stream, partial, (int)(grid.x * grid.y), plane_sums);
      GOF_LAUNCH_CHECK(stream, 0); }
    return GOF_OK;
}

int gof_ssim_backward(int32_t planes, int32_t W, int32_t H, const float* img1, const float* img2, const float* window_host,
                      const float* dmaps, const float* plane_scale, float* dL_dimg1, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int e = ssim_check(planes, W, H, img1, img2, window_host)) return e;
    if (!dmaps || !plane_scale || !dL_dimg1) { set_error("dmaps / plane_scale / dL_dimg1 is NULL"); return GOF_E_INVALID; }
    SsimWindow win;
    memcpy(win.w, window_host, sizeof(win.w));
    const dim3 grid((W + SSIM_T - 1) / SSIM_T, (H + SSIM_T - 1) / SSIM_T, planes);
    { GOF_PROFILE("ssim_backward", stream);
      hipLaunchKernelGGL(ssim_bwd_kernel<false>, grid, dim3(256), 0, stream, W, H, img1, img2, win, dmaps, plane_scale, dL_dimg1, planes, 0.0f, 0.0f);
      GOF_LAUNCH_CHECK(stream, 0); }
    return GOF_OK;
}

size_t gof_l1_scratch_bytes(uint64_t n) { return (((n + L1_BLOCK_ELEMS - 1) / L1_BLOCK_ELEMS) * sizeof(float) + 255) & ~(size_t)255; }

int gof_l1_forward(uint64_t n, const float* a, const float* b, float* out_mean, void* scratch, size_t scratch_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n == 0 || !a || !b || !out_mean || !scratch) { set_error("l1: empty input or NULL pointer"); return GOF_E_INVALID; }
    if (scratch_bytes < gof_l1_scratch_bytes(n)) { set_error("l1 scratch too small"); return GOF_E_WORKSPACE; }
    const uint64_t nb = (n + L1_BLOCK_ELEMS - 1) / L1_BLOCK_ELEMS;
    if (nb > 0x7fffffffull) { set_error("l1: too many elements"); return GOF_E_INVALID; }
    float* partial = static_cast<float*>(scratch);
    { GOF_PROFILE("l1_forward", stream);
      hipLaunchKernelGGL(l1_fwd_kernel, dim3((uint32_t)nb), dim3(256), 0, stream, n, a, b, partial);
      GOF_LAUNCH_CHECK(stream, 0);
      hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, stream, partial, (int)nb, (float)(1.0 / (double)n), out_mean);
      GOF_LAUNCH_CHECK(stream, 0); }
    return GOF_OK;
}

int gof_l1_backward(uint64_t n, const float* a, const float* b, const float* grad_out, float* dL_da, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n == 0 || !a || !b || !grad_out || !dL_da) { set_error("l1: empty input or NULL pointer"); return GOF_E_INVALID; }
    const uint64_t nb = (n + L1_BLOCK_ELEMS - 1) / L1_BLOCK_ELEMS;
    { GOF_PROFILE("l1_backward", stream);
      hipLaunchKernelGGL(l1_bwd_kernel, dim3((uint32_t)nb), dim3(256), 0, stream, n, a, b, grad_out, (float)(1.0 / (double)n), dL_da);
      GOF_LAUNCH_CHECK(stream, 0); }
    return GOF_OK;
}

static int depth_check(int32_t W, int32_t H, const void* depth, const void* wvt, float fx, float fy)
{
    if (W <= 0 || H <= 0) { set_error("bad W/H (%d, %d)", W, H); return GOF_E_INVALID; }
    if (!depth || !wvt) { set_error("depth / world_view_transform is NULL"); return GOF_E_INVALID; }
    if (!(fx != 0.0f) || !(fy != 0.0f)) { set_error("bad focal length"); return GOF_E_INVALID; }
    return GOF_OK;
}

int gof_depth_to_normal(int32_t W, int32_t H, const float* depth, const float* world_view_transform, float fx, float fy,
                        float* normals, float* points, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int e = depth_check(W, H, depth, world_view_transform, fx, fy)) return e;
    if (!normals) { set_error("normals is NULL"); return GOF_E_INVALID; }
    const dim3 grid((W + 15) / 16, (H + 15) / 16);
    // inv(K) of depth_utils.py:11-18: [[1/fx, 0, -(W/2)/fx], [0, 1/fy, -(H/2)/fy], [0, 0, 1]]
    const float inv_fx = 1.0f / fx, inv_fy = 1.0f / fy;
    const float ncx = -((float)W / 2.0f) / fx, ncy = -((float)H / 2.0f) / fy;
    { GOF_PROFILE("depth_to_normal", stream);
      hipLaunchKernelGGL(depth_normal_fwd_kernel, grid, dim3(256), 0, stream, W, H, depth, world_view_transform, inv_fx, inv_fy, ncx, ncy,
                         normals, points);
      GOF_LAUNCH_CHECK(stream, 0); }
    return GOF_OK;
}

int gof_depth_to_normal_backward(int32_t W, int32_t H, const float* depth, const float* world_view_transform, float fx, float fy,
                                 const float* dL_dnormals, const float* dL_dpoints, float* dL_ddepth, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int e = depth_check(W, H, depth, world_view_transform, fx, fy)) return e;
    if (!dL_dnormals || !dL_ddepth) { set_error("dL_dnormals / dL_ddepth is NULL"); return GOF_E_INVALID; }
    const dim3 grid((W + 15) / 16, (H + 15) / 16);
    const float inv_fx = 1.0f / fx, inv_fy = 1.0f / fy;
    const float ncx = -((float)W / 2.0f) / fx, ncy = -((float)H / 2.0f) / fy;
    { GOF_PROFILE("depth_to_normal_backward", stream);
      hipLaunchKernelGGL(depth_normal_bwd_kernel, grid, dim3(256), 0, stream, W, H, depth, world_view_transform, inv_fx, inv_fy, ncx, ncy,
                         dL_dnormals, dL_dpoints, dL_ddepth);
      GOF_LAUNCH_CHECK(stream, 0); }
    return GOF_OK;
}

static size_t loss_tiles(int32_t W, int32_t H) { return (size_t)((W + 15) / 16) * ((H + 15) / 16); }
static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

size_t gof_train_loss_scratch_bytes(int32_t W, int32_t H)
{
    if (W <= 0 || H <= 0) return 256;
    const size_t px = (size_t)W * H;
    // partial sums (ssim 3, l1 3, depth-normal 1, distortion 1 per tile) | SSIM derivative maps 3 x 3 planes | dL/d(depth normal)
    return align256(8 * loss_tiles(W, H) * sizeof(float)) + align256(9 * px * sizeof(float)) + align256(3 * px * sizeof(float));
}

int gof_train_loss(int32_t W, int32_t H, const float* rendering, const float* gt_image, const float* window_host,
                   const float* world_view_transform, float fx, float fy, double lambda_dssim, double lambda_depth_normal,
                   double lambda_distortion, float* terms, float* dL_drendering, void* scratch, size_t scratch_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int e = ssim_check(3, W, H, rendering, gt_image, window_host)) return e;
    if (int e = depth_check(W, H, rendering, world_view_transform, fx, fy)) return e;
    if (!terms || !scratch) { set_error("terms / scratch is NULL"); return GOF_E_INVALID; }
    if (scratch_bytes < gof_train_loss_scratch_bytes(W, H)) { set_error("train-loss scratch too small"); return GOF_E_WORKSPACE; }
    SsimWindow win;
    memcpy(win.w, window_host, sizeof(win.w));
    const size_t px = (size_t)W * H, tiles = loss_tiles(W, H);
    char* base = static_cast<char*>(scratch);
    float* p_ssim = reinterpret_cast<float*>(base);
    float* p_l1 = p_ssim + 3 * tiles;
    float* p_dn = p_l1 + 3 * tiles;
    float* p_dist = p_dn + tiles;
    float* dmaps = reinterpret_cast<float*>(base + align256(8 * tiles * sizeof(float)));
    float* gN = reinterpret_cast<float*>(base + align256(8 * tiles * sizeof(float)) + align256(9 * px * sizeof(float)));
    const dim3 grid3((W + 15) / 16, (H + 15) / 16, 3), grid1((W + 15) / 16, (H + 15) / 16);
    const float inv_fx = 1.0f / fx, inv_fy = 1.0f / fy;
    const float ncx = -((float)W / 2.0f) / fx, ncy = -((float)H / 2.0f) / fy;
    const double n_rgb = 3.0 * (double)px, n_pix = (double)px;
    { GOF_PROFILE("train_loss", stream);
      hipLaunchKernelGGL(ssim_fwd_kernel<true>, grid3, dim3(256), 0, stream, W, H, rendering, gt_image, win, p_ssim,
                         dL_drendering ? dmaps : (float*)nullptr, 3, p_l1);
      GOF_LAUNCH_CHECK(stream, 0);
      hipLaunchKernelGGL(loss_geom_kernel, grid1, dim3(256), 0, stream, W, H, rendering, world_view_transform, inv_fx, inv_fy, ncx, ncy,
                         (float)(lambda_depth_normal / n_pix), (float)(lambda_distortion / n_pix), p_dn, p_dist, gN, dL_drendering);
      GOF_LAUNCH_CHECK(stream, 0);
      hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1024), 0, stream, p_ssim, p_l1, p_dn, p_dist, (int)tiles, (float)n_rgb,
                         (float)n_pix, (float)(1.0 - lambda_dssim), (float)lambda_dssim, (float)lambda_depth_normal,
                         (float)lambda_distortion, terms);
      GOF_LAUNCH_CHECK(stream, 0);
      if (dL_drendering) {
          // d loss / d image = (1 - l) sign(image - gt) / (3HW) - l d ssim / d image (train.py:161)
          hipLaunchKernelGGL(ssim_bwd_kernel<true>, grid3, dim3(256), 0, stream, W, H, rendering, gt_image, win, dmaps, (const float*)nullptr,
                             dL_drendering, 3, (float)(-lambda_dssim / n_rgb), (float)((1.0 - lambda_dssim) / n_rgb));
          GOF_LAUNCH_CHECK(stream, 0);
          hipLaunchKernelGGL(depth_normal_bwd_kernel, grid1, dim3(256), 0, stream, W, H, rendering + 6 * px, world_view_transform, inv_fx, inv_fy,
                             ncx, ncy, gN, (const float*)nullptr, dL_drendering + 6 * px);
          GOF_LAUNCH_CHECK(stream, 0);
      } }
    return GOF_OK;
}

int gof_adam_step(int32_t n_tensors, const GofAdamTensor* tensors_host, double beta1, double beta2, double eps, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_tensors < 0 || n_tensors > GOF_ADAM_MAX_TENSORS) { set_error("n_tensors %d outside [0, %d]", n_tensors, GOF_ADAM_MAX_TENSORS); return GOF_E_INVALID; }
    if (n_tensors && !tensors_host) { set_error("tensors_host is NULL"); return GOF_E_INVALID; }
    AdamArgs a;
    memset(&a, 0, sizeof(a));
    uint64_t blocks = 0;
    int k = 0;
    for (int i = 0; i < n_tensors; i++) {
        const GofAdamTensor& t = tensors_host[i];
        if (t.n == 0) continue;
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) { set_error("tensor %d: a pointer is NULL", i); return GOF_E_INVALID; }
        blocks += (t.n + ADAM_BLOCK_ELEMS - 1) / ADAM_BLOCK_ELEMS;
        if (blocks > 0x7fffffffull) { set_error("too many elements for one launch"); return GOF_E_INVALID; }
        a.t[k] = t;
        a.block_end[k] = (uint32_t)blocks;
        k++;
    }
    a.n = k;
    if (!k) return GOF_OK;
    { GOF_PROFILE("adam_step", stream);
      // torch forms 1 - beta in Python doubles and rounds the scalar to fp32 once (1 - 0.999f would be off by 1.3e-5 relative)
      hipLaunchKernelGGL(adam_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a, (float)(1.0 - beta1), (float)beta2,
                         (float)(1.0 - beta2), (float)eps);
      GOF_LAUNCH_CHECK(stream, 0); }
    return GOF_OK;
}

} // extern "C"
