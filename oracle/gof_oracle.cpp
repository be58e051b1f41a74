/*
 * gof_oracle.cpp -- TEST INFRASTRUCTURE.  CPU restatement of the reference rasterizer.
 *
 * This file is the parity oracle for libgof_hip.so.  It is NOT part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  It follows
 * the reference CUDA sources line by line (citations are relative to
 * /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/):
 *
 *   preprocess          forward.cu:283-404  (+ :20-71 SH, :74-124 cov2D, :129-163 cov3D, :168-279 view2gaussian)
 *   binning             rasterizer_impl.cu:70-111 (duplicateWithKeys), :149-171 (identifyTileRanges),
 *                       :332 (inclusive scan), :355-363 (stable sort on bits [0, 32+msb)), :35-50 (getHigherMsb)
 *   forward blend       forward.cu:409-612
 *   backward blend      backward.cu:634-955
 *   preprocess backward backward.cu:593-631 (+ :381-587 view2gaussian bwd, :20-139 SH bwd)
 *   integrate           forward.cu:722-766, 803-1218; rasterizer_impl.cu:113-144, 530-792
 *   helpers             auxiliary.h:59-74 (ndc2Pix, getRect), :86-115 (transforms), :177-202 (in_frustum)
 *
 * Arithmetic contract: every expression is evaluated in the type and left-to-right order
 * the C++ source of the reference prescribes (fp32 unless the reference mixes in a double
 * literal or a double variable), with NO multiply-add fusion (compile with
 * -ffp-contract=off).  The CUDA binary's own FMA contraction cannot be reproduced (no
 * NVIDIA toolchain here); DESIGN.md discusses what that means for "bit-exact".
 * exp(): the reference calls CUDA's expf (<= 2 ulp, not reproducible on a host).  The
 * oracle uses oexpf() below -- a Cephes-style fp32 exp built only from IEEE mul/fma/rint/
 * ldexp, so that the HIP kernels can evaluate the *identical* sequence and per-pair alpha
 * is bit-identical between oracle and device (this removes threshold flips at
 * alpha < 1/255 and T < 1e-4 from the parity question).
 *
 * Pinning status: the reference has no tests or golden vectors for this path (SURVEY.md
 * section 4).  The oracle is pinned (a) sub-result by sub-result against the reference's own
 * Python restatements (utils/sh_utils.py eval_sh, scene/gaussian_model.py
 * get_view2gaussian/get_covariance, utils/tetmesh.py) through tests/golden/, (b) operator
 * by operator against the vendored GLM (oracle/check_glm.cpp), and (c) end to end against the
 * reference CUDA sources themselves compiled for gfx950 (oracle/_ref, GPU tests) and (d) against the
 * same sources compiled for the HOST on top of tests/hipemu (oracle/_ref/libgof_cudaref_host.so,
 * tests/test_reference_host.py in the CPU suite): K1 floats, keys, sorted list, ranges bit-exact.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/gof_hip.h"
#include "gof_oracle.h"
#include "glmlike.h"

using gl::M3; using gl::M4; using gl::V3;

namespace {

constexpr int BLOCK_X = 16, BLOCK_Y = 16, BLOCK_SIZE = 256;   // config.h:15-17
constexpr int MAX_NUM_CONTRIBUTORS = 256;                      // auxiliary.h:26
constexpr int MAX_NUM_PROJECTED = 256;                         // auxiliary.h:34
#define NEAR_PLANE 0.2                                         /* auxiliary.h:27 (double literal) */
#define FAR_PLANE 100.0                                        /* auxiliary.h:28 (double literal) */

const float SH_C0 = 0.28209479177387814f;
const float SH_C1 = 0.4886025119029199f;
const float SH_C2[] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f };
const float SH_C3[] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };

thread_local std::string g_err;

// ---------------------------------------------------------------------------------------------
// exp: deterministic fp32 exp (Cephes expf scheme), identical op sequence in the HIP kernels.
// |error| <= 1 ulp on [-87, 88]; arguments are clamped to that range.
// ---------------------------------------------------------------------------------------------
inline float oexpf(float x)
{
    if (x < -87.0f) x = -87.0f;
    if (x > 88.0f) x = 88.0f;
    const float n = rintf(x * 1.44269504088896341f);       // round-to-nearest-even
    float r = fmaf(n, -0.693359375f, x);                     // exact: ln2_hi has 9 significant bits
    r = fmaf(n, 2.12194440e-4f, r);                          // -ln2_lo
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float y = fmaf(p, r2, r);
    y = y + 1.0f;
    return ldexpf(y, (int)n);
}

// auxiliary.h:59-62
inline float ndc2Pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:64-74.  grid is unsigned in the reference (dim3); the inner max() yields a
// non-negative int, so min(unsigned, int) is an unsigned min.
inline void getRect(float px, float py, int max_radius, uint32_t& minx, uint32_t& miny, uint32_t& maxx, uint32_t& maxy, uint32_t gx, uint32_t gy)
{
    minx = std::min(gx, (uint32_t)std::max((int)0, (int)((px - max_radius) / BLOCK_X)));
    miny = std::min(gy, (uint32_t)std::max((int)0, (int)((py - max_radius) / BLOCK_Y)));
    maxx = std::min(gx, (uint32_t)std::max((int)0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    maxy = std::min(gy, (uint32_t)std::max((int)0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

// auxiliary.h:86-94
inline V3 transformPoint4x3(const V3& p, const float* m)
{
    return V3{ m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
               m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
               m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] };
}
struct F4 { float x, y, z, w; };
// auxiliary.h:106-115
inline F4 transformPoint4x4(const V3& p, const float* m)
{
    return F4{ m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
               m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
               m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
               m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] };
}

// auxiliary.h:177-202; returns false when culled; *prefilter_violation set when prefiltered && culled
inline bool in_frustum(int idx, const float* orig_points, const float* viewmatrix, const float* projmatrix, bool prefiltered, V3& p_view, bool* violation)
{
    V3 p_orig = { orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2] };
    p_view = transformPoint4x3(p_orig, viewmatrix);
    if (p_view.z <= 0.2f) {
        if (prefiltered && violation) *violation = true;
        return false;
    }
    (void)projmatrix;
    return true;
}

// forward.cu:20-71
inline V3 computeColorFromSH(int idx, int deg, int max_coeffs, const float* means, const float* campos, const float* shs, uint8_t* clamped)
{
    V3 pos = { means[3 * idx], means[3 * idx + 1], means[3 * idx + 2] };
    V3 cam = { campos[0], campos[1], campos[2] };
    V3 dir = pos - cam;
    dir = dir / gl::length(dir);

    const V3* sh = reinterpret_cast<const V3*>(shs) + (size_t)idx * max_coeffs;
    V3 result = SH_C0 * sh[0];
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        result = result - SH_C1 * y * sh[1] + SH_C1 * z * sh[2] - SH_C1 * x * sh[3];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            result = result +
                SH_C2[0] * xy * sh[4] +
                SH_C2[1] * yz * sh[5] +
                SH_C2[2] * (2.0f * zz - xx - yy) * sh[6] +
                SH_C2[3] * xz * sh[7] +
                SH_C2[4] * (xx - yy) * sh[8];
            if (deg > 2) {
                result = result +
                    SH_C3[0] * y * (3.0f * xx - yy) * sh[9] +
                    SH_C3[1] * xy * z * sh[10] +
                    SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11] +
                    SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                    SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13] +
                    SH_C3[5] * z * (xx - yy) * sh[14] +
                    SH_C3[6] * x * (xx - 3.0f * yy) * sh[15];
            }
        }
    }
    result = result + V3{ 0.5f, 0.5f, 0.5f };
    clamped[3 * idx + 0] = (result.x < 0);
    clamped[3 * idx + 1] = (result.y < 0);
    clamped[3 * idx + 2] = (result.z < 0);
    return V3{ std::max(result.x, 0.0f), std::max(result.y, 0.0f), std::max(result.z, 0.0f) };
}

inline M3 quatToR(const float* rot)
{
    // forward.cu:138-149 (rotation used as given, NOT normalised)
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    return gl::mat3(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

// forward.cu:129-163
inline void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D)
{
    M3 S = gl::mat3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S[0][0] = mod * scale[0];
    S[1][1] = mod * scale[1];
    S[2][2] = mod * scale[2];
    M3 R = quatToR(rot);
    M3 M = gl::mul(S, R);
    M3 Sigma = gl::mul(gl::transpose(M), M);
    cov3D[0] = Sigma[0][0]; cov3D[1] = Sigma[0][1]; cov3D[2] = Sigma[0][2];
    cov3D[3] = Sigma[1][1]; cov3D[4] = Sigma[1][2]; cov3D[5] = Sigma[2][2];
}

// forward.cu:74-124
inline F4 computeCov2D(const V3& mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, float kernel_size, const float* cov3D, const float* viewmatrix)
{
    V3 t = transformPoint4x3(mean, viewmatrix);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t.x / t.z;
    const float tytz = t.y / t.z;
    t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
    t.y = std::min(limy, std::max(-limy, tytz)) * t.z;

    M3 J = gl::mat3(
        focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z),
        0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z),
        0, 0, 0);
    M3 W = gl::mat3(
        viewmatrix[0], viewmatrix[4], viewmatrix[8],
        viewmatrix[1], viewmatrix[5], viewmatrix[9],
        viewmatrix[2], viewmatrix[6], viewmatrix[10]);
    M3 T = gl::mul(W, J);
    M3 Vrk = gl::mat3(
        cov3D[0], cov3D[1], cov3D[2],
        cov3D[1], cov3D[3], cov3D[4],
        cov3D[2], cov3D[4], cov3D[5]);
    M3 cov = gl::mul(gl::mul(gl::transpose(T), gl::transpose(Vrk)), T);

    // forward.cu:112-118: max(1e-6, float) is a double max, result narrowed to float
    const float det_0 = (float)std::max(1e-6, (double)(cov[0][0] * cov[1][1] - cov[0][1] * cov[0][1]));
    const float det_1 = (float)std::max(1e-6, (double)((cov[0][0] + kernel_size) * (cov[1][1] + kernel_size) - cov[0][1] * cov[0][1]));
    float coef = (float)sqrt((double)det_0 / ((double)det_1 + 1e-6) + 1e-6);
    if ((double)det_0 <= 1e-6 || (double)det_1 <= 1e-6) coef = 0.0f;

    cov[0][0] += kernel_size;
    cov[1][1] += kernel_size;
    return F4{ cov[0][0], cov[0][1], cov[1][1], coef };
}

struct V2GIntermediates {
    M3 R;            // rotation from quaternion
    M4 W2V, G2V;
    M3 R_transpose;  // rows of G2V's 3x3 block as columns (forward.cu:215-219)
    V3 t, t2;
    double Sx, Sy, Sz;   // S_inv_square
    M3 S_inv_square_R;
};

// forward.cu:168-279 up to (and including) S_inv_square_R; shared with the backward (backward.cu:394-444)
inline void v2gIntermediates(const float* scale, const V3& mean, const float* rot, const float* viewmatrix, V2GIntermediates& I)
{
    I.R = quatToR(rot);
    const M3& R = I.R;
    M4 G2W = gl::mat4(
        R[0][0], R[1][0], R[2][0], 0.0f,
        R[0][1], R[1][1], R[2][1], 0.0f,
        R[0][2], R[1][2], R[2][2], 0.0f,
        mean.x, mean.y, mean.z, 1.0f);
    I.W2V = gl::mat4(
        viewmatrix[0], viewmatrix[1], viewmatrix[2], viewmatrix[3],
        viewmatrix[4], viewmatrix[5], viewmatrix[6], viewmatrix[7],
        viewmatrix[8], viewmatrix[9], viewmatrix[10], viewmatrix[11],
        viewmatrix[12], viewmatrix[13], viewmatrix[14], viewmatrix[15]);
    I.G2V = gl::mul(I.W2V, G2W);
    const M4& G2V = I.G2V;
    I.R_transpose = gl::mat3(
        G2V[0][0], G2V[1][0], G2V[2][0],
        G2V[0][1], G2V[1][1], G2V[2][1],
        G2V[0][2], G2V[1][2], G2V[2][2]);
    I.t = V3{ G2V[3][0], G2V[3][1], G2V[3][2] };
    I.t2 = gl::mul(gl::neg(I.R_transpose), I.t);
    // forward.cu:255: 1.0f / ((double)s * s + 1e-7)  (double)
    I.Sx = 1.0f / ((double)scale[0] * scale[0] + 1e-7);
    I.Sy = 1.0f / ((double)scale[1] * scale[1] + 1e-7);
    I.Sz = 1.0f / ((double)scale[2] * scale[2] + 1e-7);
    const M3& Rt = I.R_transpose;
    // forward.cu:257-261: double * float products narrowed to float by the mat3 constructor
    I.S_inv_square_R = gl::mat3(
        (float)(I.Sx * Rt[0][0]), (float)(I.Sy * Rt[0][1]), (float)(I.Sz * Rt[0][2]),
        (float)(I.Sx * Rt[1][0]), (float)(I.Sy * Rt[1][1]), (float)(I.Sz * Rt[1][2]),
        (float)(I.Sx * Rt[2][0]), (float)(I.Sy * Rt[2][1]), (float)(I.Sz * Rt[2][2]));
}

// forward.cu:168-279
inline void computeView2Gaussian(const float* scale, const V3& mean, const float* rot, const float* viewmatrix, float* view2gaussian)
{
    V2GIntermediates I;
    v2gIntermediates(scale, mean, rot, viewmatrix, I);
    const V3& t2 = I.t2;
    // forward.cu:256: float*float (fp32) then * double, summed in double
    double C = (double)(t2.x * t2.x) * I.Sx + (double)(t2.y * t2.y) * I.Sy + (double)(t2.z * t2.z) * I.Sz;
    V3 B = gl::mul(t2, I.S_inv_square_R);
    M3 Sigma = gl::mul(gl::transpose(I.R_transpose), I.S_inv_square_R);
    view2gaussian[0] = Sigma[0][0];
    view2gaussian[1] = Sigma[0][1];
    view2gaussian[2] = Sigma[0][2];
    view2gaussian[3] = Sigma[1][1];
    view2gaussian[4] = Sigma[1][2];
    view2gaussian[5] = Sigma[2][2];
    view2gaussian[6] = B.x;
    view2gaussian[7] = B.y;
    view2gaussian[8] = B.z;
    view2gaussian[9] = (float)C;
}

// rasterizer_impl.cu:35-50
inline uint32_t getHigherMsb(uint32_t n)
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

} // namespace

// =============================================================================================
// State
// =============================================================================================
struct GofRefState {
    int P = 0, W = 0, H = 0, gx = 0, gy = 0;
    float focal_x = 0, focal_y = 0;
    // geometry (GeometryState, rasterizer_impl.h:29-45)
    std::vector<float> depths, means2D, cov3D, view2gaussian, conic_opacity, rgb;
    std::vector<uint8_t> clamped;
    std::vector<int32_t> radii;
    std::vector<uint32_t> tiles_touched, point_offsets;
    // binning
    uint32_t num_rendered = 0;
    std::vector<uint64_t> keys_unsorted, keys;
    std::vector<uint32_t> vals_unsorted, point_list;
    // image
    std::vector<uint32_t> ranges;       // [T,2]
    std::vector<float> final_T;         // [4,H,W]
    std::vector<uint32_t> n_contrib;    // [2,H,W]
    std::vector<uint32_t> tile_walked;  // [T] entries consumed before the whole tile is done (R_visited)
    // integrate-only
    uint32_t num_integrated = 0;
    std::vector<float> p_depths, points2D;
    std::vector<uint32_t> p_tiles_touched, p_offsets, p_list, p_ranges;
    std::vector<uint64_t> p_keys;
    // feature pointer selection
    const float* features = nullptr;
    const float* v2g = nullptr;
    bool prefilter_violation = false;
};

namespace {

// forward.cu:283-404 for all Gaussians
void preprocess_all(const GofRasterArgs& a, GofRefState& s)
{
    const int P = a.P;
    s.P = P; s.W = a.W; s.H = a.H;
    s.gx = (a.W + BLOCK_X - 1) / BLOCK_X;
    s.gy = (a.H + BLOCK_Y - 1) / BLOCK_Y;
    s.focal_y = a.H / (2.0f * a.tan_fovy);   // rasterizer_impl.cu:274-275
    s.focal_x = a.W / (2.0f * a.tan_fovx);
    s.depths.assign(P, 0.f); s.means2D.assign(2 * (size_t)P, 0.f); s.cov3D.assign(6 * (size_t)P, 0.f);
    s.view2gaussian.assign(10 * (size_t)P, 0.f); s.conic_opacity.assign(4 * (size_t)P, 0.f); s.rgb.assign(3 * (size_t)P, 0.f);
    s.clamped.assign(3 * (size_t)P, 0); s.radii.assign(P, 0); s.tiles_touched.assign(P, 0); s.point_offsets.assign(P, 0);
    bool violation = false;

#pragma omp parallel for schedule(static) reduction(|| : violation)
    for (int idx = 0; idx < P; idx++) {
        s.radii[idx] = 0;
        s.tiles_touched[idx] = 0;
        V3 p_view;
        bool viol = false;
        if (!in_frustum(idx, a.means3D, a.viewmatrix, a.projmatrix, a.prefiltered != 0, p_view, &viol)) { violation = violation || viol; continue; }

        V3 p_orig = { a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2] };
        F4 p_hom = transformPoint4x4(p_orig, a.projmatrix);
        float p_w = 1.0f / (p_hom.w + 0.0000001f);
        V3 p_proj = { p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w };

        const float* scale = a.scales ? a.scales + 3 * (size_t)idx : nullptr;
        const float* rot = a.rotations ? a.rotations + 4 * (size_t)idx : nullptr;

        const float* cov3D;
        if (a.cov3D_precomp != nullptr) cov3D = a.cov3D_precomp + (size_t)idx * 6;
        else {
            computeCov3D(scale, a.scale_modifier, rot, &s.cov3D[(size_t)idx * 6]);
            cov3D = &s.cov3D[(size_t)idx * 6];
        }
        F4 cov = computeCov2D(p_orig, s.focal_x, s.focal_y, a.tan_fovx, a.tan_fovy, a.kernel_size, cov3D, a.viewmatrix);

        float det = (cov.x * cov.z - cov.y * cov.y);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = { cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv };

        float mid = 0.5f * (cov.x + cov.z);
        float lambda1 = mid + sqrtf(std::max(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(std::max(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(std::max(lambda1, lambda2)));
        float pix = ndc2Pix(p_proj.x, a.W), piy = ndc2Pix(p_proj.y, a.H);
        uint32_t minx, miny, maxx, maxy;
        getRect(pix, piy, (int)my_radius, minx, miny, maxx, maxy, s.gx, s.gy);
        if ((maxx - minx) * (maxy - miny) == 0) continue;

        if (a.colors_precomp == nullptr) {
            V3 result = computeColorFromSH(idx, a.D, a.M, a.means3D, a.campos, a.shs, s.clamped.data());
            s.rgb[idx * 3 + 0] = result.x; s.rgb[idx * 3 + 1] = result.y; s.rgb[idx * 3 + 2] = result.z;
        }
        s.depths[idx] = p_view.z;
        s.radii[idx] = (int)my_radius;
        s.means2D[2 * idx] = pix; s.means2D[2 * idx + 1] = piy;
        s.conic_opacity[4 * idx + 0] = conic[0]; s.conic_opacity[4 * idx + 1] = conic[1];
        s.conic_opacity[4 * idx + 2] = conic[2]; s.conic_opacity[4 * idx + 3] = a.opacities[idx] * cov.w;
        s.tiles_touched[idx] = (maxy - miny) * (maxx - minx);
        if (a.view2gaussian_precomp == nullptr)
            computeView2Gaussian(scale, p_orig, rot, a.viewmatrix, &s.view2gaussian[(size_t)idx * 10]);
    }
    s.prefilter_violation = violation;
    s.features = a.colors_precomp ? a.colors_precomp : s.rgb.data();          // rasterizer_impl.cu:377
    s.v2g = a.view2gaussian_precomp ? a.view2gaussian_precomp : s.view2gaussian.data();  // :379
}

// rasterizer_impl.cu:332-373: scan, duplicateWithKeys, stable sort on the low 32+bit bits, ranges
void bin_gaussians(GofRefState& s)
{
    const int P = s.P;
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += s.tiles_touched[i]; s.point_offsets[i] = acc; }
    s.num_rendered = P > 0 ? s.point_offsets[P - 1] : 0;
    const uint32_t R = s.num_rendered;
    s.keys_unsorted.assign(R, 0); s.vals_unsorted.assign(R, 0);
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (s.radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : s.point_offsets[idx - 1];
            uint32_t minx, miny, maxx, maxy;
            getRect(s.means2D[2 * idx], s.means2D[2 * idx + 1], s.radii[idx], minx, miny, maxx, maxy, s.gx, s.gy);
            uint32_t dbits; memcpy(&dbits, &s.depths[idx], 4);
            for (int y = miny; y < (int)maxy; y++)
                for (int x = minx; x < (int)maxx; x++) {
                    uint64_t key = (uint64_t)(y * s.gx + x);
                    key <<= 32;
                    key |= dbits;
                    s.keys_unsorted[off] = key;
                    s.vals_unsorted[off] = idx;
                    off++;
                }
        }
    }
    const int bit = getHigherMsb((uint32_t)(s.gx * s.gy));
    const int nbits = 32 + bit;
    const uint64_t mask = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1);
    std::vector<uint32_t> order(R);
    std::iota(order.begin(), order.end(), 0u);
    const uint64_t* ku = s.keys_unsorted.data();
    std::stable_sort(order.begin(), order.end(), [ku, mask](uint32_t l, uint32_t r) { return (ku[l] & mask) < (ku[r] & mask); });
    s.keys.resize(R); s.point_list.resize(R);
    for (uint32_t i = 0; i < R; i++) { s.keys[i] = ku[order[i]]; s.point_list[i] = s.vals_unsorted[order[i]]; }

    const int T = s.gx * s.gy;
    s.ranges.assign(2 * (size_t)T, 0);     // cudaMemset, rasterizer_impl.cu:365
    for (uint32_t idx = 0; idx < R; idx++) {   // identifyTileRanges, :149-171
        uint32_t currtile = (uint32_t)(s.keys[idx] >> 32);
        if (idx == 0) s.ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(s.keys[idx - 1] >> 32);
            if (currtile != prevtile) { s.ranges[2 * prevtile + 1] = idx; s.ranges[2 * currtile] = idx; }
        }
        if (idx == R - 1) s.ranges[2 * currtile + 1] = R;
    }
}

// The per-pair quantities of forward.cu:499-533 / backward.cu:771-804.
struct Pair {
    float normal[3];
    double AA, BB;
    float CC, t, power, G, alpha;
    bool skip;   // t <= NEAR_PLANE or alpha < 1/255
};
inline void eval_pair(const float* v, float w, float rx, float ry, Pair& p)
{
    p.normal[0] = v[0] * rx + v[1] * ry + v[2];
    p.normal[1] = v[1] * rx + v[3] * ry + v[4];
    p.normal[2] = v[2] * rx + v[4] * ry + v[5];
    p.AA = (double)(rx * p.normal[0] + ry * p.normal[1] + p.normal[2]);   // fp32 value widened
    p.BB = (double)(2 * (v[6] * rx + v[7] * ry + v[8]));                  // 2 * float is fp32
    p.CC = v[9];
    p.t = (float)(-p.BB / (2 * p.AA));
    p.skip = false;
    if ((double)p.t <= NEAR_PLANE) { p.skip = true; return; }
    double min_value = -(p.BB / p.AA) * (p.BB / 4.) + (double)p.CC;
    float power = (float)(-0.5f * min_value);
    if (power > 0.0f) power = 0.0f;
    p.power = power;
    p.G = oexpf(power);
    p.alpha = std::min(0.99f, w * p.G);
    if (p.alpha < 1.0f / 255.0f) p.skip = true;
}

// forward.cu:409-612
void render_forward(const GofRasterArgs& a, GofRefState& s, float* out_color)
{
    const int W = a.W, H = a.H;
    const size_t HW = (size_t)W * H;
    s.final_T.assign(4 * HW, 0.f);
    s.n_contrib.assign(2 * HW, 0u);
    s.tile_walked.assign((size_t)s.gx * s.gy, 0u);
    const float focal_x = s.focal_x, focal_y = s.focal_y;
    const float* features = s.features;
    const float* v2g = s.v2g;

#pragma omp parallel for schedule(dynamic, 4) collapse(2)
    for (int ty = 0; ty < s.gy; ty++)
        for (int tx = 0; tx < s.gx; tx++) {
            const int tile = ty * s.gx + tx;
            const uint32_t r0 = s.ranges[2 * tile], r1 = s.ranges[2 * tile + 1];
            uint32_t walked_max = 0;
            for (int ly = 0; ly < BLOCK_Y; ly++)
                for (int lx = 0; lx < BLOCK_X; lx++) {
                    const uint32_t px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                    if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                    const uint32_t pix_id = W * py + px;
                    const float pixfx = (float)px + 0.5f, pixfy = (float)py + 0.5f;
                    // forward.cu:448  (pixf - W/2.) / focal in double, narrowed
                    const float rayx = (float)(((double)pixfx - W / 2.) / (double)focal_x);
                    const float rayy = (float)(((double)pixfy - H / 2.) / (double)focal_y);

                    float T = 1.0f;
                    uint32_t contributor = 0, last_contributor = 0, max_contributor = (uint32_t)-1;
                    float C[3 * 2 + 2] = { 0 };
                    float dist1 = 0, dist2 = 0, distortion = 0;

                    for (uint32_t k = r0; k < r1; k++) {
                        contributor++;
                        const uint32_t gid = s.point_list[k];
                        const float w = s.conic_opacity[4 * (size_t)gid + 3];
                        Pair p;
                        eval_pair(v2g + 10 * (size_t)gid, w, rayx, rayy, p);
                        if (p.skip) continue;
                        const float alpha = p.alpha, t = p.t;
                        float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) break;   // done = true (contributor already counted)

                        const float max_t = t;
                        const float mapped_max_t = (float)((FAR_PLANE * max_t - FAR_PLANE * NEAR_PLANE) / ((FAR_PLANE - NEAR_PLANE) * max_t));
                        float length = (float)sqrt((double)(p.normal[0] * p.normal[0] + p.normal[1] * p.normal[1] + p.normal[2] * p.normal[2]) + 1e-7);
                        const float nn[3] = { -p.normal[0] / length, -p.normal[1] / length, -p.normal[2] / length };

                        float A = 1 - T;
                        float error = mapped_max_t * mapped_max_t * A + dist2 - 2 * mapped_max_t * dist1;
                        distortion += error * alpha * T;
                        dist1 += mapped_max_t * alpha * T;
                        dist2 += mapped_max_t * mapped_max_t * alpha * T;

                        for (int ch = 0; ch < 3; ch++) C[ch] += features[(size_t)gid * 3 + ch] * alpha * T;
                        for (int ch = 0; ch < 3; ch++) C[3 + ch] += nn[ch] * alpha * T;
                        if ((double)T > 0.5) { C[6] = t; max_contributor = contributor; }
                        C[7] += alpha * T;
                        T = test_T;
                        last_contributor = contributor;
                    }
                    walked_max = std::max(walked_max, contributor);

                    const float distortion_before_normalized = distortion;
                    distortion = (float)((double)distortion / ((double)((1 - T) * (1 - T)) + 1e-7));
                    s.final_T[pix_id] = T;
                    s.final_T[pix_id + HW] = dist1;
                    s.final_T[pix_id + 2 * HW] = dist2;
                    s.final_T[pix_id + 3 * HW] = distortion_before_normalized;
                    s.n_contrib[pix_id] = last_contributor;
                    s.n_contrib[pix_id + HW] = max_contributor;
                    for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * a.background[ch];
                    for (int ch = 0; ch < 3; ch++) out_color[(3 + ch) * HW + pix_id] = C[3 + ch];
                    out_color[6 * HW + pix_id] = C[6];
                    out_color[7 * HW + pix_id] = C[7];
                    out_color[8 * HW + pix_id] = distortion;
                }
            s.tile_walked[tile] = walked_max;
        }
}

} // namespace

// =============================================================================================
// C API
// =============================================================================================
extern "C" {

const char* gofref_last_error(void) { return g_err.c_str(); }

int gofref_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

float gofref_expf(float x) { return oexpf(x); }

static int check_args(const GofRasterArgs* a)
{
    if (!a) { g_err = "args is NULL"; return GOF_E_INVALID; }
    if (a->P < 0 || a->W <= 0 || a->H <= 0) { g_err = "bad P/W/H"; return GOF_E_INVALID; }
    if ((a->shs == nullptr) == (a->colors_precomp == nullptr) && a->P > 0) { g_err = "provide exactly one of shs / colors_precomp"; return GOF_E_INVALID; }
    if (a->P > 0 && !a->cov3D_precomp && (!a->scales || !a->rotations)) { g_err = "scales/rotations or cov3D_precomp required"; return GOF_E_INVALID; }
    if (a->P > 0 && !a->view2gaussian_precomp && (!a->scales || !a->rotations)) { g_err = "scales/rotations required to compute view2gaussian"; return GOF_E_INVALID; }
    return GOF_OK;
}

int gofref_forward(const GofRasterArgs* a, float* out_color, int32_t* radii, GofRefState** state_out)
{
    int rc = check_args(a);
    if (rc) return rc;
    GofRefState* s = new GofRefState();
    const size_t HW = (size_t)a->W * a->H;
    // rasterize_points.cu:68-69: out_color = 0, radii = 0; P == 0 skips everything (:85)
    std::fill(out_color, out_color + 9 * HW, 0.f);
    if (a->P != 0) {
        preprocess_all(*a, *s);
        bin_gaussians(*s);
        render_forward(*a, *s, out_color);
        if (radii) memcpy(radii, s->radii.data(), sizeof(int32_t) * a->P);
    } else {
        s->W = a->W; s->H = a->H;
    }
    if (state_out) *state_out = s; else delete s;
    return GOF_OK;
}

void gofref_free(GofRefState* s) { delete s; }

uint32_t gofref_num_rendered(const GofRefState* s) { return s->num_rendered; }
uint32_t gofref_num_integrated(const GofRefState* s) { return s->num_integrated; }

int64_t gofref_fetch(const GofRefState* s, const char* name, void* dst, size_t dst_bytes)
{
    const void* src = nullptr; size_t bytes = 0; int64_t count = 0;
#define F(NAME, VEC) if (!strcmp(name, NAME)) { src = s->VEC.data(); bytes = s->VEC.size() * sizeof(s->VEC[0]); count = (int64_t)s->VEC.size(); }
    F("depths", depths) F("means2D", means2D) F("cov3D", cov3D) F("view2gaussian", view2gaussian)
    F("conic_opacity", conic_opacity) F("rgb", rgb) F("clamped", clamped) F("radii", radii)
    F("tiles_touched", tiles_touched) F("point_offsets", point_offsets)
    F("point_list_keys_unsorted", keys_unsorted) F("point_list_keys", keys) F("point_list_unsorted", vals_unsorted)
    F("point_list", point_list) F("ranges", ranges) F("final_T", final_T) F("n_contrib", n_contrib)
    F("tile_walked", tile_walked)
    F("p_depths", p_depths) F("points2D", points2D) F("p_tiles_touched", p_tiles_touched) F("p_list", p_list)
    F("p_ranges", p_ranges) F("p_keys", p_keys)
#undef F
    if (!src && count == 0 && bytes == 0) {
        // distinguish "unknown name" from "empty array"
        static const char* known[] = { "depths","means2D","cov3D","view2gaussian","conic_opacity","rgb","clamped","radii","tiles_touched",
            "point_offsets","point_list_keys_unsorted","point_list_keys","point_list_unsorted","point_list","ranges","final_T","n_contrib",
            "tile_walked","p_depths","points2D","p_tiles_touched","p_list","p_ranges","p_keys" };
        bool ok = false;
        for (const char* k : known) if (!strcmp(k, name)) ok = true;
        if (!ok) { g_err = std::string("unknown array ") + name; return -1; }
        return 0;
    }
    if (dst) {
        if (dst_bytes < bytes) { g_err = "dst too small"; return -2; }
        memcpy(dst, src, bytes);
    }
    return count;
}

} // extern "C"

#include "gof_oracle_backward.inc"
#include "gof_oracle_integrate.inc"
#include "gof_oracle_mtets.inc"
