// preprocess.hip -- per-Gaussian / per-point streaming kernels (HBM-bound).
//
//   preprocess_fwd   K1  replaces preprocessCUDA        (reference forward.cu:283-404)
//   preprocess_bwd   K9  replaces backward preprocessCUDA (reference backward.cu:593-631)
//   preprocess_points K10 replaces preprocessPointsCUDA (reference forward.cu:722-766)
//   mark_visible     K15 replaces checkFrustum          (reference rasterizer_impl.cu:54-66)
//
// Layout decisions (MI355X): one thread per Gaussian, 256-thread blocks; inputs are read with
// the widest loads the reference's [P,k] row-major layouts allow (16-byte loads for rotations
// and the 192-byte SH block, which is 16-byte aligned); outputs go to the 64-byte SplatRec line
// (4 x 16-byte stores) so that the blend kernels gather one aligned line per tile-list entry.
// cov3D is NOT stored: its only consumer in the reference is dead code (backward.cu:627-630).
#include "gof_common.h"

namespace gof {

__constant__ float SH_C0 = 0.28209479177387814f;
__constant__ float SH_C1 = 0.4886025119029199f;
__constant__ float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f };
__constant__ float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };


// SH -> RGB (forward.cu:20-71).  sh0 points at this Gaussian's coefficient 0, shp at where its coefficient block would start so
// that coefficient k >= 1 is shp[3k..3k+2] (the same address as sh0 for one [M][3] block; _features_rest's row minus 3 floats
// when DC and higher bands are separate tensors).
__device__ __forceinline__ V3 sh_to_rgb(int deg, V3 pos, V3 campos, const float* __restrict__ sh0, const float* __restrict__ shp, uint32_t& clamp_bits)
{
    V3 dir = pos - campos;
    dir = dir / sqrtf(dot3(dir, dir));
    const V3* sh = reinterpret_cast<const V3*>(shp);
    V3 result = SH_C0 * reinterpret_cast<const V3*>(sh0)[0];
    if (deg > 0) {
        const float x = dir.x, y = dir.y, z = dir.z;
        result = result - SH_C1 * y * sh[1] + SH_C1 * z * sh[2] - SH_C1 * x * sh[3];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
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
    clamp_bits = (result.x < 0 ? 1u : 0u) | (result.y < 0 ? 2u : 0u) | (result.z < 0 ? 4u : 0u);
    return V3{ fmaxf(result.x, 0.0f), fmaxf(result.y, 0.0f), fmaxf(result.z, 0.0f) };
}

// Intermediates shared by computeView2Gaussian forward and backward (forward.cu:168-261)
struct V2GInter {
    M4 G2V; M4 W2V;
    M3 Rt;           // R_transpose
    V3 t2;
    double Sx, Sy, Sz;
    M3 SR;           // S_inv_square_R
};
__device__ __forceinline__ void v2g_intermediates(V3 scale, V3 mean, float4 rot, const float* __restrict__ vm, V2GInter& I)
{
    const M3 R = quat_to_R(rot.x, rot.y, rot.z, rot.w);
    M4 G2W;
    G2W.m[0][0] = R.m[0][0]; G2W.m[0][1] = R.m[1][0]; G2W.m[0][2] = R.m[2][0]; G2W.m[0][3] = 0.0f;
    G2W.m[1][0] = R.m[0][1]; G2W.m[1][1] = R.m[1][1]; G2W.m[1][2] = R.m[2][1]; G2W.m[1][3] = 0.0f;
    G2W.m[2][0] = R.m[0][2]; G2W.m[2][1] = R.m[1][2]; G2W.m[2][2] = R.m[2][2]; G2W.m[2][3] = 0.0f;
    G2W.m[3][0] = mean.x; G2W.m[3][1] = mean.y; G2W.m[3][2] = mean.z; G2W.m[3][3] = 1.0f;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) I.W2V.m[c][r] = vm[4 * c + r];
    I.G2V = mul(I.W2V, G2W);
    const M4& G = I.G2V;
    I.Rt = mk3(G.m[0][0], G.m[1][0], G.m[2][0],
               G.m[0][1], G.m[1][1], G.m[2][1],
               G.m[0][2], G.m[1][2], G.m[2][2]);
    const V3 t = { G.m[3][0], G.m[3][1], G.m[3][2] };
    I.t2 = mul(neg(I.Rt), t);
    I.Sx = 1.0f / ((double)scale.x * scale.x + 1e-7);
    I.Sy = 1.0f / ((double)scale.y * scale.y + 1e-7);
    I.Sz = 1.0f / ((double)scale.z * scale.z + 1e-7);
    const M3& Rt = I.Rt;
    I.SR = mk3((float)(I.Sx * Rt.m[0][0]), (float)(I.Sy * Rt.m[0][1]), (float)(I.Sz * Rt.m[0][2]),
               (float)(I.Sx * Rt.m[1][0]), (float)(I.Sy * Rt.m[1][1]), (float)(I.Sz * Rt.m[1][2]),
               (float)(I.Sx * Rt.m[2][0]), (float)(I.Sy * Rt.m[2][1]), (float)(I.Sz * Rt.m[2][2]));
}

// Conservative footprint of the region where this splat can reach alpha >= 1/255 -- as a PIXEL bounding box (the opacity-field
// query's wave-level prefilter) and as a CONIC in ray space (the cull scans of blend_forward and integrate_pixels) -- used to skip
// (pixel, splat) pairs without touching the per-pair arithmetic.
//
// alpha = w * exp(-min_value / 2) >= 1/255  <=>  min_value <= m0 = 2 ln(255 w); min_value(ray) is the
// minimum of the Mahalanobis distance along the ray, so the candidate rays are those that hit the
// ellipsoid {x : (x - mu)^T Sigma' (x - mu) <= m0} (view space, Sigma'^-1 = A diag(s^2 + 1e-7) A^T).  The
// image of an ellipsoid under a pinhole at the origin is a conic with dual C* = mu mu^T - k Cov; the
// vertical / horizontal tangent lines give the box (valid while the ellipsoid stays in front of the
// camera plane, otherwise the box is left unbounded).  Everything is evaluated in fp64 from the
// well-conditioned view-space covariance (no inversion).
//
// The blend evaluates min_value = CC - BB^2 / (4 AA) from fp32-rounded coefficients: the prelude (normal = Sigma' r, AA = r.normal,
// BB = 2 b.r) in fp32, the rest in fp64.  Forward error, eps = 2^-24: every stored Sigma'_ij is off by <= 4 eps lambda_max (product
// chain of preprocess), so AA by <= 12 eps lambda_max |r|^2 from the coefficients + <= 10.4 eps lambda_max |r|^2 from its own 7
// roundings (|| |Sigma'| ||_2 <= sqrt(3) lambda_max); BB by <= 24 eps lambda_max |mu| |r| (b = Sigma' mu, 5 roundings, doubled) + <= 10
// eps lambda_max |mu| |r| from the fp32 rounding of the transformed mean inside b; CC by <= 11 eps lambda_max |mu|^2.  With
// d(min_value)/dBB = t, d/dAA = t^2 and t |r| ~ |mu| at the ray's closest approach:  |delta min_value| <~ (22.4 + 34 + 11) eps
// lambda_max |mu|^2 = 67.4 eps = 4.02e-6 lambda_max |mu|^2 (the opacity-field query forms BB / AA as an fp32 quotient first,
// forward.cu:936: one more eps of X <= lambda_max |mu|^2, 4.08e-6).  The level is therefore raised to k = m0 + Delta with
//     Delta = GOF_BOX_C * lambda_max * |mu|^2 + 0.05,   GOF_BOX_C = 4.2e-6
// (that bound, rounded up: a sum of worst cases, 8x what was ever observed -- see below; rounds 1-2 carried another factor 1.5, which cost 4 % of
// the forward blend's heavy trips at S1M, 5 % on the clustered scene, for nothing the bound does not already cover; the constant
// term covers the fp32 rounding of `power`, the <= 1 ulp exp and the threshold compare with a wide
// margin), which keeps box and conic conservative with respect to the arithmetic the blend actually performs (for sub-pixel,
// far-away splats they grow accordingly).  For cond(Sigma') > 1e4, a non-orthonormal frame, or an ellipsoid that reaches the
// camera plane, no statement is made (unbounded box, all-zero conic = always a candidate).  Measured: an instrumented build
// (-DGOF_CULL_AUDIT) counts the pairs the exact path accepts outside the conic -- 0 on S1M and the whole scene table
// (tests/test_parity_gpu.py::test_the_cull_scan_drops_no_pair_the_exact_path_accepts); with the constant lowered to 2e-7 / 0 the
// same count was 13 / 2438 of 1.24e8 accepted pairs at S1M (round-1 measurement), none at 1e-6; round 3 (the audit build run from
// source on the host by the test suite): none at 5e-7 either, on S1M (posed and not) and S1M-clustered.
// Also emits the footprint CONIC in ray space (fc[0..1]): a ray r = (rx, ry, 1) meets the level-set ellipsoid iff
// g(r) = r^T M r <= 0 with M = (C - k) Sigma' - b b^T, b = Sigma' mu, C = mu^T Sigma' mu (min over t of the quadratic
// along the ray is C - (b.r)^2 / (r^T Sigma' r)).  M is evaluated in fp64 from the same well-conditioned factors as the
// box, then scaled to sum |M_ij| = 1 so that an fp32 evaluation of g carries an absolute error below ~1e-6 * max(1, |r|^2):
//   fc[0] = {m00, m01, m11, m02}, fc[1] = {m12, m22, q, zfront},  q = m00 hx^2 + m11 hy^2 - 2 |m01| hx hy  (hx, hy = half a pixel in
// ray units: the lower bound of g over the +-0.5 px corner sub-rays of integrate is g - |g_x| hx - |g_y| hy + q).
// All-zero coefficients mean "no statement" (never culls): unbounded / degenerate cases.
// zfront: a point of the level-set ellipsoid has view depth >= mu_z - sqrt(k / l_min) (Mahalanobis^2 >= l_min * dz^2), so a
// query point whose depth (the clamp of t in integrate's point pass, forward.cu:1172-1178) is below zfront cannot reach
// alpha >= 1/255 from this Gaussian; -1e30 = no statement.
// (round 5, cost: this function was ~45 % of preprocess_fwd's 670 fp64 instructions, and the kernel is bound by them, not by HBM: the
// level m0 from the fp32 logarithm -- its ~1e-6 relative error is 1e-5 of the 0.05 the level carries for exactly such things --, the
// three 1 / l_c taken as s_c^2 + 1e-7, which is what l_c is the reciprocal of (`var_*`), and one reciprocal for the four box edges)
// FULL (round 6): true = everything above -- what the opacity-field query reads (pixel box: its pixel-centric pass / the capped fallback,
// and the opt-in tight tile rectangles; q: the half-pixel allowance of the pixel-centric scan; zfront: the point pass).  false = the
// conic alone, which is all blend_forward reads of a footprint (fc[0], fc[1].xy): the pixel box (five symmetric products, two
// discriminants, two fp64 square roots, a division, four edges with ceil / floor) and zfront (a square root and a division) were ~40 %
// of this kernel's fp64 instructions, computed and stored by every training forward for nobody.  Then the box is returned UNBOUNDED,
// q = -1e30 ("every entry a candidate") and zfront = -1e30 ("no statement"): a query that is handed such a workspace all the same stays
// exact, only unculled.
template <bool FULL>
__device__ __forceinline__ float4 footprint_bbox(const V2GInter& I, V3 mu, float w, float focal_x, float focal_y, int W, int H, float4* fc, double var_x, double var_y, double var_z)
{
    fc[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    fc[1] = make_float4(0.f, 0.f, FULL ? 0.f : -1e30f, -1e30f);
    const float4 unbounded = make_float4(-1e30f, 1e30f, -1e30f, 1e30f);
    const float4 empty = make_float4(1e30f, -1e30f, 1e30f, -1e30f);
    if (!(w > 0.0f)) return empty;                                  // alpha <= 0 < 1/255 everywhere
    const double m0 = 2.0 * (double)logf(255.0f * w);
    if (!(m0 > -0.05)) return empty;                                // w < 1/255: can never reach the threshold
    const double lx = I.Sx, ly = I.Sy, lz = I.Sz;                   // eigenvalues of Sigma' = 1 / (s^2 + 1e-7)
    const double lmax = fmax(lx, fmax(ly, lz)), lmin = fmin(lx, fmin(ly, lz));
    if (!(lmax < 1e4 * lmin)) return unbounded;                     // cond(Sigma') >= 1e4 (or NaN): no statement (a product instead of round 5's fp64 division)
    {   // the closed form below needs an orthonormal frame (unit quaternion, rigid view matrix); otherwise leave it unbounded
        const M3& R = I.Rt;
        float dev = 0.f;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i; j < 3; j++) {
                const float d = R.m[i][0] * R.m[j][0] + R.m[i][1] * R.m[j][1] + R.m[i][2] * R.m[j][2] - (i == j ? 1.f : 0.f);
                dev = fmaxf(dev, fabsf(d));
            }
        if (!(dev < 1e-5f)) return unbounded;
    }
    const double mx = mu.x, my = mu.y, mz = mu.z;
    const double mu2 = mx * mx + my * my + mz * mz;
#ifndef GOF_BOX_C
#define GOF_BOX_C 4.2e-6
#endif
    const double k = m0 + 0.05 + GOF_BOX_C * lmax * mu2;
    // view-space covariance Cov = A diag(1/l) A^T with A = Rt^T (rows of Rt are the Gaussian axes in view space):
    // Cov_ij = sum_c Rt[i][c] * Rt[j][c] / l_c   (Rt.m[col][row] = G2V[row][col])
    const M3& Rt = I.Rt;
    const double ix = var_x, iy = var_y, iz = var_z;       // 1 / l_c up to the rounding of l_c's own division
    // A = G2V 3x3 block: A[r][c] = Rt.m[r][c]?  Rt = mk3(G00,G10,G20, G01,G11,G21, G02,G12,G22) -> Rt.m[c][r] = G2V.m[r][c],
    // and G2V.m[c][r] is row r of the matrix A applied to column c, i.e. A[r][c] = G2V.m[c][r] = Rt.m[r][c].
    const double a00 = Rt.m[0][0], a01 = Rt.m[0][1], a02 = Rt.m[0][2];
    const double a10 = Rt.m[1][0], a11 = Rt.m[1][1], a12 = Rt.m[1][2];
    const double a20 = Rt.m[2][0], a21 = Rt.m[2][1], a22 = Rt.m[2][2];
    {   // footprint conic (see the header comment): Sigma'_ij = sum_c a_ic a_jc l_c
        const double pxx = a00 * a00 * lx + a01 * a01 * ly + a02 * a02 * lz, pxy = a00 * a10 * lx + a01 * a11 * ly + a02 * a12 * lz;
        const double pyy = a10 * a10 * lx + a11 * a11 * ly + a12 * a12 * lz, pxz = a00 * a20 * lx + a01 * a21 * ly + a02 * a22 * lz;
        const double pyz = a10 * a20 * lx + a11 * a21 * ly + a12 * a22 * lz, pzz = a20 * a20 * lx + a21 * a21 * ly + a22 * a22 * lz;
        const double bx = pxx * mx + pxy * my + pxz * mz, by = pxy * mx + pyy * my + pyz * mz, bz = pxz * mx + pyz * my + pzz * mz;
        const double Kp = (mx * bx + my * by + mz * bz) - k;
        const double m00 = Kp * pxx - bx * bx, m01 = Kp * pxy - bx * by, m11 = Kp * pyy - by * by;
        const double m02 = Kp * pxz - bx * bz, m12 = Kp * pyz - by * bz, m22 = Kp * pzz - bz * bz;
        const double Sm = fabs(m00) + 2.0 * fabs(m01) + fabs(m11) + 2.0 * fabs(m02) + 2.0 * fabs(m12) + fabs(m22);
        if (Sm > 0.0 && Sm < 1e300) {
            const double is = 1.0 / Sm;
            fc[0] = make_float4((float)(m00 * is), (float)(m01 * is), (float)(m11 * is), (float)(m02 * is));
            if (FULL) {
                const double hx = 0.5 / (double)focal_x, hy = 0.5 / (double)focal_y;
                const double q = (m00 * hx * hx + m11 * hy * hy - 2.0 * fabs(m01) * hx * hy) * is;
                fc[1] = make_float4((float)(m12 * is), (float)(m22 * is), (float)q, -1e30f);
            } else {
                fc[1] = make_float4((float)(m12 * is), (float)(m22 * is), -1e30f, -1e30f);
            }
        }
        if (FULL) {
            const double zf = mz - sqrt(k / lmin) * (1.0 + 1e-5) - 1e-5 * fabs(mz);
            if (zf == zf) fc[1].w = (float)(zf - 1e-6 * fabs(zf));           // rounded down
        }
    }
    if (!FULL) return unbounded;
    const double Sxx = k * (a00 * a00 * ix + a01 * a01 * iy + a02 * a02 * iz);
    const double Syy = k * (a10 * a10 * ix + a11 * a11 * iy + a12 * a12 * iz);
    const double Szz = k * (a20 * a20 * ix + a21 * a21 * iy + a22 * a22 * iz);
    const double Sxz = k * (a00 * a20 * ix + a01 * a21 * iy + a02 * a22 * iz);
    const double Syz = k * (a10 * a20 * ix + a11 * a21 * iy + a12 * a22 * iz);
    const double czz = mz * mz - Szz;
    // camera inside (or the ellipsoid reaching the camera plane): unbounded
    if (!(czz > 1e-9 * mz * mz) || !(mz > 0.0)) return unbounded;
    const double Dx = Sxx * mz * mz - 2.0 * Sxz * mx * mz + Szz * mx * mx - (Sxx * Szz - Sxz * Sxz);
    const double Dy = Syy * mz * mz - 2.0 * Syz * my * mz + Szz * my * my - (Syy * Szz - Syz * Syz);
    if (!(Dx >= 0.0) || !(Dy >= 0.0)) return unbounded;
    const double sx = sqrt(Dx), sy = sqrt(Dy);
    const double cx = mx * mz - Sxz, cy = my * mz - Syz;
    const double iczz = 1.0 / czz;                     // (the edges are widened by 0.02 px below: an ulp of a product is nothing against that)
    const double u0 = (cx - sx) * iczz, u1 = (cx + sx) * iczz;
    const double v0 = (cy - sy) * iczz, v1 = (cy + sy) * iczz;
    // ray of pixel p: r = (p + 0.5 - W/2) / focal  ->  p = r * focal + W/2 - 0.5; widen by 0.02 px for the fp32 ray
    const double px0 = u0 * (double)focal_x + W / 2. - 0.5 - 0.02, px1 = u1 * (double)focal_x + W / 2. - 0.5 + 0.02;
    const double py0 = v0 * (double)focal_y + H / 2. - 0.5 - 0.02, py1 = v1 * (double)focal_y + H / 2. - 0.5 + 0.02;
    return make_float4((float)fmax(-1e9, ceil(px0)), (float)fmin(1e9, floor(px1)), (float)fmax(-1e9, ceil(py0)), (float)fmin(1e9, floor(py1)));
}

// SH rows of the workgroup's 256 Gaussians staged through the LDS (K1 and K9): one coefficient row per thread, 48 floats padded to 49
// words (conflict-free per-thread reads); the global reads are 16-byte loads over the block's contiguous rows instead of 48 dword
// loads per thread at a 192-byte stride (64 cache lines per load instruction: the address units, not the bytes, bounded K1 --
// 3.4 TB/s on its algorithmic bytes until round 4)
constexpr int K9_ROW = 49;

// The SH coefficients are read per thread from global memory (any M, unaligned tensors, the (_features_dc, _features_rest) pair of
// GofRasterArgs.shs_rest, or colors_precomp instead).  (Rows staged through the LDS as in preprocess_bwd: measured slower in round 5,
// removed in round 6 -- profiles/r05_ab_call4_preprocess_fwd.txt.)
// STAGE (round 5): 0 = the whole kernel; 1 = the culls and what binning needs of a Gaussian (radii, tiles_touched, tile rectangle,
// depth key) and nothing else; 2 = everything else (record, conics, footprint, depth, clamp flags), the culls recomputed.  Nothing of
// the binning chain -- depth sort, scan, instance emission, tile sort, ranges, tile order: 0.34 ms at S1M, 0.9 ms at 6M Gaussians --
// reads what stage 2 writes (the blend does), and stage 2 is bound by its fp64 arithmetic at 4 waves per SIMD: the sync-free forward
// queues stage 1 (a third of the instructions), then stage 2 on a second stream BESIDE the binning chain, and joins in front of the
// blend (api.hip: forward_stage1).  Both stages evaluate the culls with the same instructions: same bits.
// FOOT: 1 = the full footprint (pixel box, q, zfront: the opacity-field query, tight tile rectangles), 0 = the conic alone (a forward
// that is followed by the blend only: footprint_bbox<false>)
template <int STAGE, int FOOT>
__device__ __forceinline__ void
preprocess_one(const int idx, int P, int D, int M,
               const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
               const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
               const float* __restrict__ shs_rest,
               const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
               const float* __restrict__ v2g_precomp, Cam cam,
               int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, float kernel_size,
               uint32_t gx, uint32_t gy, int mode_bits /* bit 0: prefiltered; bit 1: tight tile rectangles */,
               int32_t* __restrict__ radii, float* __restrict__ depths, SplatRec* __restrict__ rec,
               float4* __restrict__ conic_out, float4* __restrict__ bbox_out, float4* __restrict__ fconic_out, uint32_t* __restrict__ tiles_touched,
               uint2* __restrict__ rect_out, uint8_t* __restrict__ clamped,
               uint32_t* __restrict__ depth_key, uint32_t* __restrict__ depth_val, uint32_t* __restrict__ flags)
{
    const bool live = idx < P;
    int32_t my_radii = 0;
    uint32_t my_tiles = 0;
    uint2 my_rect = make_uint2(0u, 0u);

    // ---- part 1: the culls (near plane, degenerate 2D covariance, empty tile rectangle) and everything they need ----
    bool vis = false;
    V3 p_orig = { 0, 0, 0 }, p_view = { 0, 0, 1 }, scale = { 0, 0, 0 };
    float4 rot = { 0, 0, 0, 0 };
    float coef = 0, conx = 0, cony = 0, conz = 0, my_radius = 0, pix = 0, piy = 0;
    uint32_t minx = 0, miny = 0, maxx = 0, maxy = 0;
    if (live) do {
        p_orig = V3{ means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        p_view = transform_point_4x3(p_orig, cam.view);
        // near cull only (auxiliary.h:189): the lateral frustum test is commented out in the reference
        if (p_view.z <= 0.2f) {
            if (STAGE != 2 && (mode_bits & 1)) atomicOr(&flags[0], 1u);
            break;
        }
        const float* pm = cam.proj;
        const float hx = pm[0] * p_orig.x + pm[4] * p_orig.y + pm[8] * p_orig.z + pm[12];
        const float hy = pm[1] * p_orig.x + pm[5] * p_orig.y + pm[9] * p_orig.z + pm[13];
        const float hw = pm[3] * p_orig.x + pm[7] * p_orig.y + pm[11] * p_orig.z + pm[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;

        if (scales) scale = V3{ scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2] };
        if (rotations) rot = reinterpret_cast<const float4*>(rotations)[idx];

        // 3D covariance (forward.cu:129-163)
        float c3[6];
        if (cov3D_precomp) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * (size_t)idx + i];
        } else {
            M3 S = mk3(1, 0, 0, 0, 1, 0, 0, 0, 1);
            S.m[0][0] = scale_modifier * scale.x;
            S.m[1][1] = scale_modifier * scale.y;
            S.m[2][2] = scale_modifier * scale.z;
            const M3 R = quat_to_R(rot.x, rot.y, rot.z, rot.w);
            const M3 Mm = mul(S, R);
            const M3 Sigma = mul(transpose(Mm), Mm);
            c3[0] = Sigma.m[0][0]; c3[1] = Sigma.m[0][1]; c3[2] = Sigma.m[0][2];
            c3[3] = Sigma.m[1][1]; c3[4] = Sigma.m[1][2]; c3[5] = Sigma.m[2][2];
        }

        // 2D covariance + mip low-pass coefficient (forward.cu:74-124)
        V3 t = p_view;
        const float limx = 1.3f * tan_fovx;
        const float limy = 1.3f * tan_fovy;
        const float txtz = t.x / t.z;
        const float tytz = t.y / t.z;
        t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
        t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
        const M3 J = mk3(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z),
                         0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z),
                         0, 0, 0);
        const float* vm = cam.view;
        const M3 Wm = mk3(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
        const M3 T = mul(Wm, J);
        const M3 Vrk = mk3(c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]);
        M3 cov = mul(mul(transpose(T), transpose(Vrk)), T);
        const float det_0 = (float)fmax(1e-6, (double)(cov.m[0][0] * cov.m[1][1] - cov.m[0][1] * cov.m[0][1]));
        const float det_1 = (float)fmax(1e-6, (double)((cov.m[0][0] + kernel_size) * (cov.m[1][1] + kernel_size) - cov.m[0][1] * cov.m[0][1]));
        coef = (float)sqrt((double)det_0 / ((double)det_1 + 1e-6) + 1e-6);
        if ((double)det_0 <= 1e-6 || (double)det_1 <= 1e-6) coef = 0.0f;
        const float covx = cov.m[0][0] + kernel_size, covy = cov.m[0][1], covz = cov.m[1][1] + kernel_size;

        const float det = (covx * covz - covy * covy);
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        conx = covz * det_inv; cony = -covy * det_inv; conz = covx * det_inv;

        const float mid = 0.5f * (covx + covz);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        pix = (float)((((double)projx + 1.0) * W - 1.0) * 0.5);
        piy = (float)((((double)projy + 1.0) * H - 1.0) * 0.5);
        get_rect(pix, piy, (int)my_radius, minx, miny, maxx, maxy, gx, gy);
        if ((maxx - minx) * (maxy - miny) == 0) break;
        vis = true;
        my_radii = (int32_t)my_radius;
        my_tiles = (maxy - miny) * (maxx - minx);
        my_rect = make_uint2(minx | (miny << 16), (maxx - minx) | ((maxy - miny) << 16));
    } while (0);
    if (STAGE == 1) {          // (never with tight tile rectangles: those need stage 2's footprint box)
        if (!live) return;
        radii[idx] = my_radii;
        tiles_touched[idx] = my_tiles;
        rect_out[idx] = my_rect;
        depth_key[idx] = my_radii > 0 ? __float_as_uint(p_view.z) : 0xFFFFFFFFu;
        depth_val[idx] = (uint32_t)idx;
        return;
    }

    // ---- part 2: colour, view2gaussian, footprint, the record ----
    if (vis) {
        SplatRec r;
        uint32_t cb = 0;
        if (colors_precomp == nullptr) {
            const V3 campos = { cam.campos[0], cam.campos[1], cam.campos[2] };
            const float* sh0 = shs_rest ? shs + (size_t)idx * 3 : shs + (size_t)idx * M * 3;
            const float* shp = shs_rest ? shs_rest + (size_t)idx * (M - 1) * 3 - 3 : sh0;
            const V3 rgb = sh_to_rgb(D, p_orig, campos, sh0, shp, cb);
            r.f[REC_RGB] = rgb.x; r.f[REC_RGB + 1] = rgb.y; r.f[REC_RGB + 2] = rgb.z;
        } else {
            r.f[REC_RGB] = colors_precomp[3 * (size_t)idx]; r.f[REC_RGB + 1] = colors_precomp[3 * (size_t)idx + 1]; r.f[REC_RGB + 2] = colors_precomp[3 * (size_t)idx + 2];
        }
        clamped[idx] = (uint8_t)cb;

        float4 box = make_float4(-1e30f, 1e30f, -1e30f, 1e30f);     // unbounded unless proven otherwise
        float4 fc[2] = { make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, -1e30f) };     // (all-zero conic: never culls, whatever q)
        if (v2g_precomp == nullptr) {
            V2GInter I;
            v2g_intermediates(scale, p_orig, rot, cam.view, I);
            const V3 t2 = I.t2;
            const double C = (double)(t2.x * t2.x) * I.Sx + (double)(t2.y * t2.y) * I.Sy + (double)(t2.z * t2.z) * I.Sz;
            box = footprint_bbox<FOOT != 0>(I, p_view, opacities[idx] * coef, focal_x, focal_y, W, H, fc,      // (s_c^2 + 1e-7: the doubles v2g_intermediates divides by)
                                 (double)scale.x * scale.x + 1e-7, (double)scale.y * scale.y + 1e-7, (double)scale.z * scale.z + 1e-7);
            const V3 B = mul(t2, I.SR);
            const M3 Sigma = mul(transpose(I.Rt), I.SR);
            r.f[0] = Sigma.m[0][0]; r.f[1] = Sigma.m[0][1]; r.f[2] = Sigma.m[0][2];
            r.f[3] = Sigma.m[1][1]; r.f[4] = Sigma.m[1][2]; r.f[5] = Sigma.m[2][2];
            r.f[6] = B.x; r.f[7] = B.y; r.f[8] = B.z; r.f[9] = (float)C;
        } else {
#pragma unroll
            for (int i = 0; i < 10; i++) r.f[i] = v2g_precomp[10 * (size_t)idx + i];
        }
        r.f[REC_W] = opacities[idx] * coef;
        r.f[REC_XY] = pix; r.f[REC_XY + 1] = piy;

        float4* dst = reinterpret_cast<float4*>(&rec[idx]);
        dst[0] = make_float4(r.f[0], r.f[1], r.f[2], r.f[3]);
        dst[1] = make_float4(r.f[4], r.f[5], r.f[6], r.f[7]);
        dst[2] = make_float4(r.f[8], r.f[9], r.f[10], r.f[11]);
        dst[3] = make_float4(r.f[12], r.f[13], r.f[14], r.f[15]);
        conic_out[idx] = make_float4(conx, cony, conz, r.f[REC_W]);      // .w: the blend weight again, where gather_tile_partials finds it beside nothing else it reads (blend_backward.hip)
        if (FOOT && bbox_out) bbox_out[idx] = box;      // (only the query reads it: stored into a full-size geometry workspace, api.hip)
        fconic_out[2 * (size_t)idx] = fc[0];
        fconic_out[2 * (size_t)idx + 1] = fc[1];
        depths[idx] = p_view.z;
        // Opt-in (gof_set_tight_tile_rects(1) / GOF_TIGHT_RECTS=1; the default keeps the reference's tile lists entry for entry): the
        // reference bins a Gaussian into every tile of the square of its 3-sigma radius (auxiliary.h:64-74); a tile none of whose pixels
        // lies inside the footprint box (the conservative pixel box of the alpha >= 1/255 region incl. the error allowance,
        // footprint_bbox above) cannot receive a contribution from it.  Intersecting the two leaves image and gradients unchanged and
        // shortens the lists (R -21 % at S1M) -- emission, tile sort, the forward's cull scan and the backward's staging shrink with
        // them (measured: -2.3 % of the S1M step, -5 % clustered); radii (returned to the caller) stay the reference's.
        if ((mode_bits & 2) && box.x > -1e29f) {                      // (unbounded boxes carry -1e30 / 1e30: no statement)
            // (widened by one pixel: the opacity-field query's corner sub-rays sit half a pixel outside the pixel centres, integrate.hip)
            const float bx0 = box.x - 1.0f, bx1 = box.y + 1.0f, by0 = box.z - 1.0f, by1 = box.w + 1.0f;
            const int tx0 = (int)floorf(fmaxf(bx0, 0.0f) * (1.0f / TILE_X)), tx1 = (bx1 < 0.0f) ? 0 : (int)floorf(fminf(bx1, 1e9f) * (1.0f / TILE_X)) + 1;
            const int ty0 = (int)floorf(fmaxf(by0, 0.0f) * (1.0f / TILE_Y)), ty1 = (by1 < 0.0f) ? 0 : (int)floorf(fminf(by1, 1e9f) * (1.0f / TILE_Y)) + 1;
            const bool none = (bx0 > bx1) | (by0 > by1);        // (a box between two pixel centres is still reachable by a corner sub-ray)
            minx = max(minx, (uint32_t)min(tx0, (int)gx)); maxx = min(maxx, (uint32_t)min(max(tx1, 0), (int)gx));
            miny = max(miny, (uint32_t)min(ty0, (int)gy)); maxy = min(maxy, (uint32_t)min(max(ty1, 0), (int)gy));
            if (none || maxx <= minx || maxy <= miny) { maxx = minx; maxy = miny; }
            my_tiles = (maxy - miny) * (maxx - minx);
            my_rect = make_uint2(minx | (miny << 16), (maxx - minx) | ((maxy - miny) << 16));
        }
    }
    if (!live || STAGE == 2) return;
    radii[idx] = my_radii;
    tiles_touched[idx] = my_tiles;
    rect_out[idx] = my_rect;
    // sort key of the binning stage: positive floats order like their bit patterns; culled Gaussians go last
    depth_key[idx] = my_radii > 0 ? __float_as_uint(p_view.z) : 0xFFFFFFFFu;
    depth_val[idx] = (uint32_t)idx;
}

// The kernel: one Gaussian per thread, grid-stride.  Stages 0 and 1 are launched with a workgroup per 256 Gaussians.  Stage 2 runs on
// the library's second stream BESIDE the binning chain (api.hip: forward_stage1), whose launches are chains of dependent steps in a
// few hundred workgroups -- a depth-sort pass of 1 M pairs is 245 of them -- and a stage 2 that fills every wave slot of the device
// makes those wait for a slot: measured (round 6, profiles/r06_ab_call1_*.txt) the depth sort ran 0.108 ms alone, 0.147 beside round
// 5's stage 2 and 0.177 ms beside the lighter stage 2 of this round, which occupies MORE slots.  Stage 2 is therefore launched with a
// BOUNDED grid (api.hip: a few workgroups per CU) and strides over the Gaussians: it has until the blend to finish.
template <int STAGE, int FOOT>
__global__ void __launch_bounds__(256)
preprocess_fwd(int P, int D, int M,
               const float* __restrict__ means3D, const float* __restrict__ scales, float scale_modifier,
               const float* __restrict__ rotations, const float* __restrict__ opacities, const float* __restrict__ shs,
               const float* __restrict__ shs_rest,
               const float* __restrict__ cov3D_precomp, const float* __restrict__ colors_precomp,
               const float* __restrict__ v2g_precomp, Cam cam,
               int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y, float kernel_size,
               uint32_t gx, uint32_t gy, int mode_bits /* bit 0: prefiltered; bit 1: tight tile rectangles */,
               int32_t* __restrict__ radii, float* __restrict__ depths, SplatRec* __restrict__ rec,
               float4* __restrict__ conic_out, float4* __restrict__ bbox_out, float4* __restrict__ fconic_out, uint32_t* __restrict__ tiles_touched,
               uint2* __restrict__ rect_out, uint8_t* __restrict__ clamped,
               uint32_t* __restrict__ depth_key, uint32_t* __restrict__ depth_val, uint32_t* __restrict__ flags,
               uint32_t* __restrict__ zero_ptr, uint32_t zero_n)
{
    // zero_n words at zero_ptr are cleared on the way (the first kernel of a frame: the scratch of the depth sort's single-kernel passes and
    // the state of the fused gather + scan behind it must be zero before those launches -- a memset launch of its own until round 5);
    // never by stage 2, which runs BESIDE them
    if (STAGE != 2 && zero_n)
        for (uint32_t w = blockIdx.x * 256u + threadIdx.x; w < zero_n; w += gridDim.x * 256u) zero_ptr[w] = 0u;
    for (int base = (int)blockIdx.x * 256; base < P; base += (int)gridDim.x * 256)       // (one trip unless the grid is bounded: stage 2)
        preprocess_one<STAGE, FOOT>(base + (int)threadIdx.x, P, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, shs_rest, cov3D_precomp,
                                    colors_precomp, v2g_precomp, cam, W, H, tan_fovx, tan_fovy, focal_x, focal_y, kernel_size, gx, gy, mode_bits, radii,
                                    depths, rec, conic_out, bbox_out, fconic_out, tiles_touched, rect_out, clamped, depth_key, depth_val, flags);
}
template __global__ void preprocess_fwd<0, 0>(int, int, int, const float*, const float*, float, const float*, const float*, const float*, const float*, const float*,
                                              const float*, const float*, Cam, int, int, float, float, float, float, float, uint32_t, uint32_t, int, int32_t*, float*, SplatRec*,
                                              float4*, float4*, float4*, uint32_t*, uint2*, uint8_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t);
template __global__ void preprocess_fwd<0, 1>(int, int, int, const float*, const float*, float, const float*, const float*, const float*, const float*, const float*,
                                              const float*, const float*, Cam, int, int, float, float, float, float, float, uint32_t, uint32_t, int, int32_t*, float*, SplatRec*,
                                              float4*, float4*, float4*, uint32_t*, uint2*, uint8_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t);
template __global__ void preprocess_fwd<1, 0>(int, int, int, const float*, const float*, float, const float*, const float*, const float*, const float*, const float*,
                                              const float*, const float*, Cam, int, int, float, float, float, float, float, uint32_t, uint32_t, int, int32_t*, float*, SplatRec*,
                                              float4*, float4*, float4*, uint32_t*, uint2*, uint8_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t);
template __global__ void preprocess_fwd<2, 0>(int, int, int, const float*, const float*, float, const float*, const float*, const float*, const float*, const float*,
                                              const float*, const float*, Cam, int, int, float, float, float, float, float, uint32_t, uint32_t, int, int32_t*, float*, SplatRec*,
                                              float4*, float4*, float4*, uint32_t*, uint2*, uint8_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t);
template __global__ void preprocess_fwd<2, 1>(int, int, int, const float*, const float*, float, const float*, const float*, const float*, const float*, const float*,
                                              const float*, const float*, Cam, int, int, float, float, float, float, float, uint32_t, uint32_t, int, int32_t*, float*, SplatRec*,
                                              float4*, float4*, float4*, uint32_t*, uint2*, uint8_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t);

// ---------------------------------------------------------------------------------------------------
// K9: backward of the per-Gaussian stage (backward.cu:593-631): view2gaussian backward
// (overwrites dL_dmeans / dL_dscales / dL_drots, backward.cu:494-497, 570-573, 585-586) and SH
// backward (adds to dL_dmeans, backward.cu:138).  Skips radii <= 0.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sh_backward(int deg, V3 pos, V3 campos, const float* __restrict__ shp, uint32_t clamp_bits,
                                            V3 dL_dRGB, float* __restrict__ dL_dshp, int M, V3& dL_dmean_add)
{
    const V3 dir_orig = pos - campos;
    const V3 dir = dir_orig / sqrtf(dot3(dir_orig, dir_orig));
    const V3* sh = reinterpret_cast<const V3*>(shp);
    dL_dRGB.x *= (clamp_bits & 1u) ? 0 : 1;
    dL_dRGB.y *= (clamp_bits & 2u) ? 0 : 1;
    dL_dRGB.z *= (clamp_bits & 4u) ? 0 : 1;
    V3 dRGBdx = { 0, 0, 0 }, dRGBdy = { 0, 0, 0 }, dRGBdz = { 0, 0, 0 };
    const float x = dir.x, y = dir.y, z = dir.z;
    V3* dL_dsh = reinterpret_cast<V3*>(dL_dshp);

    dL_dsh[0] = SH_C0 * dL_dRGB;
    if (deg > 0) {
        const float dRGBdsh1 = -SH_C1 * y;
        const float dRGBdsh2 = SH_C1 * z;
        const float dRGBdsh3 = -SH_C1 * x;
        dL_dsh[1] = dRGBdsh1 * dL_dRGB;
        dL_dsh[2] = dRGBdsh2 * dL_dRGB;
        dL_dsh[3] = dRGBdsh3 * dL_dRGB;
        dRGBdx = -SH_C1 * sh[3];
        dRGBdy = -SH_C1 * sh[1];
        dRGBdz = SH_C1 * sh[2];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            const float dRGBdsh4 = SH_C2[0] * xy;
            const float dRGBdsh5 = SH_C2[1] * yz;
            const float dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
            const float dRGBdsh7 = SH_C2[3] * xz;
            const float dRGBdsh8 = SH_C2[4] * (xx - yy);
            dL_dsh[4] = dRGBdsh4 * dL_dRGB;
            dL_dsh[5] = dRGBdsh5 * dL_dRGB;
            dL_dsh[6] = dRGBdsh6 * dL_dRGB;
            dL_dsh[7] = dRGBdsh7 * dL_dRGB;
            dL_dsh[8] = dRGBdsh8 * dL_dRGB;
            dRGBdx = dRGBdx + (SH_C2[0] * y * sh[4] + SH_C2[2] * 2.f * -x * sh[6] + SH_C2[3] * z * sh[7] + SH_C2[4] * 2.f * x * sh[8]);
            dRGBdy = dRGBdy + (SH_C2[0] * x * sh[4] + SH_C2[1] * z * sh[5] + SH_C2[2] * 2.f * -y * sh[6] + SH_C2[4] * 2.f * -y * sh[8]);
            dRGBdz = dRGBdz + (SH_C2[1] * y * sh[5] + SH_C2[2] * 2.f * 2.f * z * sh[6] + SH_C2[3] * x * sh[7]);
            if (deg > 2) {
                const float dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy);
                const float dRGBdsh10 = SH_C3[1] * xy * z;
                const float dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                const float dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                const float dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
                const float dRGBdsh14 = SH_C3[5] * z * (xx - yy);
                const float dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
                dL_dsh[9] = dRGBdsh9 * dL_dRGB;
                dL_dsh[10] = dRGBdsh10 * dL_dRGB;
                dL_dsh[11] = dRGBdsh11 * dL_dRGB;
                dL_dsh[12] = dRGBdsh12 * dL_dRGB;
                dL_dsh[13] = dRGBdsh13 * dL_dRGB;
                dL_dsh[14] = dRGBdsh14 * dL_dRGB;
                dL_dsh[15] = dRGBdsh15 * dL_dRGB;
                dRGBdx = dRGBdx + (
                    SH_C3[0] * sh[9] * 3.f * 2.f * xy +
                    SH_C3[1] * sh[10] * yz +
                    SH_C3[2] * sh[11] * -2.f * xy +
                    SH_C3[3] * sh[12] * -3.f * 2.f * xz +
                    SH_C3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) +
                    SH_C3[5] * sh[14] * 2.f * xz +
                    SH_C3[6] * sh[15] * 3.f * (xx - yy));
                dRGBdy = dRGBdy + (
                    SH_C3[0] * sh[9] * 3.f * (xx - yy) +
                    SH_C3[1] * sh[10] * xz +
                    SH_C3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) +
                    SH_C3[3] * sh[12] * -3.f * 2.f * yz +
                    SH_C3[4] * sh[13] * -2.f * xy +
                    SH_C3[5] * sh[14] * -2.f * yz +
                    SH_C3[6] * sh[15] * -3.f * 2.f * xy);
                dRGBdz = dRGBdz + (
                    SH_C3[1] * sh[10] * xy +
                    SH_C3[2] * sh[11] * 4.f * 2.f * yz +
                    SH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) +
                    SH_C3[4] * sh[13] * 4.f * 2.f * xz +
                    SH_C3[5] * sh[14] * (xx - yy));
            }
        }
    }
    (void)M;
    const V3 dv = { dot3(dRGBdx, dL_dRGB), dot3(dRGBdy, dL_dRGB), dot3(dRGBdz, dL_dRGB) };
    const V3 v = dir_orig;
    // auxiliary.h:125-135
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dL_dmean_add.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    dL_dmean_add.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    dL_dmean_add.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
}

// TILED (M == 16, 16-byte aligned SH arrays): the 192-byte SH row of a Gaussian is read and its gradient row written with a
// 192-byte stride between lanes, i.e. every 16-byte access of a wave touches 64 different 64-byte sectors.  The workgroup instead
// moves its 256 x 192 B tile through LDS with fully coalesced 16-byte accesses (rows padded to 49 floats: conflict-free), each
// thread works on its own LDS row, and culled Gaussians / coefficients above the active degree leave zeros in the tile -- so the
// host skips the 192 B/Gaussian memset of dL_dsh as well.
// MODE 0: rows read / written in place; 1: TILED; 2: TILED with DC and higher bands in separate tensors (shs = [P,1,3],
// shs_rest = [P,15,3] and the same for the gradient): the LDS row is columns 0-2 | 3-47 of the two tiles.
template <int MODE>
__global__ void __launch_bounds__(256)
preprocess_bwd(int P, int D, int M,
               const float* __restrict__ means3D, const int32_t* __restrict__ radii, const float* __restrict__ shs,
               const float* __restrict__ shs_rest,
               const uint8_t* __restrict__ clamped, const float* __restrict__ scales, const float* __restrict__ rotations,
               Cam cam, const float* __restrict__ dL_dv2g, const float* __restrict__ dL_dcolor,
               float* __restrict__ dL_dmeans, float* __restrict__ dL_dsh, float* __restrict__ dL_dsh_rest, float* __restrict__ dL_dscales,
               float* __restrict__ dL_drots)
{
    constexpr bool TILED = MODE != 0;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    __shared__ float s_sh[TILED ? 256 * K9_ROW : 1];
    const int b0 = blockIdx.x * 256;
    const int rows = min(256, P - b0);
    if (MODE == 1) {
        const float4* src = reinterpret_cast<const float4*>(shs + (size_t)b0 * 48);
        for (int i = threadIdx.x; i < rows * 12; i += 256) {
            const float4 v = src[i];
            const int f = i * 4, g = f / 48, k = f - g * 48;
            float* d = &s_sh[g * K9_ROW + k];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
    }
    if (MODE == 2) {
        const float* dc = shs + (size_t)b0 * 3;
        for (int i = threadIdx.x; i < rows * 3; i += 256) {
            const int g = i / 3, k = i - g * 3;
            s_sh[g * K9_ROW + k] = dc[i];
        }
        const float* rest = shs_rest + (size_t)b0 * 45;
        const int n4 = (reinterpret_cast<uintptr_t>(rest) & 15) ? 0 : (rows * 45) >> 2;       // 16-byte loads when the tensor allows
        const float4* src = reinterpret_cast<const float4*>(rest);
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float4 v = src[i];
            const float q[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int f = i * 4 + e, g = f / 45, k = f - g * 45;
                s_sh[g * K9_ROW + 3 + k] = q[e];
            }
        }
        for (int f = n4 * 4 + threadIdx.x; f < rows * 45; f += 256) {
            const int g = f / 45, k = f - g * 45;
            s_sh[g * K9_ROW + 3 + k] = rest[f];
        }
        __syncthreads();
    }
    const bool visible = idx < P && radii[idx] > 0;
    if (idx < P && !visible) {
        // culled Gaussian: the reference returns here and relies on the binding's torch::zeros (rasterize_points.cu:161-170); this
        // kernel owns dL_dmeans / dL_dscales / dL_drots completely, so the host does not memset them
        dL_dmeans[3 * idx] = 0.0f; dL_dmeans[3 * idx + 1] = 0.0f; dL_dmeans[3 * idx + 2] = 0.0f;
        dL_dscales[3 * idx] = 0.0f; dL_dscales[3 * idx + 1] = 0.0f; dL_dscales[3 * idx + 2] = 0.0f;
        reinterpret_cast<float4*>(dL_drots)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!TILED && !visible) return;
    float* const my_row = &s_sh[TILED ? threadIdx.x * K9_ROW : 0];
    if (visible) {
    const V3 mean = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
    const V3 scale = { scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2] };
    const float4 rot = reinterpret_cast<const float4*>(rotations)[idx];
    const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;

    V2GInter I;
    v2g_intermediates(scale, mean, rot, cam.view, I);
    const M3& Rt = I.Rt;
    const M3& SR = I.SR;
    const V3 t2 = I.t2;
    const M4& G2V = I.G2V;
    const double Sx = I.Sx, Sy = I.Sy, Sz = I.Sz;
    float d[10];
#pragma unroll
    for (int i = 0; i < 10; i++) d[i] = dL_dv2g[10 * (size_t)idx + i];

    const M3 dL_dSigma = mk3(d[0], 0.5f * d[1], 0.5f * d[2],
                             0.5f * d[1], d[3], 0.5f * d[4],
                             0.5f * d[2], 0.5f * d[4], d[5]);
    const V3 dL_dB = { d[6], d[7], d[8] };
    const float dL_dC = d[9];

    const M3 q = add(mul(Rt, dL_dSigma), outer(t2, dL_dB));                 // dL_dS_inv_square_R
    M3 dL_dRt = transpose(mul(dL_dSigma, transpose(SR)));
    dL_dRt = add(dL_dRt, mk3((float)(Sx * q.m[0][0]), (float)(Sy * q.m[0][1]), (float)(Sz * q.m[0][2]),
                             (float)(Sx * q.m[1][0]), (float)(Sy * q.m[1][1]), (float)(Sz * q.m[1][2]),
                             (float)(Sx * q.m[2][0]), (float)(Sy * q.m[2][1]), (float)(Sz * q.m[2][2])));
    float dSx = q.m[0][0] * Rt.m[0][0] + q.m[1][0] * Rt.m[1][0] + q.m[2][0] * Rt.m[2][0];
    float dSy = q.m[0][1] * Rt.m[0][1] + q.m[1][1] * Rt.m[1][1] + q.m[2][1] * Rt.m[2][1];
    float dSz = q.m[0][2] * Rt.m[0][2] + q.m[1][2] * Rt.m[1][2] + q.m[2][2] * Rt.m[2][2];
    const float dtx = (float)(2 * t2.x * Sx * dL_dC + dL_dB.x * SR.m[0][0] + dL_dB.y * SR.m[1][0] + dL_dB.z * SR.m[2][0]);
    const float dty = (float)(2 * t2.y * Sy * dL_dC + dL_dB.x * SR.m[0][1] + dL_dB.y * SR.m[1][1] + dL_dB.z * SR.m[2][1]);
    const float dtz = (float)(2 * t2.z * Sz * dL_dC + dL_dB.x * SR.m[0][2] + dL_dB.y * SR.m[1][2] + dL_dB.z * SR.m[2][2]);
    dSx += dL_dC * t2.x * t2.x;
    dSy += dL_dC * t2.y * t2.y;
    dSz += dL_dC * t2.z * t2.z;
    dL_dscales[3 * idx + 0] = (float)(-2 / scale.x * Sx * dSx);
    dL_dscales[3 * idx + 1] = (float)(-2 / scale.y * Sy * dSy);
    dL_dscales[3 * idx + 2] = (float)(-2 / scale.z * Sz * dSz);

    const M3 G2V_R_t = mk3(G2V.m[0][0], G2V.m[1][0], G2V.m[2][0],
                           G2V.m[0][1], G2V.m[1][1], G2V.m[2][1],
                           G2V.m[0][2], G2V.m[1][2], G2V.m[2][2]);
    const V3 G2V_t = { G2V.m[3][0], G2V.m[3][1], G2V.m[3][2] };
    const V3 dV = { dtx, dty, dtz };
    const M3 dL_dV2G_R_t = transpose(dL_dRt);
    const M3 from_t = mk3(-dV.x * G2V_t.x, -dV.x * G2V_t.y, -dV.x * G2V_t.z,
                          -dV.y * G2V_t.x, -dV.y * G2V_t.y, -dV.y * G2V_t.z,
                          -dV.z * G2V_t.x, -dV.z * G2V_t.y, -dV.z * G2V_t.z);
    const M3 dL_dG2V_R = add(dL_dV2G_R_t, from_t);
    const V3 dL_dG2V_t = mul(-dV, G2V_R_t);
    M4 dL_dG2V;
#pragma unroll
    for (int c = 0; c < 3; c++) { dL_dG2V.m[c][0] = dL_dG2V_R.m[c][0]; dL_dG2V.m[c][1] = dL_dG2V_R.m[c][1]; dL_dG2V.m[c][2] = dL_dG2V_R.m[c][2]; dL_dG2V.m[c][3] = 0.0f; }
    dL_dG2V.m[3][0] = dL_dG2V_t.x; dL_dG2V.m[3][1] = dL_dG2V_t.y; dL_dG2V.m[3][2] = dL_dG2V_t.z; dL_dG2V.m[3][3] = 0.0f;
    const M4 dL_dG2W = mul(transpose(I.W2V), dL_dG2V);
    const float (*Mt)[4] = dL_dG2W.m;   // dL_dMt[c][r] = dL_dG2W[c][r] for c,r < 3

    V3 dmean = { dL_dG2W.m[3][0], dL_dG2W.m[3][1], dL_dG2W.m[3][2] };

    float4 dq;
    dq.x = 2 * z * (Mt[0][1] - Mt[1][0]) + 2 * y * (Mt[2][0] - Mt[0][2]) + 2 * x * (Mt[1][2] - Mt[2][1]);
    dq.y = 2 * y * (Mt[1][0] + Mt[0][1]) + 2 * z * (Mt[2][0] + Mt[0][2]) + 2 * r * (Mt[1][2] - Mt[2][1]) - 4 * x * (Mt[2][2] + Mt[1][1]);
    dq.z = 2 * x * (Mt[1][0] + Mt[0][1]) + 2 * r * (Mt[2][0] - Mt[0][2]) + 2 * z * (Mt[1][2] + Mt[2][1]) - 4 * y * (Mt[2][2] + Mt[0][0]);
    dq.w = 2 * r * (Mt[0][1] - Mt[1][0]) + 2 * x * (Mt[2][0] + Mt[0][2]) + 2 * y * (Mt[1][2] + Mt[2][1]) - 4 * z * (Mt[1][1] + Mt[0][0]);
    reinterpret_cast<float4*>(dL_drots)[idx] = dq;

    if (shs) {
        const V3 campos = { cam.campos[0], cam.campos[1], cam.campos[2] };
        const V3 dL_dRGB = { dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2] };
        V3 addm;
        if (TILED) {
            float shr[48];                              // this Gaussian's coefficients: LDS row -> registers (static indices)
            const int nco = 3 * (D + 1) * (D + 1);
#pragma unroll
            for (int k = 0; k < 48; k++) shr[k] = (k < nco) ? my_row[k] : 0.0f;
            sh_backward(D, mean, campos, shr, clamped[idx], dL_dRGB, my_row, M, addm);
            for (int k = nco; k < 48; k++) my_row[k] = 0.0f;                 // coefficients above the active degree
        } else {
            sh_backward(D, mean, campos, shs + (size_t)idx * M * 3, clamped[idx], dL_dRGB, dL_dsh + (size_t)idx * M * 3, M, addm);
        }
        dmean = dmean + addm;
    }
    dL_dmeans[3 * idx + 0] = dmean.x;
    dL_dmeans[3 * idx + 1] = dmean.y;
    dL_dmeans[3 * idx + 2] = dmean.z;
    } else if (TILED && idx < P) {
#pragma unroll
        for (int k = 0; k < 48; k++) my_row[k] = 0.0f;                       // culled Gaussian: zero gradient row
    }
    if (MODE == 1) {
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(dL_dsh + (size_t)b0 * 48);
        for (int i = threadIdx.x; i < rows * 12; i += 256) {
            const int f = i * 4, g = f / 48, k = f - g * 48;
            const float* r = &s_sh[g * K9_ROW + k];
            dst[i] = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
    if (MODE == 2) {
        __syncthreads();
        float* dc = dL_dsh + (size_t)b0 * 3;
        for (int i = threadIdx.x; i < rows * 3; i += 256) {
            const int g = i / 3, k = i - g * 3;
            dc[i] = s_sh[g * K9_ROW + k];
        }
        float* rest = dL_dsh_rest + (size_t)b0 * 45;
        const int n4 = (reinterpret_cast<uintptr_t>(rest) & 15) ? 0 : (rows * 45) >> 2;
        float4* dst = reinterpret_cast<float4*>(rest);
        for (int i = threadIdx.x; i < n4; i += 256) {
            float q[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int f = i * 4 + e, g = f / 45, k = f - g * 45;
                q[e] = s_sh[g * K9_ROW + 3 + k];
            }
            dst[i] = make_float4(q[0], q[1], q[2], q[3]);
        }
        for (int f = n4 * 4 + threadIdx.x; f < rows * 45; f += 256) {
            const int g = f / 45, k = f - g * 45;
            rest[f] = s_sh[g * K9_ROW + 3 + k];
        }
    }
}
#define GOF_K9_INST(MODE) template __global__ void preprocess_bwd<MODE>(int, int, int, const float*, const int32_t*, const float*, const float*, \
    const uint8_t*, const float*, const float*, Cam, const float*, const float*, float*, float*, float*, float*, float*);
GOF_K9_INST(0) GOF_K9_INST(1) GOF_K9_INST(2)
#undef GOF_K9_INST

// ---------------------------------------------------------------------------------------------------
// K10: query points (forward.cu:722-766)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
preprocess_points(int PN, const float* __restrict__ points3D, Cam cam, int W, int H, float focal_x, float focal_y,
                  float4* __restrict__ pos, uint32_t* __restrict__ tiles_touched)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= PN) return;
    uint32_t touched = 0;
    const V3 p = { points3D[3 * idx], points3D[3 * idx + 1], points3D[3 * idx + 2] };
    const V3 pv = transform_point_4x3(p, cam.view);
    if (!(pv.z <= 0.2f)) {
        const float pix = (float)((double)(focal_x * pv.x / (pv.z + 0.0000001f)) + W / 2.);
        const float piy = (float)((double)(focal_y * pv.y / (pv.z + 0.0000001f)) + H / 2.);
        if (!(pix < 0 || pix >= W || piy < 0 || piy >= H)) {
            pos[idx] = make_float4(pix, piy, pv.z, 0.0f);       // projected position + depth (forward.cu:758-760) as one 16-byte line
            touched = 1;
        }
    }
    tiles_touched[idx] = touched;
}

// K15 (rasterizer_impl.cu:54-66)
__global__ void __launch_bounds__(256)
mark_visible_kernel(int P, const float* __restrict__ means3D, Cam cam, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const V3 p = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
    const V3 pv = transform_point_4x3(p, cam.view);
    present[idx] = (pv.z <= 0.2f) ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------------
// Data-parallel training: compressed exchange of the SH gradient (no reference counterpart).
// K9's SH gradient of one view is an outer product: dL_dsh[k] = basis_k(dir) * dL_dRGB with dL_dRGB = the blend's colour gradient
// masked by the forward's clamp bits (sh_backward above) -- 12 bytes of information expanded to 192.  Ranks therefore exchange
// dL_dRGB (all-gather, 12 B per Gaussian and view) instead of all-reducing dL_dsh (192 B), and every rank expands the sum
//     dL_dsh[k] = sum_views basis_k(normalize(mean - campos_view)) * dL_dRGB_view
// locally, views in rank order (deterministic, identical on every rank; each product is bit-identical to the one K9 forms).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sh_grad_pack(int P, const float* __restrict__ dL_dcolor, const uint8_t* __restrict__ clamped, const int32_t* __restrict__ radii,
             float* __restrict__ packed)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    V3 g = { 0.f, 0.f, 0.f };
    if (radii[idx] > 0) {                                   // preprocess_bwd skips the others (backward.cu:612-613)
        const uint32_t cb = clamped[idx];
        g = { dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2] };
        g.x *= (cb & 1u) ? 0 : 1;                           // backward.cu:36-38, as sh_backward
        g.y *= (cb & 2u) ? 0 : 1;
        g.z *= (cb & 4u) ? 0 : 1;
    }
    packed[3 * idx] = g.x; packed[3 * idx + 1] = g.y; packed[3 * idx + 2] = g.z;
}

// One thread per Gaussian: per view the direction and the dRGBdsh_k of sh_backward (the same expressions) once, 48 running sums in
// registers; the workgroup's 256 x 192-byte tile of results leaves through LDS with coalesced stores (rows padded to 49 floats as
// in preprocess_bwd<true>) -- to one [P,M,3] tensor or to the reference's separate DC / higher-band tensors
// (scene/gaussian_model.py:351-352).  View v: packed + v*packed_stride is [P][3], campos + v*campos_stride is [3] (strides in
// floats: the all-gathered buffer carries each rank's camera centre after its P rows).
template <int MC>                                          // MC = 16: the coefficient count as a constant (index arithmetic), 0: any M
__global__ void __launch_bounds__(256)
sh_grad_expand(int P, int D, int M, int n_views, const float* __restrict__ means3D, const float* __restrict__ campos, long campos_stride,
               const float* __restrict__ packed, long packed_stride, float scale, float* __restrict__ out_dc, long stride_dc,
               float* __restrict__ out_rest, long stride_rest)
{
    extern __shared__ float s_tile[];                       // [256][3 M + 1]
    if (MC) M = MC;
    const int row_len = 3 * M + 1;
    const long base = (long)blockIdx.x * 256;
    const long idx = base + threadIdx.x;
    V3 acc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = { 0.f, 0.f, 0.f };
    if (idx < P) {
        const V3 mean = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        for (int v = 0; v < n_views; v++) {
            const float* g = packed + (size_t)v * packed_stride + (size_t)idx * 3;
            const V3 rgb = { g[0], g[1], g[2] };
            if (rgb.x == 0.f && rgb.y == 0.f && rgb.z == 0.f) continue;       // culled or not reached in this view
            const float* c = campos + (size_t)v * campos_stride;
            const V3 cp = { c[0], c[1], c[2] };
            const V3 dir_orig = mean - cp;
            const V3 dir = dir_orig / sqrtf(dot3(dir_orig, dir_orig));
            const float x = dir.x, y = dir.y, z = dir.z;
            acc[0] = acc[0] + SH_C0 * rgb;
            if (D > 0) {
                acc[1] = acc[1] + (-SH_C1 * y) * rgb;
                acc[2] = acc[2] + (SH_C1 * z) * rgb;
                acc[3] = acc[3] + (-SH_C1 * x) * rgb;
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z;
                    const float xy = x * y, yz = y * z, xz = x * z;
                    acc[4] = acc[4] + (SH_C2[0] * xy) * rgb;
                    acc[5] = acc[5] + (SH_C2[1] * yz) * rgb;
                    acc[6] = acc[6] + (SH_C2[2] * (2.f * zz - xx - yy)) * rgb;
                    acc[7] = acc[7] + (SH_C2[3] * xz) * rgb;
                    acc[8] = acc[8] + (SH_C2[4] * (xx - yy)) * rgb;
                    if (D > 2) {
                        acc[9] = acc[9] + (SH_C3[0] * y * (3.f * xx - yy)) * rgb;
                        acc[10] = acc[10] + (SH_C3[1] * xy * z) * rgb;
                        acc[11] = acc[11] + (SH_C3[2] * y * (4.f * zz - xx - yy)) * rgb;
                        acc[12] = acc[12] + (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * rgb;
                        acc[13] = acc[13] + (SH_C3[4] * x * (4.f * zz - xx - yy)) * rgb;
                        acc[14] = acc[14] + (SH_C3[5] * z * (xx - yy)) * rgb;
                        acc[15] = acc[15] + (SH_C3[6] * x * (xx - 3.f * yy)) * rgb;
                    }
                }
            }
        }
    }
    float* my = s_tile + threadIdx.x * row_len;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < M) {
            const V3 r = scale * acc[k];
            my[3 * k] = r.x; my[3 * k + 1] = r.y; my[3 * k + 2] = r.z;
        }
    }
    __syncthreads();
    const int rows = (int)min((long)256, (long)P - base);
    const int per_row = 3 * M;
    if (out_rest == out_dc + 3 && stride_dc == per_row && stride_rest == per_row) {
        // one [P,M,3] tensor: the tile is one contiguous run of rows * 3M floats
        float* o = out_dc + base * per_row;
        if (MC && rows == 256 && (((uintptr_t)o) & 15) == 0) {
            for (int j = threadIdx.x * 4; j < 256 * 3 * MC; j += 1024) {
                float4 q;
                { const int r = j / (3 * MC), c = j - r * (3 * MC); q.x = s_tile[r * row_len + c]; }
                { const int r = (j + 1) / (3 * MC), c = j + 1 - r * (3 * MC); q.y = s_tile[r * row_len + c]; }
                { const int r = (j + 2) / (3 * MC), c = j + 2 - r * (3 * MC); q.z = s_tile[r * row_len + c]; }
                { const int r = (j + 3) / (3 * MC), c = j + 3 - r * (3 * MC); q.w = s_tile[r * row_len + c]; }
                *reinterpret_cast<float4*>(o + j) = q;
            }
        } else {
            for (int j = threadIdx.x; j < rows * per_row; j += 256) {
                const int r = j / per_row, c = j - r * per_row;
                o[j] = s_tile[r * row_len + c];
            }
        }
    } else {
        // separate DC [P,1,3] and higher-band [P,M-1,3] tensors (or any other pair of row strides)
        for (int j = threadIdx.x; j < rows * 3; j += 256) {
            const int r = j / 3, c = j - r * 3;
            out_dc[(base + r) * stride_dc + c] = s_tile[r * row_len + c];
        }
        const int rest = per_row - 3;
        for (int j = threadIdx.x; j < rows * rest; j += 256) {
            const int r = j / rest, c = j - r * rest;
            out_rest[(base + r) * stride_rest + c] = s_tile[r * row_len + 3 + c];
        }
    }
}

template __global__ void sh_grad_expand<16>(int, int, int, int, const float*, const float*, long, const float*, long, float, float*, long, float*, long);
template __global__ void sh_grad_expand<0>(int, int, int, int, const float*, const float*, long, const float*, long, float, float*, long, float*, long);

} // namespace gof
