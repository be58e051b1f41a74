// ref_capi.cpp -- TEST INFRASTRUCTURE.  extern "C" wrapper around the REFERENCE rasterizer
// (CudaRasterizer::Rasterizer, reference cuda_rasterizer/rasterizer.h:20-124) compiled for gfx950 by
// oracle/build_ref.sh from the sources under /root/reference.  It plays the role of the torch binding
// (reference rasterize_points.cu) without torch: growable device buffers + raw pointers.
// All pointers are DEVICE pointers; GofRasterArgs is the product's POD (include/gof_hip.h).
#include <cstdio>
#include <functional>
#include <string>
#include <hip/hip_runtime.h>
#include "cuda_runtime.h"
#define GLM_FORCE_CUDA
#include <glm/glm.hpp>
#include "rasterizer.h"
#include "rasterizer_impl.h"
#include "../include/gof_hip.h"

namespace {
struct Buf {
    char* p = nullptr; size_t cap = 0;
    char* get(size_t n) { if (n > cap) { if (p) hipFree(p); hipMalloc((void**)&p, n ? n : 1); cap = n; } return p; }
    ~Buf() { if (p) hipFree(p); }
};
struct State { Buf geom, binning, img, point, pbinning; int P = 0, W = 0, H = 0, R = 0, PN = 0; };
std::function<char*(size_t)> fn(Buf& b) { return [&b](size_t n) { return b.get(n); }; }
}

extern "C" {
void* cudaref_create() { return new State(); }
void cudaref_destroy(void* s) { delete static_cast<State*>(s); }

int cudaref_forward(void* sp, const GofRasterArgs* a, float* out_color, int* radii)
{
    State& s = *static_cast<State*>(sp);
    s.P = a->P; s.W = a->W; s.H = a->H;
    hipMemset(out_color, 0, sizeof(float) * 9 * (size_t)a->W * a->H);
    hipMemset(radii, 0, sizeof(int) * (size_t)a->P);
    if (a->P == 0) { s.R = 0; return 0; }
    s.R = CudaRasterizer::Rasterizer::forward(fn(s.geom), fn(s.binning), fn(s.img), a->P, a->D, a->M, a->background, a->W, a->H,
        a->means3D, a->shs, a->colors_precomp, a->opacities, a->scales, a->scale_modifier, a->rotations, a->cov3D_precomp,
        a->view2gaussian_precomp, a->viewmatrix, a->projmatrix, a->campos, a->tan_fovx, a->tan_fovy, a->kernel_size,
        a->subpixel_offset, a->prefiltered != 0, out_color, radii, a->debug != 0);
    hipDeviceSynchronize();
    return s.R;
}

int cudaref_backward(void* sp, const GofRasterArgs* a, const int* radii, const float* dL_dout, float* dL_dmeans2D, float* dL_dcolors,
                     float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations,
                     float* dL_dview2gaussian, float* dL_dconic_scratch /* [P,4] */)
{
    State& s = *static_cast<State*>(sp);
    const size_t P = (size_t)a->P;
    hipMemset(dL_dmeans2D, 0, 12 * P); hipMemset(dL_dcolors, 0, 12 * P); hipMemset(dL_dopacity, 0, 4 * P);
    hipMemset(dL_dmeans3D, 0, 12 * P); hipMemset(dL_dcov3D, 0, 24 * P); if (a->M) hipMemset(dL_dsh, 0, 12 * P * a->M);
    hipMemset(dL_dscales, 0, 12 * P); hipMemset(dL_drotations, 0, 16 * P); hipMemset(dL_dview2gaussian, 0, 40 * P);
    hipMemset(dL_dconic_scratch, 0, 16 * P);
    CudaRasterizer::Rasterizer::backward(a->P, a->D, a->M, s.R, a->background, a->W, a->H, a->means3D, a->shs, a->colors_precomp,
        a->view2gaussian_precomp, a->scales, a->scale_modifier, a->rotations, a->cov3D_precomp, a->viewmatrix, a->projmatrix,
        a->campos, a->tan_fovx, a->tan_fovy, a->kernel_size, a->subpixel_offset, radii, s.geom.p, s.binning.p, s.img.p, dL_dout,
        dL_dmeans2D, dL_dconic_scratch, dL_dopacity, dL_dcolors, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
        dL_dview2gaussian, a->debug != 0);
    hipDeviceSynchronize();
    return 0;
}

int cudaref_integrate(void* sp, const GofRasterArgs* a, int PN, const float* points3D, float* out_color, float* out_alpha,
                      float* out_color_pts, int* radii)
{
    State& s = *static_cast<State*>(sp);
    s.P = a->P; s.W = a->W; s.H = a->H; s.PN = PN;
    // the binding pre-fills: out_color 0, alpha 1, colour 0 (rasterize_points.cu:275-278) -- done by the caller for alpha
    hipMemset(out_color, 0, sizeof(float) * 9 * (size_t)a->W * a->H);
    hipMemset(out_color_pts, 0, sizeof(float) * 3 * (size_t)PN);
    hipMemset(radii, 0, sizeof(int) * (size_t)a->P);
    s.R = CudaRasterizer::Rasterizer::integrate(fn(s.geom), fn(s.binning), fn(s.img), fn(s.point), fn(s.pbinning), PN, a->P, a->D, a->M,
        a->background, a->W, a->H, points3D, a->means3D, a->shs, a->colors_precomp, a->opacities, a->scales, a->scale_modifier,
        a->rotations, a->cov3D_precomp, a->view2gaussian_precomp, a->viewmatrix, a->projmatrix, a->campos, a->tan_fovx, a->tan_fovy,
        a->kernel_size, a->subpixel_offset, a->prefiltered != 0, out_color, radii, out_alpha, out_color_pts, a->debug != 0);
    hipDeviceSynchronize();
    return s.R;
}

// device pointer + element count of a named intermediate of the last forward (element size by name, see tests)
const void* cudaref_fetch(void* sp, const char* name, long long* count)
{
    State& s = *static_cast<State*>(sp);
    using namespace CudaRasterizer;
    char* g = s.geom.p; char* b = s.binning.p; char* i = s.img.p;
    GeometryState gs = GeometryState::fromChunk(g, s.P);
    BinningState bs = BinningState::fromChunk(b, s.R);
    ImageState is = ImageState::fromChunk(i, (size_t)s.W * s.H);
    const std::string n(name);
    const size_t P = s.P, R = s.R, N = (size_t)s.W * s.H, T = (size_t)((s.W + 15) / 16) * ((s.H + 15) / 16);
    if (n == "depths") { *count = P; return gs.depths; }
    if (n == "means2D") { *count = 2 * P; return gs.means2D; }
    if (n == "cov3D") { *count = 6 * P; return gs.cov3D; }
    if (n == "view2gaussian") { *count = 10 * P; return gs.view2gaussian; }
    if (n == "conic_opacity") { *count = 4 * P; return gs.conic_opacity; }
    if (n == "rgb") { *count = 3 * P; return gs.rgb; }
    if (n == "clamped") { *count = 3 * P; return gs.clamped; }
    if (n == "tiles_touched") { *count = P; return gs.tiles_touched; }
    if (n == "point_offsets") { *count = P; return gs.point_offsets; }
    if (n == "point_list") { *count = R; return bs.point_list; }
    if (n == "point_list_keys") { *count = R; return bs.point_list_keys; }
    if (n == "ranges") { *count = 2 * T; return is.ranges; }
    if (n == "final_T") { *count = 4 * N; return is.accum_alpha; }
    if (n == "n_contrib") { *count = 2 * N; return is.n_contrib; }
    *count = -1;
    return nullptr;
}
}
