// gaussian_model_ops.hip -- per-Gaussian bookkeeping of the reference GaussianModel on HIP (C ABI in include/gof_train_hip.h):
// compute_3D_filter, add_densification_stats and the three parameter activations render() reads every iteration.
//
// GaussianModel.compute_3D_filter (reference scene/gaussian_model.py:262-311) as two launches instead of
// ~25 torch kernels + two host-synchronising boolean-mask index ops PER CAMERA.  C ABI in include/gof_train_hip.h.
//
//   filter3d_min_depth: thread = point, the camera table streams through LDS in chunks of 128; per camera the arithmetic of
//                       :279-302 in source order (row-vector product xyz @ R + T, depth test on the UNclamped z, projection with
//                       the clamped z, 15 % screen margin); running min of the clamped z over the cameras that see the point;
//                       block-wide max of the seen points' depths -> one ordered-integer atomicMax.
//   filter3d_finish:    unseen points take that maximum (:304), then / max focal_x * sqrt(0.2) (:308).
// HBM-bound on 12 B read + 8 B written per point; the camera loop is ~25 flops per (point, camera) out of LDS.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/gof_hip.h"
#include "../../include/gof_train_hip.h"
#include "gof_common.h"

namespace gof {
size_t scan_tmp_words(size_t n);                 // radix.hip
hipError_t device_scan_u32(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool inclusive, uint32_t* tmp,
                           const uint32_t** total_dev_out, hipStream_t stream);

constexpr int F3_CHUNK = 128;

__device__ __forceinline__ uint32_t f3_ordered(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float f3_unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// state[0] = ordered max depth of the seen points, state[1] = any seen, state[2] = ordered max focal_x
__global__ void filter3d_init(uint32_t* state)
{
    if (threadIdx.x == 0) { state[0] = 0u; state[1] = 0u; state[2] = f3_ordered(0.0f); }
}

__global__ void __launch_bounds__(256)
filter3d_min_depth(int64_t P, const float* __restrict__ xyz, int num_cams, const float* __restrict__ cams,
                   float* __restrict__ distance, uint8_t* __restrict__ seen, uint32_t* __restrict__ state)
{
    __shared__ float s_cam[F3_CHUNK][GOF_FILTER_CAM_FLOATS];
    __shared__ float s_red[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < P;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (live) { px = xyz[3 * i]; py = xyz[3 * i + 1]; pz = xyz[3 * i + 2]; }
    float dist = 100000.0f;                                   // :267
    bool any = false;
    float fmax_ = 0.0f;                                       // :271
    for (int c0 = 0; c0 < num_cams; c0 += F3_CHUNK) {
        const int cn = min(F3_CHUNK, num_cams - c0);
        __syncthreads();
        for (int k = threadIdx.x; k < cn * GOF_FILTER_CAM_FLOATS; k += 256)
            (&s_cam[0][0])[k] = cams[(size_t)c0 * GOF_FILTER_CAM_FLOATS + k];
        __syncthreads();
        for (int c = 0; c < cn; c++) {
            const float* m = s_cam[c];
            // xyz @ R + T (:278): out_j = x R0j + y R1j + z R2j
            const float cx = px * m[0] + py * m[3] + pz * m[6] + m[9];
            const float cy = px * m[1] + py * m[4] + pz * m[7] + m[10];
            const float cz = px * m[2] + py * m[5] + pz * m[8] + m[11];
            const bool valid_depth = cz > 0.2f;               // :283
            const float z = fmaxf(cz, 0.001f);                // :287
            const float fx = m[12], fy = m[13], w = m[14], h = m[15];
            const float sx = cx / z * fx + w / 2.0f;          // :289-290
            const float sy = cy / z * fy + h / 2.0f;
            const bool in_screen = (sx >= -0.15f * w) & (sx <= w * 1.15f) & (sy >= -0.15f * h) & (sy <= 1.15f * h);   // :295
            if (valid_depth & in_screen) { dist = fminf(dist, z); any = true; }     // :298-302
            fmax_ = fmaxf(fmax_, fx);                         // :303-304
        }
    }
    if (live) { distance[i] = dist; seen[i] = any ? 1 : 0; }
    // max depth over the seen points of the block -> global
    float v = (live && any) ? dist : -3.0e38f;
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    const bool block_any = __syncthreads_or(live && any);
    if (threadIdx.x == 0) {
        if (block_any) {
            atomicMax(&state[0], f3_ordered(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]))));
            atomicOr(&state[1], 1u);
        }
        if (blockIdx.x == 0) state[2] = f3_ordered(fmax_);   // identical in every thread
    }
}

__global__ void __launch_bounds__(256)
filter3d_finish(int64_t P, const float* __restrict__ distance, const uint8_t* __restrict__ seen, const uint32_t* __restrict__ state,
                float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float dmax = f3_unordered(state[0]);
    const float focal = f3_unordered(state[2]);
    const float d = seen[i] ? distance[i] : dmax;             // :306
    out[i] = d / focal * (float)0.4472135954999579;           // :310: distance / focal_length * (0.2 ** 0.5)
}

// GaussianModel.add_densification_stats (gaussian_model.py:709-714)
__global__ void __launch_bounds__(256)
densification_stats_kernel(int64_t P, const float* __restrict__ grad, const uint8_t* __restrict__ filter, float* __restrict__ accum,
                           float* __restrict__ accum_abs, float* __restrict__ accum_abs_max, float* __restrict__ denom)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P || !filter[i]) return;
    const float gx = grad[3 * i], gy = grad[3 * i + 1], gz = grad[3 * i + 2];
    const float n2 = sqrtf(gx * gx + gy * gy);      // torch.norm(grad[filter, :2], dim=-1)
    const float n1 = sqrtf(gz * gz);                // torch.norm(grad[filter, 2:], dim=-1): sqrt(x^2), not |x| (underflows like torch)
    accum[i] += n2;
    accum_abs[i] += n1;
    accum_abs_max[i] = fmaxf(accum_abs_max[i], n1);
    denom[i] += 1.0f;
}

// ---- parameter activations (gaussian_model.py:157-166, 183-194); thread = Gaussian, streaming ------------------------------
__global__ void __launch_bounds__(256)
act_scaling_fwd(int64_t P, const float* __restrict__ rs, const float* __restrict__ f3, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float f = f3[i], ff = f * f;                          // torch.square(self.filter_3D)
#pragma unroll
    for (int c = 0; c < 3; c++) { const float s = expf(rs[3 * i + c]); out[3 * i + c] = sqrtf(s * s + ff); }     // :160-161
}
__global__ void __launch_bounds__(256)
act_scaling_bwd(int64_t P, const float* __restrict__ rs, const float* __restrict__ f3, const float* g, float* grs)      // grs may alias g (in-place, activations.INPLACE_GRAD)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float f = f3[i], ff = f * f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float s = expf(rs[3 * i + c]), ss = s * s;
        const float o = sqrtf(ss + ff);
        // d sqrt(u)/du = 1/(2 sqrt u); d(s^2)/ds = 2 s; d exp/dx = s   (autograd's chain, :158-161)
        grs[3 * i + c] = g[3 * i + c] / (2.0f * o) * (2.0f * s) * s;
    }
}
__global__ void __launch_bounds__(256)
act_opacity_fwd(int64_t P, const float* __restrict__ ro, const float* __restrict__ rs, const float* __restrict__ f3, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float f = f3[i], ff = f * f;
    const float s0 = expf(rs[3 * i]), s1 = expf(rs[3 * i + 1]), s2 = expf(rs[3 * i + 2]);
    const float q0 = s0 * s0, q1 = s1 * s1, q2 = s2 * s2;       // scales_square (:188)
    const float det1 = q0 * q1 * q2;                            // :189
    const float det2 = (q0 + ff) * (q1 + ff) * (q2 + ff);       // :191-192
    const float coef = sqrtf(det1 / det2);                      // :193
    const float o = 1.0f / (1.0f + expf(-ro[i]));               // sigmoid
    out[i] = o * coef;                                          // :194
}
__global__ void __launch_bounds__(256)
act_opacity_bwd(int64_t P, const float* __restrict__ ro, const float* __restrict__ rs, const float* __restrict__ f3, const float* g,
                float* gro, float* __restrict__ grs)                                                                    // gro may alias g
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float f = f3[i], ff = f * f;
    float q[3], a[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { const float s = expf(rs[3 * i + c]); q[c] = s * s; a[c] = q[c] + ff; }
    const float det1 = q[0] * q[1] * q[2], det2 = a[0] * a[1] * a[2];
    const float coef = sqrtf(det1 / det2);
    const float o = 1.0f / (1.0f + expf(-ro[i]));
    const float go = g[i];
    gro[i] = go * coef * (o * (1.0f - o));
    // d coef / d raw_scale_c = coef * (1 - q_c / a_c): coef = sqrt(r), r = prod q / prod a, d r / d q_c = r (1/q_c - 1/a_c), d q_c / d raw = 2 q_c
    const float gc = go * o * coef;
#pragma unroll
    for (int c = 0; c < 3; c++) grs[3 * i + c] = gc * (1.0f - q[c] / a[c]);
}
__global__ void __launch_bounds__(256)
act_rotation_fwd(int64_t P, const float4* __restrict__ rr, float4* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float4 r = rr[i];
    const float d = fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);     // F.normalize: v / max(||v||, eps)
    out[i] = make_float4(r.x / d, r.y / d, r.z / d, r.w / d);
}
__global__ void __launch_bounds__(256)
act_rotation_bwd(int64_t P, const float4* __restrict__ rr, const float4* g, float4* grr)                                  // grr may alias g
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float4 r = rr[i], go = g[i];
    const float n = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
    if (n >= 1e-12f) {
        const float inv = 1.0f / n;
        const float4 u = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
        const float ug = u.x * go.x + u.y * go.y + u.z * go.z + u.w * go.w;
        grr[i] = make_float4((go.x - u.x * ug) * inv, (go.y - u.y * ug) * inv, (go.z - u.z * ug) * inv, (go.w - u.w * ug) * inv);
    } else {
        grr[i] = make_float4(go.x / 1e-12f, go.y / 1e-12f, go.z / 1e-12f, go.w / 1e-12f);      // clamped denominator: constant
    }
}


// ---- densification (scene/gaussian_model.py:631-707): role of every Gaussian, ordered index lists, one-pass row gather ----------
// role: 0 = stays, 1 = cloned (stays + one new Gaussian), 2 = split (leaves + two new Gaussians).  The selection is the
// reference's: gradient condition `norm(grads) >= max_grad  or  norm(grads_abs) >= Q` (densify_and_clone :660-662; densify_and_split
// :636-643 applies the same thresholds to the same values), clone where max(get_scaling) <= percent_dense * extent, split where it
// is larger.  grads = xyz_gradient_accum / denom with NaN -> 0 (:686-690).  q_abs is a DEVICE scalar (the torch.quantile value).
__global__ void __launch_bounds__(256)
densify_roles_kernel(int64_t P, const float* __restrict__ accum, const float* __restrict__ accum_abs, const float* __restrict__ denom,
                     const float* __restrict__ scale_max, float max_grad, const float* __restrict__ q_abs, float size_threshold,
                     uint8_t* __restrict__ role, uint32_t* __restrict__ f_keep, uint32_t* __restrict__ f_clone, uint32_t* __restrict__ f_split)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float g = accum[i] / denom[i], ga = accum_abs[i] / denom[i];
    if (g != g) g = 0.0f;
    if (ga != ga) ga = 0.0f;
    // the clone test takes torch.norm over the size-1 last dimension = |g| (:660-661), the split test the raw value (:636-640);
    // fabsf is exact where sqrtf(g * g) underflowed for |g| < ~1e-19 (matters for a tiny or zero threshold)
    const float q = q_abs[0];
    const bool sel_clone = (fabsf(g) >= max_grad) || (fabsf(ga) >= q);
    const bool sel_split = (g >= max_grad) || (ga >= q);
    const bool small = scale_max[i] <= size_threshold;
    const uint8_t r = small ? (sel_clone ? 1 : 0) : (sel_split ? 2 : 0);
    role[i] = r;
    f_keep[i] = r != 2; f_clone[i] = r == 1; f_split[i] = r == 2;
}
// ordered index lists from the exclusive scans of the three flag arrays (in place: flags in, ranks in the same arrays)
__global__ void __launch_bounds__(256)
densify_lists_kernel(int64_t P, const uint8_t* __restrict__ role, const uint32_t* __restrict__ r_keep, const uint32_t* __restrict__ r_clone,
                     const uint32_t* __restrict__ r_split, int32_t* __restrict__ keep_idx, int32_t* __restrict__ clone_idx, int32_t* __restrict__ split_idx)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint8_t r = role[i];
    if (r != 2) keep_idx[r_keep[i]] = (int32_t)i;
    if (r == 1) clone_idx[r_clone[i]] = (int32_t)i;
    if (r == 2) split_idx[r_split[i]] = (int32_t)i;
}
// out[rank(i)] = src ? src[i] : i for the rows with keep[i] != 0 (rank = exclusive scan of keep)
__global__ void __launch_bounds__(256)
compact_rows_kernel(int64_t n, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ rank, const int32_t* __restrict__ src, int32_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !keep[i]) return;
    out[rank[i]] = src ? src[i] : (int32_t)i;
}
__global__ void __launch_bounds__(256)
bytes_to_flags_kernel(int64_t n, const uint8_t* __restrict__ b, uint32_t* __restrict__ f)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) f[i] = b[i] != 0;
}
// out[r, :] = src[row[r], :] if row[r] >= 0, else extra[-row[r] - 1, :] (zeros when extra == nullptr).  One thread per float4 (or float).
template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ float zero_of<float>() { return 0.0f; }
template <> __device__ __forceinline__ float4 zero_of<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <typename T>
__global__ void __launch_bounds__(256)
rows_gather_kernel(int64_t total, int32_t per_row, const int32_t* __restrict__ row, const T* __restrict__ src, const T* __restrict__ extra, T* __restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= total) return;
    const int64_t r = k / per_row;
    const int32_t c = (int32_t)(k - r * per_row);
    const int32_t s = row[r];
    T v;
    if (s >= 0) v = src[(int64_t)s * per_row + c];
    else if (extra) v = extra[(int64_t)(-s - 1) * per_row + c];
    else v = zero_of<T>();
    out[k] = v;
}

} // namespace gof

using namespace gof;

extern "C" {

size_t gof_filter3d_ws_bytes(int64_t P)
{
    const size_t n = (size_t)(P < 1 ? 1 : P);
    return 256 + ((n * 4 + 255) & ~(size_t)255) + ((n + 255) & ~(size_t)255) + 256;
}

int gof_compute_3d_filter(int64_t P, const float* xyz, int32_t num_cams, const float* cameras, float* filter_3D, void* ws, size_t ws_bytes,
                          int32_t* any_valid_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (any_valid_host) *any_valid_host = 0;
    if (P < 0 || num_cams < 0) { set_error("bad sizes (%lld points, %d cameras)", (long long)P, num_cams); return GOF_E_INVALID; }
    if (P == 0) return GOF_OK;
    if (!xyz || !filter_3D || !ws || (num_cams > 0 && !cameras)) { set_error("a pointer is NULL"); return GOF_E_INVALID; }
    if (ws_bytes < gof_filter3d_ws_bytes(P)) { set_error("filter workspace too small"); return GOF_E_WORKSPACE; }
    char* base = reinterpret_cast<char*>((reinterpret_cast<size_t>(ws) + 255) & ~(size_t)255);
    uint32_t* state = reinterpret_cast<uint32_t*>(base);
    float* distance = reinterpret_cast<float*>(base + 256);
    uint8_t* seen = reinterpret_cast<uint8_t*>(base + 256 + (((size_t)P * 4 + 255) & ~(size_t)255));
    const unsigned blocks = (unsigned)((P + 255) / 256);
    { GOF_PROFILE("compute_3d_filter", stream);
      hipLaunchKernelGGL(filter3d_init, dim3(1), dim3(64), 0, stream, state);
      hipLaunchKernelGGL(filter3d_min_depth, dim3(blocks), dim3(256), 0, stream, P, xyz, (int)num_cams, cameras, distance, seen, state);
      hipLaunchKernelGGL(filter3d_finish, dim3(blocks), dim3(256), 0, stream, P, distance, seen, state, filter_3D);
      GOF_LAUNCH_CHECK(stream, 0); }
    if (any_valid_host) {
        uint32_t flag = 0;
        GOF_HIP_CHECK(hipMemcpyAsync(&flag, state + 1, sizeof(flag), hipMemcpyDeviceToHost, stream));
        GOF_HIP_CHECK(hipStreamSynchronize(stream));
        *any_valid_host = (int32_t)flag;
    }
    return GOF_OK;
}


size_t gof_densify_ws_bytes(int64_t n)
{
    const size_t m = (size_t)(n < 1 ? 1 : n);
    return 256 + 3 * ((m * 4 + 255) & ~(size_t)255) + 3 * ((scan_tmp_words(m) * 4 + 255) & ~(size_t)255) + 256;
}

int gof_densify_select(int64_t P, const float* accum, const float* accum_abs, const float* denom, const float* scale_max, float max_grad,
                       const float* q_abs_dev, float size_threshold, uint8_t* role, int32_t* keep_idx, int32_t* clone_idx, int32_t* split_idx,
                       void* ws, size_t ws_bytes, int64_t* counts_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (counts_host) counts_host[0] = counts_host[1] = counts_host[2] = 0;
    if (P < 0) { set_error("bad number of points"); return GOF_E_INVALID; }
    if (P == 0) return GOF_OK;
    if (P >= (int64_t)1 << 31) { set_error("more than 2^31 points"); return GOF_E_INVALID; }
    if (!accum || !accum_abs || !denom || !scale_max || !q_abs_dev || !role || !keep_idx || !clone_idx || !split_idx || !ws || !counts_host) { set_error("a pointer is NULL"); return GOF_E_INVALID; }
    if (ws_bytes < gof_densify_ws_bytes(P)) { set_error("densify workspace too small"); return GOF_E_WORKSPACE; }
    char* base = reinterpret_cast<char*>((reinterpret_cast<size_t>(ws) + 255) & ~(size_t)255);
    const size_t fl = ((size_t)P * 4 + 255) & ~(size_t)255, st = (scan_tmp_words((size_t)P) * 4 + 255) & ~(size_t)255;
    uint32_t* f[3]; uint32_t* t[3];
    for (int k = 0; k < 3; k++) { f[k] = reinterpret_cast<uint32_t*>(base + k * fl); t[k] = reinterpret_cast<uint32_t*>(base + 3 * fl + k * st); }
    const unsigned blocks = (unsigned)((P + 255) / 256);
    GOF_PROFILE("densify_select", stream);
    hipLaunchKernelGGL(densify_roles_kernel, dim3(blocks), dim3(256), 0, stream, P, accum, accum_abs, denom, scale_max, max_grad, q_abs_dev, size_threshold,
                       role, f[0], f[1], f[2]);
    const uint32_t* tot[3];
    for (int k = 0; k < 3; k++) GOF_HIP_CHECK(device_scan_u32(f[k], nullptr, f[k], (size_t)P, false, t[k], &tot[k], stream));
    hipLaunchKernelGGL(densify_lists_kernel, dim3(blocks), dim3(256), 0, stream, P, role, f[0], f[1], f[2], keep_idx, clone_idx, split_idx);
    GOF_LAUNCH_CHECK(stream, 0);
    uint32_t h[3] = { 0, 0, 0 };
    for (int k = 0; k < 3; k++) GOF_HIP_CHECK(hipMemcpyAsync(&h[k], tot[k], sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    for (int k = 0; k < 3; k++) counts_host[k] = h[k];
    return GOF_OK;
}

int gof_compact_rows(int64_t n, const uint8_t* keep, const int32_t* src_rows, int32_t* out_rows, void* ws, size_t ws_bytes, int64_t* count_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (count_host) *count_host = 0;
    if (n < 0 || n >= (int64_t)1 << 31) { set_error("bad row count"); return GOF_E_INVALID; }
    if (n == 0) return GOF_OK;
    if (!keep || !out_rows || !ws || !count_host) { set_error("a pointer is NULL"); return GOF_E_INVALID; }
    if (ws_bytes < gof_densify_ws_bytes(n)) { set_error("compaction workspace too small"); return GOF_E_WORKSPACE; }
    char* base = reinterpret_cast<char*>((reinterpret_cast<size_t>(ws) + 255) & ~(size_t)255);
    const size_t fl = ((size_t)n * 4 + 255) & ~(size_t)255;
    uint32_t* flags = reinterpret_cast<uint32_t*>(base);
    uint32_t* tmp = reinterpret_cast<uint32_t*>(base + 3 * fl);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    GOF_PROFILE("compact_rows", stream);
    hipLaunchKernelGGL(bytes_to_flags_kernel, dim3(blocks), dim3(256), 0, stream, n, keep, flags);
    const uint32_t* tot = nullptr;
    GOF_HIP_CHECK(device_scan_u32(flags, nullptr, flags, (size_t)n, false, tmp, &tot, stream));
    hipLaunchKernelGGL(compact_rows_kernel, dim3(blocks), dim3(256), 0, stream, n, keep, flags, src_rows, out_rows);
    GOF_LAUNCH_CHECK(stream, 0);
    uint32_t h = 0;
    GOF_HIP_CHECK(hipMemcpyAsync(&h, tot, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    *count_host = h;
    return GOF_OK;
}

int gof_rows_gather(int64_t n_rows, int32_t floats_per_row, const int32_t* rows, const float* src, const float* extra, float* out, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_rows < 0 || floats_per_row <= 0) { set_error("bad sizes"); return GOF_E_INVALID; }
    if (n_rows == 0) return GOF_OK;
    if (!rows || !src || !out) { set_error("a pointer is NULL"); return GOF_E_INVALID; }
    GOF_PROFILE("rows_gather", stream);
    const bool v4 = (floats_per_row % 4 == 0) && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(extra)) % 16 == 0);
    if (v4) {
        const int64_t total = n_rows * (floats_per_row / 4);
        hipLaunchKernelGGL(rows_gather_kernel<float4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, total, floats_per_row / 4, rows,
                           reinterpret_cast<const float4*>(src), reinterpret_cast<const float4*>(extra), reinterpret_cast<float4*>(out));
    } else {
        const int64_t total = n_rows * floats_per_row;
        hipLaunchKernelGGL(rows_gather_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, total, floats_per_row, rows, src, extra, out);
    }
    GOF_LAUNCH_CHECK(stream, 0);
    return GOF_OK;
}

int gof_add_densification_stats(int64_t P, const float* viewspace_grad, const uint8_t* update_filter, float* accum, float* accum_abs,
                                float* accum_abs_max, float* denom, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P < 0) { set_error("bad number of points"); return GOF_E_INVALID; }
    if (P == 0) return GOF_OK;
    if (!viewspace_grad || !update_filter || !accum || !accum_abs || !accum_abs_max || !denom) { set_error("a pointer is NULL"); return GOF_E_INVALID; }
    GOF_PROFILE("add_densification_stats", stream);
    hipLaunchKernelGGL(densification_stats_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, P, viewspace_grad, update_filter,
                       accum, accum_abs, accum_abs_max, denom);
    GOF_LAUNCH_CHECK(stream, 0);
    return GOF_OK;
}

// Y[i][n] = sum_j A[i][j] X[j][n] over n < N, A = the 3x3 block at M (element (i, j) at M[i * rs + j * cs]; transposed: (j, i)).
// train.py:177-179 rotates the rendered normals into world space with `c2w[:3, :3] @ render_normal.reshape(3, -1)`: torch hands that
// 3 x 3 x 1.7M product to a GEMM library (128 us forward + 125 us backward on MI355X: a tile shape built for large K); it is 40 MB
// of streaming -- one launch of this kernel each way (train_epilogue/pose.py: SmallMatrix.__matmul__).
__global__ void __launch_bounds__(256)
rot3_apply_kernel(int64_t N, const float* __restrict__ M, int rs, int cs, int transpose, const float* __restrict__ X, float* __restrict__ Y)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float a[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) a[i][j] = transpose ? M[j * rs + i * cs] : M[i * rs + j * cs];
    const float x0 = X[n], x1 = X[N + n], x2 = X[2 * N + n];
#pragma unroll
    for (int i = 0; i < 3; i++) Y[(int64_t)i * N + n] = fmaf(a[i][2], x2, fmaf(a[i][1], x1, a[i][0] * x0));
}
int gof_rot3_apply(int64_t N, const float* M, int32_t row_stride, int32_t col_stride, int32_t transpose, const float* X, float* Y, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N < 0) { set_error("bad number of columns"); return GOF_E_INVALID; }
    if (N == 0) return GOF_OK;
    if (!M || !X || !Y) { set_error("rot3_apply: a pointer is NULL"); return GOF_E_INVALID; }
    GOF_PROFILE("rot3_apply", stream);
    hipLaunchKernelGGL(rot3_apply_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N, M, (int)row_stride, (int)col_stride, (int)transpose, X, Y);
    GOF_LAUNCH_CHECK(stream, 0);
    return GOF_OK;
}

#define GOF_ACT_ENTRY(NAME, KERNEL, NULLCHECK, ...)                                                                  \
    {                                                                                                                \
        hipStream_t stream = static_cast<hipStream_t>(stream_);                                                      \
        if (P < 0) { set_error("bad number of points"); return GOF_E_INVALID; }                                      \
        if (P == 0) return GOF_OK;                                                                                   \
        if (NULLCHECK) { set_error(NAME ": a pointer is NULL"); return GOF_E_INVALID; }                              \
        GOF_PROFILE(NAME, stream);                                                                                   \
        hipLaunchKernelGGL(KERNEL, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, stream, P, __VA_ARGS__);          \
        GOF_LAUNCH_CHECK(stream, 0);                                                                                 \
        return GOF_OK;                                                                                               \
    }

int gof_act_scaling(int64_t P, const float* rs, const float* f3, float* out, void* stream_)
    GOF_ACT_ENTRY("act_scaling", act_scaling_fwd, !rs || !f3 || !out, rs, f3, out)
int gof_act_scaling_backward(int64_t P, const float* rs, const float* f3, const float* g, float* grs, void* stream_)
    GOF_ACT_ENTRY("act_scaling_backward", act_scaling_bwd, !rs || !f3 || !g || !grs, rs, f3, g, grs)
int gof_act_opacity(int64_t P, const float* ro, const float* rs, const float* f3, float* out, void* stream_)
    GOF_ACT_ENTRY("act_opacity", act_opacity_fwd, !ro || !rs || !f3 || !out, ro, rs, f3, out)
int gof_act_opacity_backward(int64_t P, const float* ro, const float* rs, const float* f3, const float* g, float* gro, float* grs, void* stream_)
    GOF_ACT_ENTRY("act_opacity_backward", act_opacity_bwd, !ro || !rs || !f3 || !g || !gro || !grs, ro, rs, f3, g, gro, grs)
int gof_act_rotation(int64_t P, const float* rr, float* out, void* stream_)
    GOF_ACT_ENTRY("act_rotation", act_rotation_fwd, !rr || !out, reinterpret_cast<const float4*>(rr), reinterpret_cast<float4*>(out))
int gof_act_rotation_backward(int64_t P, const float* rr, const float* g, float* grr, void* stream_)
    GOF_ACT_ENTRY("act_rotation_backward", act_rotation_bwd, !rr || !g || !grr, reinterpret_cast<const float4*>(rr),
                  reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(grr))

} // extern "C"
