// knn.hip -- mean squared distance to the 3 nearest neighbours (replaces reference submodules/simple-knn/simple_knn.cu).
//
// Same idea as the reference (Morton order -> boxes -> conservative pruning -> exact 3-NN), laid out for wave64 / LDS:
//   1. bounding box of the points AND the origin (the reference reduces with init = {0,0,0}, simple_knn.cu:191-199) with
//      order-preserving integer atomics; no host read-back (the reference copies min/max to the host twice);
//   2. 30-bit Morton codes (simple_knn.cu:39-57), stable radix sort (the library's own, 4 passes);
//   3. the points are GATHERED into Morton order once (float4), boxes of 256 consecutive points (one workgroup) and groups
//      of 32 boxes get their min/max;
//   4. one workgroup per box: every thread owns one query point; the workgroup walks the group boxes, then the boxes of the
//      groups some thread still needs (__syncthreads_or), stages the 256 candidates of a box in LDS (coalesced float4) and
//      every thread that needs the box scans them from LDS (same-address broadcasts).  The reference lets every thread walk
//      all boxes alone and gather candidates through an index array (uncoalesced, divergent).
// Pruning is exact in floating point: every operation of the box distance is monotone in |difference|, so
// fl(dist to box) <= fl(dist to any point of the box); a box is skipped only if its distance exceeds the current 3rd best
// (simple_knn.cu:171-173), ties are visited.
#include <hip/hip_runtime.h>
#include <cfloat>
#include "../../include/gof_hip.h"
#include "../../include/gof_knn_hip.h"
#include "gof_common.h"

namespace gof {

constexpr int KNN_BOX = 256;
constexpr int KNN_GROUP = 32;     // boxes per group

hipError_t radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int end_bit,
                                uint32_t* tmp, uint32_t** keys_res, uint32_t** vals_res, hipStream_t stream, const uint32_t* n_dev = nullptr);
size_t rs_tmp_words(size_t n);

struct KnnBox { float lo[3]; float hi[3]; float pad[2]; };

__device__ __forceinline__ uint32_t ordered(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ void __launch_bounds__(256)
knn_minmax(int64_t N, const float* __restrict__ pts, uint32_t* __restrict__ mm)   // mm[0..2] = min, mm[3..5] = max (ordered encoding)
{
    float lo[3] = { 0.f, 0.f, 0.f }, hi[3] = { 0.f, 0.f, 0.f };                  // init = origin (simple_knn.cu:191)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256)
#pragma unroll
        for (int c = 0; c < 3; c++) { const float v = pts[3 * i + c]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&mm[c], ordered(lo[c])); atomicMax(&mm[3 + c], ordered(hi[c])); }
    }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x)                     // simple_knn.cu:39-46
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ void __launch_bounds__(256)
knn_morton(int64_t N, const float* __restrict__ pts, const uint32_t* __restrict__ mm, uint32_t* __restrict__ codes, uint32_t* __restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float lo = unordered(mm[c]), hi = unordered(mm[3 + c]);
        // simple_knn.cu:50-52; a degenerate axis (hi == lo) maps to cell 0 instead of converting a NaN (the order only
        // affects the pruning efficiency, never the result)
        const float ext = hi - lo;
        const float cell = ext > 0.f ? ((pts[3 * i + c] - lo) / ext) * (float)((1 << 10) - 1) : 0.f;
        code |= prep_morton((uint32_t)fminf(fmaxf(cell, 0.f), 1023.f)) << c;
    }
    codes[i] = code;
    idx[i] = (uint32_t)i;
}

// gather into Morton order + box bounds; one workgroup per box
__global__ void __launch_bounds__(KNN_BOX)
knn_gather_boxes(int64_t N, const float* __restrict__ pts, const uint32_t* __restrict__ order, float4* __restrict__ sorted, KnnBox* __restrict__ boxes)
{
    const int64_t i = (int64_t)blockIdx.x * KNN_BOX + threadIdx.x;
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };   // simple_knn.cu:88-92
    if (i < N) {
        const size_t s = order[i];
        const float x = pts[3 * s], y = pts[3 * s + 1], z = pts[3 * s + 2];
        sorted[i] = make_float4(x, y, z, 0.f);
        lo[0] = hi[0] = x; lo[1] = hi[1] = y; lo[2] = hi[2] = z;
    }
    __shared__ float s_lo[3][KNN_BOX / 64], s_hi[3][KNN_BOX / 64];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
        if ((threadIdx.x & 63) == 0) { s_lo[c][threadIdx.x >> 6] = lo[c]; s_hi[c][threadIdx.x >> 6] = hi[c]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        KnnBox b;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            b.lo[c] = fminf(fminf(s_lo[c][0], s_lo[c][1]), fminf(s_lo[c][2], s_lo[c][3]));
            b.hi[c] = fmaxf(fmaxf(s_hi[c][0], s_hi[c][1]), fmaxf(s_hi[c][2], s_hi[c][3]));
        }
        b.pad[0] = b.pad[1] = 0.f;
        boxes[blockIdx.x] = b;
    }
}

__global__ void __launch_bounds__(64)
knn_group_boxes(int64_t num_boxes, const KnnBox* __restrict__ boxes, KnnBox* __restrict__ groups)
{
    const int64_t g = blockIdx.x;
    const int64_t b = g * KNN_GROUP + (threadIdx.x & (KNN_GROUP - 1));
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    if (b < num_boxes && threadIdx.x < KNN_GROUP) {
#pragma unroll
        for (int c = 0; c < 3; c++) { lo[c] = boxes[b].lo[c]; hi[c] = boxes[b].hi[c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
        for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
    if (threadIdx.x == 0) {
        KnnBox r;
#pragma unroll
        for (int c = 0; c < 3; c++) { r.lo[c] = lo[c]; r.hi[c] = hi[c]; }
        r.pad[0] = r.pad[1] = 0.f;
        groups[g] = r;
    }
}

__device__ __forceinline__ float dist_box_point(const KnnBox& box, float px, float py, float pz)    // simple_knn.cu:126-136
{
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (px < box.lo[0] || px > box.hi[0]) dx = fminf(fabsf(px - box.lo[0]), fabsf(px - box.hi[0]));
    if (py < box.lo[1] || py > box.hi[1]) dy = fminf(fabsf(py - box.lo[1]), fabsf(py - box.hi[1]));
    if (pz < box.lo[2] || pz > box.hi[2]) dz = fminf(fabsf(pz - box.lo[2]), fabsf(pz - box.hi[2]));
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ void update3(float px, float py, float pz, float4 c, float best[3])    // updateKBest<3>, simple_knn.cu:139-153
{
    const float dx = c.x - px, dy = c.y - py, dz = c.z - pz;
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
}

__global__ void __launch_bounds__(KNN_BOX)
knn_search(int64_t N, const float4* __restrict__ sorted, const uint32_t* __restrict__ order, const KnnBox* __restrict__ boxes,
           const KnnBox* __restrict__ groups, int64_t num_boxes, int64_t num_groups, float* __restrict__ out)
{
    __shared__ float4 s_pts[KNN_BOX];
    const int64_t mybox = blockIdx.x;
    const int64_t i = mybox * KNN_BOX + threadIdx.x;
    const bool valid = i < N;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) q = sorted[i];
    float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };

    // own box first: its 255 other points give a tight bound before anything is pruned
    s_pts[threadIdx.x] = q;
    __syncthreads();
    {
        const int cnt = (int)min((int64_t)KNN_BOX, N - mybox * KNN_BOX);
        if (valid)
            for (int j = 0; j < cnt; j++)
                if (j != (int)threadIdx.x) update3(q.x, q.y, q.z, s_pts[j], best);      // skips SELF by position only (simple_knn.cu:176-177)
    }
    for (int64_t g = 0; g < num_groups; g++) {
        const bool need_g = valid && !(dist_box_point(groups[g], q.x, q.y, q.z) > best[2]);
        if (!__syncthreads_or(need_g)) continue;
        const int64_t b1 = min(num_boxes, (g + 1) * KNN_GROUP);
        for (int64_t b = g * KNN_GROUP; b < b1; b++) {
            if (b == mybox) continue;
            const bool need = valid && !(dist_box_point(boxes[b], q.x, q.y, q.z) > best[2]);     // simple_knn.cu:171-173
            if (!__syncthreads_or(need)) continue;                                                  // (also the barrier before restaging)
            const int64_t j0 = b * KNN_BOX;
            const int cnt = (int)min((int64_t)KNN_BOX, N - j0);
            if ((int)threadIdx.x < cnt) s_pts[threadIdx.x] = sorted[j0 + threadIdx.x];
            __syncthreads();
            if (need)
                for (int j = 0; j < cnt; j++) update3(q.x, q.y, q.z, s_pts[j], best);
        }
    }
    if (valid) out[order[i]] = (best[0] + best[1] + best[2]) / 3.0f;                               // simple_knn.cu:183
}

struct KnnWs { uint32_t* mm; uint32_t *ka, *kb, *va, *vb, *tmp; float4* sorted; KnnBox* boxes; KnnBox* groups; };
static size_t knn_layout(int64_t N, void* base, KnnWs* out)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t n = (size_t)(N < 1 ? 1 : N);
    const size_t nb = (n + KNN_BOX - 1) / KNN_BOX, ng = (nb + KNN_GROUP - 1) / KNN_GROUP;
    size_t off = 0;
    char* p = static_cast<char*>(base);
    KnnWs w;
    auto carve = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += up(bytes); return r; };
    w.mm = (uint32_t*)carve(8 * 4);
    w.ka = (uint32_t*)carve(n * 4); w.kb = (uint32_t*)carve(n * 4); w.va = (uint32_t*)carve(n * 4); w.vb = (uint32_t*)carve(n * 4);
    w.tmp = (uint32_t*)carve(rs_tmp_words(n) * 4);
    w.sorted = (float4*)carve(n * 16);
    w.boxes = (KnnBox*)carve(nb * sizeof(KnnBox));
    w.groups = (KnnBox*)carve(ng * sizeof(KnnBox));
    if (out) *out = w;
    return off + 256;
}

__global__ void knn_init_mm(uint32_t* mm)
{
    if (threadIdx.x < 6) mm[threadIdx.x] = 0x80000000u;      // ordered(0.0f): the origin
}

} // namespace gof

using namespace gof;

extern "C" {

size_t gof_knn_ws_bytes(int64_t N) { return knn_layout(N, nullptr, nullptr); }

int gof_knn_mean_dist3(int64_t N, const float* points, float* mean_dists, void* ws, size_t ws_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N < 0 || N >= ((int64_t)1 << 31)) { set_error("bad number of points (%lld)", (long long)N); return GOF_E_INVALID; }
    if (N == 0) return GOF_OK;
    if (!points || !mean_dists || !ws) { set_error("points / mean_dists / workspace is NULL"); return GOF_E_INVALID; }
    if (ws_bytes < gof_knn_ws_bytes(N)) { set_error("knn workspace too small"); return GOF_E_WORKSPACE; }
    KnnWs w;
    knn_layout(N, reinterpret_cast<void*>((reinterpret_cast<size_t>(ws) + 255) & ~(size_t)255), &w);
    const int64_t nb = (N + KNN_BOX - 1) / KNN_BOX, ng = (nb + KNN_GROUP - 1) / KNN_GROUP;
    GOF_PROFILE("knn_mean_dist3", stream);
    hipLaunchKernelGGL(knn_init_mm, dim3(1), dim3(64), 0, stream, w.mm);
    hipLaunchKernelGGL(knn_minmax, dim3((unsigned)min((int64_t)2048, (N + 255) / 256)), dim3(256), 0, stream, N, points, w.mm);
    hipLaunchKernelGGL(knn_morton, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, N, points, w.mm, w.ka, w.va);
    GOF_LAUNCH_CHECK(stream, 0);
    uint32_t *kr = nullptr, *vr = nullptr;
    GOF_HIP_CHECK(radix_sort_pairs_u32(w.ka, w.va, w.kb, w.vb, (size_t)N, 30, w.tmp, &kr, &vr, stream));
    hipLaunchKernelGGL(knn_gather_boxes, dim3((unsigned)nb), dim3(KNN_BOX), 0, stream, N, points, vr, w.sorted, w.boxes);
    hipLaunchKernelGGL(knn_group_boxes, dim3((unsigned)ng), dim3(64), 0, stream, nb, w.boxes, w.groups);
    hipLaunchKernelGGL(knn_search, dim3((unsigned)nb), dim3(KNN_BOX), 0, stream, N, w.sorted, vr, w.boxes, w.groups, nb, ng, mean_dists);
    GOF_LAUNCH_CHECK(stream, 0);
    return GOF_OK;
}

} // extern "C"
