// mtets.hip -- marching tetrahedra on the GPU (replaces the pure-torch
// utils/tetmesh.py:47-138 `_unbatched_marching_tetrahedra` of the reference).
//
// The reference builds the mesh with generic torch ops: boolean gathers, a (6*Tv, 2) int64
// torch.unique(dim=0) (a full sort) and table gathers, chunked at 32 Mi tets with a second
// torch.unique per chunk merge (tetmesh.py:55-95).  Here it is two phases over caller-owned
// scratch, integer work only (HBM-bound: no floating point besides `sdf > 0`):
//
//   count:  classify every tet (4 sdf reads -> 4-bit case index), exclusive scans of
//           {valid, 1-triangle, 2-triangle} flags, emit the 6 (min,max) vertex pairs of every valid
//           tet as one u64 key, radix-sort the keys, mark first occurrences (= torch.unique order:
//           ascending (min, max)), scan the unique flags and the "crossing" flags
//           (exactly one end inside, tetmesh.py:113) -> #edges, #faces.
//   emit:   write edge end-point ids / positions / sdf / scales for crossing edges; every valid tet
//           binary-searches its 6 edge keys in the unique list and writes its 1 or 2 triangles from
//           the triangle table at its scanned position (1-triangle tets first, then 2-triangle tets,
//           per 32 Mi-tet chunk: the order tetmesh.py:126-136 / :55-95 produces).
//
// All indices are int64 (the reference's dtype); vertex ids must be < 2^32 (u64 edge key).
#include "gof_common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace gof {

__constant__ int8_t MT_TRI[16][6] = {                      // tetmesh.py:23-40
    {-1,-1,-1,-1,-1,-1},{1,0,2,-1,-1,-1},{4,0,3,-1,-1,-1},{1,4,2,1,3,4},
    {3,1,5,-1,-1,-1},{2,3,0,2,5,3},{1,4,0,1,5,4},{4,2,5,-1,-1,-1},
    {4,5,2,-1,-1,-1},{4,1,0,4,5,1},{3,2,0,3,5,2},{1,3,5,-1,-1,-1},
    {4,1,2,4,3,1},{3,0,4,-1,-1,-1},{2,0,1,-1,-1,-1},{-1,-1,-1,-1,-1,-1} };
__constant__ int8_t MT_NTRI[16] = { 0,1,1,2,1,2,2,1,1,2,2,1,2,1,1,0 };      // tetmesh.py:42
__constant__ int8_t MT_EDGE[12] = { 0,1, 0,2, 0,3, 1,2, 1,3, 2,3 };         // tetmesh.py:43

constexpr int64_t MT_CHUNK = 32ll * 1024 * 1024;                             // tetmesh.py:54

struct MtWs {
    uint8_t* tetcase;      // [Tt] 4-bit case index, 0xFF = not valid
    int64_t* vscan;        // [Tt+1] exclusive scan of valid
    int64_t* f1scan;       // [Tt+1] exclusive scan of (ntri == 1)
    int64_t* f2scan;       // [Tt+1] exclusive scan of (ntri == 2)
    uint64_t* ekeys;       // [6*Tt] edge keys of valid tets, emission order
    uint64_t* ekeys_sorted;// [6*Tt]
    int64_t* uscan;        // [6*Tt+1] exclusive scan of "first occurrence"
    uint64_t* ukeys;       // [6*Tt] unique keys
    int64_t* cscan;        // [6*Tt+1] exclusive scan of "crossing" over unique keys
    int64_t* counters;     // [8] Tv, U, E, F1, F2
    void* tmp; size_t tmp_bytes;
};

template <typename T>
static inline void carve(char*& p, T*& ptr, size_t count)
{
    p = reinterpret_cast<char*>(align_up(reinterpret_cast<size_t>(p)));
    ptr = reinterpret_cast<T*>(p);
    p += count * sizeof(T);
}

static size_t mt_layout(int64_t Tt, void* base, MtWs* out)
{
    MtWs w;
    char* p = static_cast<char*>(base);
    const size_t n = (size_t)Tt, e = 6 * n;
    carve(p, w.tetcase, n);
    carve(p, w.vscan, n + 1);
    carve(p, w.f1scan, n + 1);
    carve(p, w.f2scan, n + 1);
    carve(p, w.ekeys, e);
    carve(p, w.ekeys_sorted, e);
    carve(p, w.uscan, e + 1);
    carve(p, w.ukeys, e);
    carve(p, w.cscan, e + 1);
    carve(p, w.counters, 8);
    size_t t1 = 0, t2 = 0;
    { int64_t* q = nullptr; (void)rocprim::exclusive_scan(nullptr, t1, q, q, (int64_t)0, e + 1, rocprim::plus<int64_t>()); }
    { uint64_t* q = nullptr; (void)rocprim::radix_sort_keys(nullptr, t2, q, q, e, 0, 64); }
    w.tmp_bytes = t1 > t2 ? t1 : t2;
    char* tp; carve(p, tp, w.tmp_bytes); w.tmp = tp;
    if (out) *out = w;
    return (size_t)(p - static_cast<char*>(base)) + ALIGN;
}

__global__ void __launch_bounds__(256)
mt_classify(int64_t Tt, const int64_t* __restrict__ tets, const float* __restrict__ sdf, uint8_t* __restrict__ tetcase,
            int64_t* __restrict__ vflag, int64_t* __restrict__ f1flag, int64_t* __restrict__ f2flag)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t > Tt) return;
    if (t == Tt) { vflag[t] = 0; f1flag[t] = 0; f2flag[t] = 0; return; }   // sentinel so scan[Tt] = total
    const longlong4 v = reinterpret_cast<const longlong4*>(tets)[t];
    const int idx = (sdf[v.x] > 0 ? 1 : 0) | (sdf[v.y] > 0 ? 2 : 0) | (sdf[v.z] > 0 ? 4 : 0) | (sdf[v.w] > 0 ? 8 : 0);
    const int nt = MT_NTRI[idx];                       // 0 for all-out / all-in (tetmesh.py:102)
    tetcase[t] = nt ? (uint8_t)idx : (uint8_t)0xFF;
    vflag[t] = nt ? 1 : 0;
    f1flag[t] = (nt == 1);
    f2flag[t] = (nt == 2);
}

__global__ void __launch_bounds__(256)
mt_emit_edges(int64_t Tt, const int64_t* __restrict__ tets, const uint8_t* __restrict__ tetcase, const int64_t* __restrict__ vscan,
              uint64_t* __restrict__ ekeys)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Tt || tetcase[t] == 0xFF) return;
    const int64_t v[4] = { tets[4 * t], tets[4 * t + 1], tets[4 * t + 2], tets[4 * t + 3] };
    const int64_t slot = vscan[t];
#pragma unroll
    for (int e = 0; e < 6; e++) {
        uint64_t a = (uint64_t)v[MT_EDGE[2 * e]], b = (uint64_t)v[MT_EDGE[2 * e + 1]];
        if (a > b) { const uint64_t s = a; a = b; b = s; }                    // tetmesh.py:107-108
        ekeys[6 * slot + e] = (a << 32) | b;
    }
}

__global__ void __launch_bounds__(256)
mt_mark_unique(int64_t n, const uint64_t* __restrict__ sorted, int64_t* __restrict__ uflag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    uflag[i] = (i < n) ? ((i == 0 || sorted[i] != sorted[i - 1]) ? 1 : 0) : 0;
}

__global__ void __launch_bounds__(256)
mt_compact_unique(int64_t n, const uint64_t* __restrict__ sorted, const int64_t* __restrict__ uscan, const float* __restrict__ sdf,
                  uint64_t* __restrict__ ukeys, int64_t* __restrict__ cflag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || sorted[i] != sorted[i - 1]) {
        const int64_t u = uscan[i];
        const uint64_t k = sorted[i];
        ukeys[u] = k;
        const int s2 = (sdf[k >> 32] > 0 ? 1 : 0) + (sdf[k & 0xFFFFFFFFull] > 0 ? 1 : 0);
        cflag[u] = (s2 == 1);                                                // tetmesh.py:113
    }
}

__global__ void __launch_bounds__(256)
mt_write_edges(int64_t U, const uint64_t* __restrict__ ukeys, const int64_t* __restrict__ cscan, const float* __restrict__ vertices,
               const float* __restrict__ sdf, const float* __restrict__ scales, int64_t* __restrict__ edge_ids,
               float* __restrict__ edge_pos, float* __restrict__ edge_sdf, float* __restrict__ edge_scales)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    const int64_t m = cscan[u];
    if (cscan[u + 1] == m) return;      // not a crossing edge
    const uint64_t k = ukeys[u];
    const int64_t ab[2] = { (int64_t)(k >> 32), (int64_t)(k & 0xFFFFFFFFull) };
#pragma unroll
    for (int e = 0; e < 2; e++) {
        edge_ids[2 * m + e] = ab[e];
        edge_pos[(2 * m + e) * 3 + 0] = vertices[3 * ab[e] + 0];
        edge_pos[(2 * m + e) * 3 + 1] = vertices[3 * ab[e] + 1];
        edge_pos[(2 * m + e) * 3 + 2] = vertices[3 * ab[e] + 2];
        edge_sdf[2 * m + e] = sdf[ab[e]];
        edge_scales[2 * m + e] = scales[ab[e]];
    }
}

__device__ __forceinline__ int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
mt_write_faces(int64_t Tt, int64_t per_chunk, const int64_t* __restrict__ tets, const uint8_t* __restrict__ tetcase,
               const int64_t* __restrict__ f1scan, const int64_t* __restrict__ f2scan, int64_t U, const uint64_t* __restrict__ ukeys,
               const int64_t* __restrict__ cscan, int64_t* __restrict__ faces)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Tt) return;
    const uint8_t idx = tetcase[t];
    if (idx == 0xFF) return;
    const int nt = MT_NTRI[idx];
    const int64_t c0 = (t / per_chunk) * per_chunk;
    const int64_t c1 = (c0 + per_chunk < Tt) ? c0 + per_chunk : Tt;
    // faces are emitted chunk by chunk, inside a chunk all 1-triangle tets first (see header)
    const int64_t fidx = (nt == 1) ? 2 * f2scan[c0] + f1scan[t] : f1scan[c1] + 2 * f2scan[t];
    const int64_t v[4] = { tets[4 * t], tets[4 * t + 1], tets[4 * t + 2], tets[4 * t + 3] };
    int64_t em[6];
#pragma unroll
    for (int e = 0; e < 6; e++) {
        uint64_t a = (uint64_t)v[MT_EDGE[2 * e]], b = (uint64_t)v[MT_EDGE[2 * e + 1]];
        if (a > b) { const uint64_t s = a; a = b; b = s; }
        const int64_t pos = lower_bound_u64(ukeys, U, (a << 32) | b);
        em[e] = (cscan[pos + 1] != cscan[pos]) ? cscan[pos] : -1;              // tetmesh.py:114-116
    }
    for (int k = 0; k < 3 * nt; k++) {
        const int e = MT_TRI[idx][k];
        int64_t val = em[0];
#pragma unroll
        for (int q = 1; q < 6; q++) val = (e == q) ? em[q] : val;
        faces[3 * fidx + k] = val;
    }
}

static inline int64_t per_chunk_of(int64_t Tt)
{
    if (Tt <= MT_CHUNK) return Tt > 0 ? Tt : 1;
    const int64_t nchunks = Tt / MT_CHUNK + 1;          // torch.chunk(tets, Tt // chunk_size + 1), tetmesh.py:60
    return (Tt + nchunks - 1) / nchunks;
}

} // namespace gof

using namespace gof;

extern "C" {

size_t gof_mtets_ws_bytes(int64_t num_tets) { return mt_layout(num_tets < 0 ? 0 : num_tets, nullptr, nullptr) + ALIGN; }

int gof_mtets_count(int64_t V, int64_t Tt, const int64_t* tets, const float* sdf, void* ws, size_t ws_bytes,
                    int64_t* num_edges_host, int64_t* num_faces_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!num_edges_host || !num_faces_host) { set_error("mtets: output pointers are NULL"); return GOF_E_INVALID; }
    *num_edges_host = 0; *num_faces_host = 0;
    if (Tt < 0 || V < 0 || V >= (1ll << 32)) { set_error("mtets: bad sizes (V must be < 2^32)"); return GOF_E_INVALID; }
    if (Tt == 0) return GOF_OK;
    if (!tets || !sdf || !ws) { set_error("mtets: NULL input"); return GOF_E_INVALID; }
    if (ws_bytes < gof_mtets_ws_bytes(Tt)) { set_error("mtets: workspace too small"); return GOF_E_WORKSPACE; }
    MtWs w;
    mt_layout(Tt, reinterpret_cast<void*>(align_up(reinterpret_cast<size_t>(ws))), &w);
    const dim3 blk(256);
    const auto grid = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    // flags are written into the scan arrays and scanned in place
    hipLaunchKernelGGL(mt_classify, grid(Tt + 1), blk, 0, stream, Tt, tets, sdf, w.tetcase, w.vscan, w.f1scan, w.f2scan);
    GOF_LAUNCH_CHECK(stream, 0);
    for (int64_t* arr : { w.vscan, w.f1scan, w.f2scan })
        GOF_HIP_CHECK(rocprim::exclusive_scan(w.tmp, w.tmp_bytes, arr, arr, (int64_t)0, (size_t)Tt + 1, rocprim::plus<int64_t>(), stream));
    int64_t host[3];
    GOF_HIP_CHECK(hipMemcpyAsync(&host[0], w.vscan + Tt, 8, hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipMemcpyAsync(&host[1], w.f1scan + Tt, 8, hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipMemcpyAsync(&host[2], w.f2scan + Tt, 8, hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t Tv = host[0], F = host[1] + 2 * host[2];
    int64_t U = 0, E = 0;
    if (Tv > 0) {
        const int64_t ne = 6 * Tv;
        hipLaunchKernelGGL(mt_emit_edges, grid(Tt), blk, 0, stream, Tt, tets, w.tetcase, w.vscan, w.ekeys);
        GOF_LAUNCH_CHECK(stream, 0);
        GOF_HIP_CHECK(rocprim::radix_sort_keys(w.tmp, w.tmp_bytes, w.ekeys, w.ekeys_sorted, (size_t)ne, 0, 64, stream));
        hipLaunchKernelGGL(mt_mark_unique, grid(ne + 1), blk, 0, stream, ne, w.ekeys_sorted, w.uscan);
        GOF_LAUNCH_CHECK(stream, 0);
        GOF_HIP_CHECK(rocprim::exclusive_scan(w.tmp, w.tmp_bytes, w.uscan, w.uscan, (int64_t)0, (size_t)ne + 1, rocprim::plus<int64_t>(), stream));
        GOF_HIP_CHECK(hipMemcpyAsync(&U, w.uscan + ne, 8, hipMemcpyDeviceToHost, stream));
        GOF_HIP_CHECK(hipStreamSynchronize(stream));
        GOF_HIP_CHECK(hipMemsetAsync(w.cscan, 0, (size_t)(U + 1) * 8, stream));
        hipLaunchKernelGGL(mt_compact_unique, grid(ne), blk, 0, stream, ne, w.ekeys_sorted, w.uscan, sdf, w.ukeys, w.cscan);
        GOF_LAUNCH_CHECK(stream, 0);
        GOF_HIP_CHECK(rocprim::exclusive_scan(w.tmp, w.tmp_bytes, w.cscan, w.cscan, (int64_t)0, (size_t)U + 1, rocprim::plus<int64_t>(), stream));
        GOF_HIP_CHECK(hipMemcpyAsync(&E, w.cscan + U, 8, hipMemcpyDeviceToHost, stream));
        GOF_HIP_CHECK(hipStreamSynchronize(stream));
    }
    const int64_t cnt[5] = { Tv, U, E, host[1], host[2] };
    GOF_HIP_CHECK(hipMemcpyAsync(w.counters, cnt, sizeof(cnt), hipMemcpyHostToDevice, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    *num_edges_host = E;
    *num_faces_host = F;
    return GOF_OK;
}

int gof_mtets_emit(int64_t V, int64_t Tt, const int64_t* tets, const float* vertices, const float* sdf, const float* scales,
                   const void* ws, size_t ws_bytes, int64_t num_edges, int64_t num_faces, int64_t* edge_ids, float* edge_pos,
                   float* edge_sdf, float* edge_scales, int64_t* faces, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    (void)V;
    if (Tt <= 0 || (num_edges == 0 && num_faces == 0)) return GOF_OK;
    if (!tets || !vertices || !sdf || !scales || !ws || !edge_ids || !edge_pos || !edge_sdf || !edge_scales || !faces) {
        set_error("mtets: NULL pointer"); return GOF_E_INVALID; }
    if (ws_bytes < gof_mtets_ws_bytes(Tt)) { set_error("mtets: workspace too small"); return GOF_E_WORKSPACE; }
    MtWs w;
    mt_layout(Tt, reinterpret_cast<void*>(align_up(reinterpret_cast<size_t>(ws))), &w);
    int64_t cnt[5];
    GOF_HIP_CHECK(hipMemcpyAsync(cnt, w.counters, sizeof(cnt), hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t U = cnt[1];
    if (cnt[2] != num_edges || cnt[3] + 2 * cnt[4] != num_faces) { set_error("mtets: counts do not match the workspace (run gof_mtets_count first)"); return GOF_E_INVALID; }
    const dim3 blk(256);
    const auto grid = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    if (U > 0) {
        hipLaunchKernelGGL(mt_write_edges, grid(U), blk, 0, stream, U, w.ukeys, w.cscan, vertices, sdf, scales, edge_ids, edge_pos, edge_sdf, edge_scales);
        GOF_LAUNCH_CHECK(stream, 0);
    }
    hipLaunchKernelGGL(mt_write_faces, grid(Tt), blk, 0, stream, Tt, per_chunk_of(Tt), tets, w.tetcase, w.f1scan, w.f2scan, U, w.ukeys, w.cscan, faces);
    GOF_LAUNCH_CHECK(stream, 0);
    return GOF_OK;
}

} // extern "C"
