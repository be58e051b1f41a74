// mtets.hip -- marching tetrahedra on the GPU (replaces the pure-torch
// utils/tetmesh.py:47-138 `_unbatched_marching_tetrahedra` of the reference).
//
// The reference builds the mesh with generic torch ops: boolean gathers, a (6*Tv, 2) int64
// torch.unique(dim=0) (a full sort) and table gathers, chunked at 32 Mi tets with a second
// torch.unique per chunk merge (tetmesh.py:55-95).  Here it is two phases over caller-owned
// scratch, integer work only (HBM-bound: no floating point besides `sdf > 0`):
//
//   count:  classify every tet (4 sdf reads -> 4-bit case index, 1 byte per tet); the three exclusive scans
//           {valid, 1-triangle, 2-triangle} come from that byte array in two passes (per-block counts, scan of
//           the block counts, apply: 2 B read + 12 B written per tet); emit the 6 (min,max) vertex pairs of
//           every valid tet; sort them by (min, max) with two stable radix sorts of (key, value) pairs --
//           by max, then by min -- on ceil(log2 V) bits each (radix.hip, the rasterizer's own sort);
//           mark first occurrences (= torch.unique order: ascending (min, max)), scan the unique flags
//           and the "crossing" flags (exactly one end inside, tetmesh.py:113) -> #edges, #faces.
//   emit:   write edge end-point ids / positions / sdf / scales for crossing edges; every valid tet
//           binary-searches its 6 edge keys in the unique list and writes its 1 or 2 triangles from
//           the triangle table at its scanned position (1-triangle tets first, then 2-triangle tets,
//           per 32 Mi-tet chunk: the order tetmesh.py:126-136 / :55-95 produces).
//
// Indices at the boundary are int64 (the reference's dtype); vertex ids must be < 2^32 and 6 * #tets < 2^32 (the scans and the
// sort count in 32 bits).  Scan and sort are the hand-written kernels of radix.hip -- no library primitives.
#include "gof_common.h"
#include <cstring>

namespace gof {

size_t scan_tmp_words(size_t n);
hipError_t device_scan_u32(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool inclusive, uint32_t* tmp,
                           const uint32_t** total_dev_out, hipStream_t stream);
size_t rs_tmp_words(size_t n);
hipError_t radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int end_bit,
                                uint32_t* tmp, uint32_t** keys_res, uint32_t** vals_res, hipStream_t stream, const uint32_t* n_dev);

__constant__ int8_t MT_TRI[16][6] = {                      // tetmesh.py:23-40
    {-1,-1,-1,-1,-1,-1},{1,0,2,-1,-1,-1},{4,0,3,-1,-1,-1},{1,4,2,1,3,4},
    {3,1,5,-1,-1,-1},{2,3,0,2,5,3},{1,4,0,1,5,4},{4,2,5,-1,-1,-1},
    {4,5,2,-1,-1,-1},{4,1,0,4,5,1},{3,2,0,3,5,2},{1,3,5,-1,-1,-1},
    {4,1,2,4,3,1},{3,0,4,-1,-1,-1},{2,0,1,-1,-1,-1},{-1,-1,-1,-1,-1,-1} };
__constant__ int8_t MT_NTRI[16] = { 0,1,1,2,1,2,2,1,1,2,2,1,2,1,1,0 };      // tetmesh.py:42
__constant__ int8_t MT_EDGE[12] = { 0,1, 0,2, 0,3, 1,2, 1,3, 2,3 };         // tetmesh.py:43

constexpr int64_t MT_CHUNK = 32ll * 1024 * 1024;                             // tetmesh.py:54
constexpr int MT_ITEMS = 16;                                                 // tets per thread in the counting passes
constexpr int MT_BLOCK = 256 * MT_ITEMS;                                     // tets per workgroup there
constexpr int MT_MAX_CHUNKS = 32;                                            // 6 * #tets < 2^32  ->  at most 23 chunks of 32 Mi tets

struct MtWs {
    uint8_t* tetcase;      // [Tt] 4-bit case index, 0xFF = not valid
    uint32_t* bsum;        // [3][nbp] per-block (4096 tets) counts of {valid, 1-triangle, 2-triangle} -> their exclusive scans
                           // (nbp = #blocks + 1).  There are no per-tet scan arrays: the kernels that need a tet's rank rebuild the
                           // block-local scan from the case bytes (1 B per tet) on top of these block prefixes
    uint32_t* vt;          // [Tv] ids of the valid tets, ascending (compacted)
    uint32_t* frank;       // [Tv] per valid tet: its rank among the 1-triangle (resp. 2-triangle) tets before it
    uint32_t* chunk_tab;   // [MT_MAX_CHUNKS + 1][2] #1-triangle / #2-triangle tets before every chunk boundary (tetmesh.py:55-95)
    uint32_t* e_lo[2];     // [6*Tv] larger vertex id of every edge of every valid tet (emission order) + sort buffer
    uint32_t* e_hi[2];     // [6*Tv] smaller vertex id
    uint32_t* uscan;       // [6*Tv+1] exclusive scan of "first occurrence"
    uint64_t* ukeys;       // [6*Tv] unique keys (min << 32 | max), ascending
    uint32_t* cscan;       // [6*Tv+1] exclusive scan of "crossing" over unique keys
    int64_t* counters;     // [8] Tv, U, E, F1, F2
    uint32_t* tmp;         // scan / sort scratch of the edge stage
    uint32_t* tmp_tet;     // scan scratch of the block counts
    size_t nbp;
};

template <typename T>
static inline void carve(char*& p, T*& ptr, size_t count)
{
    p = reinterpret_cast<char*>(align_up(reinterpret_cast<size_t>(p)));
    ptr = reinterpret_cast<T*>(p);
    p += count * sizeof(T);
}

// Two caller-owned workspaces: the per-TET part (1 byte per tet + block prefixes; known before anything ran) and the per-VALID-tet
// part (sized after gof_mtets_classify has returned how many tets the surface crosses -- a fraction of a percent of a
// Delaunay triangulation, so sizing the edge arrays for the worst case would cost ~200 B per tet).
static size_t mt_tet_layout(int64_t Tt, void* base, MtWs* w)
{
    MtWs tmp;
    MtWs& o = w ? *w : tmp;
    char* p = static_cast<char*>(base);
    const size_t n = (size_t)Tt;
    o.nbp = (n + 1 + MT_BLOCK - 1) / MT_BLOCK + 1;
    carve(p, o.tetcase, n);
    carve(p, o.bsum, 3 * o.nbp);
    carve(p, o.chunk_tab, 2 * (MT_MAX_CHUNKS + 1));
    carve(p, o.counters, 8);
    carve(p, o.tmp_tet, scan_tmp_words(o.nbp));
    return (size_t)(p - static_cast<char*>(base)) + ALIGN;
}
static size_t mt_edge_layout(int64_t Tv, void* base, MtWs* w)
{
    MtWs tmp;
    MtWs& o = w ? *w : tmp;
    char* p = static_cast<char*>(base);
    const size_t n = (size_t)Tv, e = 6 * n;
    carve(p, o.vt, n);
    carve(p, o.frank, n);
    carve(p, o.e_lo[0], e); carve(p, o.e_lo[1], e);
    carve(p, o.e_hi[0], e); carve(p, o.e_hi[1], e);
    carve(p, o.uscan, e + 1);
    carve(p, o.ukeys, e);
    carve(p, o.cscan, e + 1);
    size_t words = rs_tmp_words(e);
    if (scan_tmp_words(e + 1) > words) words = scan_tmp_words(e + 1);
    carve(p, o.tmp, words);
    return (size_t)(p - static_cast<char*>(base)) + ALIGN;
}

__global__ void __launch_bounds__(256)
mt_classify(int64_t Tt, const int64_t* __restrict__ tets, const float* __restrict__ sdf, uint8_t* __restrict__ tetcase)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Tt) return;
    const longlong4 v = reinterpret_cast<const longlong4*>(tets)[t];
    const int idx = (sdf[v.x] > 0 ? 1 : 0) | (sdf[v.y] > 0 ? 2 : 0) | (sdf[v.z] > 0 ? 4 : 0) | (sdf[v.w] > 0 ? 8 : 0);
    tetcase[t] = MT_NTRI[idx] ? (uint8_t)idx : (uint8_t)0xFF;          // 0 triangles for all-out / all-in (tetmesh.py:102)
}

// The case bytes of a thread's MT_ITEMS = 16 consecutive tets in two registers (byte k of the pair = tet base + k; 0xFF beyond Tt)
struct MtCases { uint64_t lo, hi; };
__device__ __forceinline__ uint32_t mt_case_at(const MtCases& c, int k) { return (uint32_t)(((k < 8) ? c.lo : c.hi) >> (8 * (k & 7))) & 0xFFu; }

// ... and the counts of {valid, 1-triangle, 2-triangle} among them
__device__ __forceinline__ MtCases mt_thread_counts(int64_t Tt, const uint8_t* __restrict__ tetcase, int64_t base, uint32_t c[3])
{
    MtCases cs;
    if (base + MT_ITEMS <= Tt) {
        const uint4 q = *reinterpret_cast<const uint4*>(tetcase + base);        // base is a multiple of 16
        cs.lo = (uint64_t)q.x | ((uint64_t)q.y << 32);
        cs.hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
    } else {
        cs.lo = cs.hi = ~0ull;
        for (int k = 0; k < MT_ITEMS; k++) {
            if (base + k < Tt) {
                const uint64_t v = tetcase[base + k];
                uint64_t& word = (k < 8) ? cs.lo : cs.hi;
                word = (word & ~(0xFFull << (8 * (k & 7)))) | (v << (8 * (k & 7)));
            }
        }
    }
    c[0] = c[1] = c[2] = 0;
#pragma unroll
    for (int k = 0; k < MT_ITEMS; k++) {
        const uint32_t cse = mt_case_at(cs, k);
        if (cse != 0xFFu) {
            const int nt = MT_NTRI[cse];
            c[0]++; c[1] += (nt == 1); c[2] += (nt == 2);
        }
    }
    return cs;
}

__global__ void __launch_bounds__(256)
mt_count_blocks(int64_t Tt, const uint8_t* __restrict__ tetcase, uint32_t* __restrict__ bsum, uint32_t nbp)
{
    __shared__ uint32_t s_red[3][4];
    uint32_t c[3];
    mt_thread_counts(Tt, tetcase, (int64_t)blockIdx.x * MT_BLOCK + (int64_t)threadIdx.x * MT_ITEMS, c);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t x = c[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) s_red[k][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3) bsum[(size_t)threadIdx.x * nbp + blockIdx.x] = s_red[threadIdx.x][0] + s_red[threadIdx.x][1] + s_red[threadIdx.x][2] + s_red[threadIdx.x][3];
}

// bsum holds the exclusive scans of the block counts.  Exclusive ranks {valid, 1-triangle, 2-triangle} of this thread's first tet
// (thread layout of mt_thread_counts: workgroup = 4096 consecutive tets, thread = 16 consecutive ones); cs = its case bytes.
__device__ __forceinline__ MtCases mt_thread_ranks(int64_t Tt, const uint8_t* __restrict__ tetcase, const uint32_t* __restrict__ bsum, uint32_t nbp,
                                                   uint32_t run[3], uint32_t (*s_wave)[4])
{
    uint32_t c[3];
    const MtCases cs = mt_thread_counts(Tt, tetcase, (int64_t)blockIdx.x * MT_BLOCK + (int64_t)threadIdx.x * MT_ITEMS, c);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t x = c[k];                                   // inclusive scan over the wave
        for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
        if (lane == 63) s_wave[k][wave] = x;
        run[k] = x - c[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t add = bsum[(size_t)k * nbp + blockIdx.x];
        for (int w = 0; w < wave; w++) add += s_wave[k][w];
        run[k] += add;
    }
    return cs;
}

// Compact the valid tets: vt[rank] = tet id, frank[rank] = its rank among the 1-triangle (2-triangle) tets.  The heavy per-tet work
// below then runs one thread per VALID tet (surface tets are clustered: 16 consecutive tets per thread would serialise them).
__global__ void __launch_bounds__(256)
mt_compact_valid(int64_t Tt, const uint8_t* __restrict__ tetcase, const uint32_t* __restrict__ bsum, uint32_t nbp,
                 uint32_t* __restrict__ vt, uint32_t* __restrict__ frank)
{
    __shared__ uint32_t s_wave[3][4];
    uint32_t run[3];
    const MtCases cs = mt_thread_ranks(Tt, tetcase, bsum, nbp, run, s_wave);
    const int64_t base = (int64_t)blockIdx.x * MT_BLOCK + (int64_t)threadIdx.x * MT_ITEMS;
#pragma unroll
    for (int k = 0; k < MT_ITEMS; k++) {
        const uint32_t idx = mt_case_at(cs, k);
        if (idx == 0xFFu) continue;
        const int nt = MT_NTRI[idx];
        vt[run[0]] = (uint32_t)(base + k);
        frank[run[0]] = (nt == 1) ? run[1] : run[2];
        run[0]++; run[1] += (nt == 1); run[2] += (nt == 2);
    }
}

__global__ void __launch_bounds__(256)
mt_emit_edges(int64_t Tv, const int64_t* __restrict__ tets, const uint32_t* __restrict__ vt, uint32_t* __restrict__ e_lo, uint32_t* __restrict__ e_hi)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Tv) return;
    const longlong4 q = reinterpret_cast<const longlong4*>(tets)[vt[i]];
    const int64_t v[4] = { q.x, q.y, q.z, q.w };
    const size_t slot = (size_t)i * 6;
#pragma unroll
    for (int e = 0; e < 6; e++) {
        uint32_t a = (uint32_t)v[MT_EDGE[2 * e]], b = (uint32_t)v[MT_EDGE[2 * e + 1]];
        if (a > b) { const uint32_t s = a; a = b; b = s; }                    // tetmesh.py:107-108
        e_hi[slot + e] = a;
        e_lo[slot + e] = b;
    }
}

// #1-triangle / #2-triangle tets before position x_i = min(i * per_chunk, Tt), i = blockIdx.x
__global__ void __launch_bounds__(256)
mt_chunk_table(int64_t Tt, int64_t per_chunk, const uint8_t* __restrict__ tetcase, const uint32_t* __restrict__ bsum, uint32_t nbp,
               uint32_t* __restrict__ tab)
{
    __shared__ uint32_t s_red[2][4];
    int64_t x = (int64_t)blockIdx.x * per_chunk;
    if (x > Tt) x = Tt;
    const int64_t b = x / MT_BLOCK, t0 = b * MT_BLOCK + (int64_t)threadIdx.x * MT_ITEMS;
    uint32_t c1 = 0, c2 = 0;
    for (int k = 0; k < MT_ITEMS; k++) {
        const int64_t t = t0 + k;
        if (t < x) {
            const uint8_t cse = tetcase[t];
            if (cse != 0xFF) { const int nt = MT_NTRI[cse]; c1 += (nt == 1); c2 += (nt == 2); }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int off = 32; off > 0; off >>= 1) { c1 += __shfl_down(c1, off, 64); c2 += __shfl_down(c2, off, 64); }
    if (lane == 0) { s_red[0][wave] = c1; s_red[1][wave] = c2; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int k = threadIdx.x;
        tab[2 * blockIdx.x + k] = bsum[(size_t)(k + 1) * nbp + b] + s_red[k][0] + s_red[k][1] + s_red[k][2] + s_red[k][3];
    }
}

__global__ void __launch_bounds__(256)
mt_mark_unique(int64_t n, const uint32_t* __restrict__ hi, const uint32_t* __restrict__ lo, uint32_t* __restrict__ uflag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    uflag[i] = (i < n) ? ((i == 0 || hi[i] != hi[i - 1] || lo[i] != lo[i - 1]) ? 1u : 0u) : 0u;
}

__global__ void __launch_bounds__(256)
mt_compact_unique(int64_t n, const uint32_t* __restrict__ hi, const uint32_t* __restrict__ lo, const uint32_t* __restrict__ uscan,
                  const float* __restrict__ sdf, uint64_t* __restrict__ ukeys, uint32_t* __restrict__ cflag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || hi[i] != hi[i - 1] || lo[i] != lo[i - 1]) {
        const uint32_t u = uscan[i];
        ukeys[u] = ((uint64_t)hi[i] << 32) | lo[i];
        const int s2 = (sdf[hi[i]] > 0 ? 1 : 0) + (sdf[lo[i]] > 0 ? 1 : 0);
        cflag[u] = (s2 == 1) ? 1u : 0u;                                      // tetmesh.py:113
    }
}

__global__ void __launch_bounds__(256)
mt_write_edges(int64_t U, const uint64_t* __restrict__ ukeys, const uint32_t* __restrict__ cscan, const float* __restrict__ vertices,
               const float* __restrict__ sdf, const float* __restrict__ scales, int64_t* __restrict__ edge_ids,
               float* __restrict__ edge_pos, float* __restrict__ edge_sdf, float* __restrict__ edge_scales)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    const int64_t m = cscan[u];
    if ((int64_t)cscan[u + 1] == m) return;      // not a crossing edge
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
mt_write_faces(int64_t Tv, int64_t per_chunk, const int64_t* __restrict__ tets, const uint8_t* __restrict__ tetcase,
               const uint32_t* __restrict__ vt, const uint32_t* __restrict__ frank, const uint32_t* __restrict__ chunk_tab, int64_t U,
               const uint64_t* __restrict__ ukeys, const uint32_t* __restrict__ cscan, int64_t* __restrict__ faces)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Tv) return;
    const int64_t t = vt[i];
    const uint32_t idx = tetcase[t];
    const int nt = MT_NTRI[idx];
    // faces are emitted chunk by chunk, inside a chunk all 1-triangle tets first (see header):
    // 1-triangle tet: 2 * F2(chunk start) + F1(t); 2-triangle tet: F1(chunk end) + 2 * F2(t)
    const int64_t ci = t / per_chunk;
    const int64_t fidx = (nt == 1) ? 2 * (int64_t)chunk_tab[2 * ci + 1] + (int64_t)frank[i]
                                   : (int64_t)chunk_tab[2 * (ci + 1)] + 2 * (int64_t)frank[i];
    const longlong4 q = reinterpret_cast<const longlong4*>(tets)[t];
    const int64_t v[4] = { q.x, q.y, q.z, q.w };
    int64_t em[6];
#pragma unroll
    for (int e = 0; e < 6; e++) {
        uint64_t a = (uint64_t)v[MT_EDGE[2 * e]], b = (uint64_t)v[MT_EDGE[2 * e + 1]];
        if (a > b) { const uint64_t s = a; a = b; b = s; }
        const int64_t pos = lower_bound_u64(ukeys, U, (a << 32) | b);
        em[e] = (cscan[pos + 1] != cscan[pos]) ? (int64_t)cscan[pos] : -1;        // tetmesh.py:114-116
    }
    for (int j = 0; j < 3 * nt; j++) {
        const int e = MT_TRI[idx][j];
        int64_t val = em[0];
#pragma unroll
        for (int q2 = 1; q2 < 6; q2++) val = (e == q2) ? em[q2] : val;
        faces[3 * fidx + j] = val;
    }
}

static inline int64_t per_chunk_of(int64_t Tt)
{
    if (Tt <= MT_CHUNK) return Tt > 0 ? Tt : 1;
    const int64_t nchunks = Tt / MT_CHUNK + 1;          // torch.chunk(tets, Tt // chunk_size + 1), tetmesh.py:60
    return (Tt + nchunks - 1) / nchunks;
}

static inline int bits_for(uint64_t max_value)
{
    int b = 1;
    while (b < 32 && (max_value >> b) != 0) b++;
    return b;
}

} // namespace gof

using namespace gof;

extern "C" {

size_t gof_mtets_tet_ws_bytes(int64_t num_tets) { return mt_tet_layout(num_tets < 0 ? 0 : num_tets, nullptr, nullptr) + ALIGN; }
size_t gof_mtets_edge_ws_bytes(int64_t num_valid_tets) { return mt_edge_layout(num_valid_tets < 0 ? 0 : num_valid_tets, nullptr, nullptr) + ALIGN; }

static int mt_check_sizes(int64_t V, int64_t Tt)
{
    if (Tt < 0 || V < 0 || V >= (1ll << 32)) { set_error("mtets: bad sizes (V must be < 2^32)"); return GOF_E_INVALID; }
    if (6 * Tt + 1 >= (1ll << 32)) { set_error("mtets: too many tets (%lld): 6 * #tets must be < 2^32", (long long)Tt); return GOF_E_INVALID; }
    return GOF_OK;
}

int gof_mtets_classify(int64_t V, int64_t Tt, const int64_t* tets, const float* sdf, void* tet_ws, size_t tet_ws_bytes,
                       int64_t* num_valid_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!num_valid_host) { set_error("mtets: output pointer is NULL"); return GOF_E_INVALID; }
    *num_valid_host = 0;
    if (int e = mt_check_sizes(V, Tt)) return e;
    if (Tt == 0) return GOF_OK;
    if (!tets || !sdf || !tet_ws) { set_error("mtets: NULL input"); return GOF_E_INVALID; }
    if (tet_ws_bytes < gof_mtets_tet_ws_bytes(Tt)) { set_error("mtets: tet workspace too small"); return GOF_E_WORKSPACE; }
    MtWs w;
    mt_tet_layout(Tt, reinterpret_cast<void*>(align_up(reinterpret_cast<size_t>(tet_ws))), &w);
    const dim3 blk(256);
    const uint32_t nbp = (uint32_t)w.nbp, nb = nbp - 1;
    hipLaunchKernelGGL(mt_classify, dim3((unsigned)((Tt + 255) / 256)), blk, 0, stream, Tt, tets, sdf, w.tetcase);
    GOF_LAUNCH_CHECK(stream, 0);
    GOF_HIP_CHECK(hipMemsetAsync(w.bsum, 0, 3 * (size_t)nbp * sizeof(uint32_t), stream));   // entry nb of each row stays 0 -> its scan = the total
    hipLaunchKernelGGL(mt_count_blocks, dim3(nb), blk, 0, stream, Tt, w.tetcase, w.bsum, nbp);
    GOF_LAUNCH_CHECK(stream, 0);
    for (int k = 0; k < 3; k++)
        GOF_HIP_CHECK(device_scan_u32(w.bsum + (size_t)k * nbp, nullptr, w.bsum + (size_t)k * nbp, nbp, false, w.tmp_tet, nullptr, stream));
    uint32_t host[3];
    for (int k = 0; k < 3; k++)
        GOF_HIP_CHECK(hipMemcpyAsync(&host[k], w.bsum + (size_t)k * nbp + nb, 4, hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t cnt[5] = { (int64_t)host[0], 0, 0, (int64_t)host[1], (int64_t)host[2] };
    GOF_HIP_CHECK(hipMemcpyAsync(w.counters, cnt, sizeof(cnt), hipMemcpyHostToDevice, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    *num_valid_host = host[0];
    return GOF_OK;
}

int gof_mtets_count(int64_t V, int64_t Tt, const int64_t* tets, const float* sdf, void* tet_ws, size_t tet_ws_bytes,
                    void* edge_ws, size_t edge_ws_bytes, int64_t* num_edges_host, int64_t* num_faces_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!num_edges_host || !num_faces_host) { set_error("mtets: output pointers are NULL"); return GOF_E_INVALID; }
    *num_edges_host = 0; *num_faces_host = 0;
    if (int e = mt_check_sizes(V, Tt)) return e;
    if (Tt == 0) return GOF_OK;
    if (!tets || !sdf || !tet_ws) { set_error("mtets: NULL input"); return GOF_E_INVALID; }
    if (tet_ws_bytes < gof_mtets_tet_ws_bytes(Tt)) { set_error("mtets: tet workspace too small"); return GOF_E_WORKSPACE; }
    MtWs w;
    mt_tet_layout(Tt, reinterpret_cast<void*>(align_up(reinterpret_cast<size_t>(tet_ws))), &w);
    int64_t cnt[5];
    GOF_HIP_CHECK(hipMemcpyAsync(cnt, w.counters, sizeof(cnt), hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t Tv = cnt[0], F = cnt[3] + 2 * cnt[4];
    if (Tv < 0 || Tv > Tt) { set_error("mtets: the tet workspace holds no classification (run gof_mtets_classify first)"); return GOF_E_INVALID; }
    const dim3 blk(256);
    const auto grid = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    const uint32_t nbp = (uint32_t)w.nbp, nb = nbp - 1;
    int64_t U = 0, E = 0;
    if (Tv > 0) {
        if (!edge_ws || edge_ws_bytes < gof_mtets_edge_ws_bytes(Tv)) { set_error("mtets: edge workspace too small for %lld valid tets", (long long)Tv); return GOF_E_WORKSPACE; }
        mt_edge_layout(Tv, reinterpret_cast<void*>(align_up(reinterpret_cast<size_t>(edge_ws))), &w);
        const int64_t ne = 6 * Tv;
        hipLaunchKernelGGL(mt_compact_valid, dim3(nb), blk, 0, stream, Tt, w.tetcase, w.bsum, nbp, w.vt, w.frank);
        GOF_LAUNCH_CHECK(stream, 0);
        hipLaunchKernelGGL(mt_emit_edges, grid(Tv), blk, 0, stream, Tv, tets, w.vt, w.e_lo[0], w.e_hi[0]);
        GOF_LAUNCH_CHECK(stream, 0);
        // ascending (min, max): stable sort by max, then stable sort by min (torch.unique(dim=0) order, tetmesh.py:110)
        const int bits = bits_for((uint64_t)(V > 0 ? V - 1 : 0));
        uint32_t *lo1 = nullptr, *hi1 = nullptr, *hi2 = nullptr, *lo2 = nullptr;
        GOF_HIP_CHECK(radix_sort_pairs_u32(w.e_lo[0], w.e_hi[0], w.e_lo[1], w.e_hi[1], (size_t)ne, bits, w.tmp, &lo1, &hi1, stream, nullptr));
        uint32_t* hi_other = (hi1 == w.e_hi[0]) ? w.e_hi[1] : w.e_hi[0];
        uint32_t* lo_other = (lo1 == w.e_lo[0]) ? w.e_lo[1] : w.e_lo[0];
        GOF_HIP_CHECK(radix_sort_pairs_u32(hi1, lo1, hi_other, lo_other, (size_t)ne, bits, w.tmp, &hi2, &lo2, stream, nullptr));
        hipLaunchKernelGGL(mt_mark_unique, grid(ne + 1), blk, 0, stream, ne, hi2, lo2, w.uscan);
        GOF_LAUNCH_CHECK(stream, 0);
        GOF_HIP_CHECK(device_scan_u32(w.uscan, nullptr, w.uscan, (size_t)ne + 1, false, w.tmp, nullptr, stream));
        uint32_t u32 = 0;
        GOF_HIP_CHECK(hipMemcpyAsync(&u32, w.uscan + ne, 4, hipMemcpyDeviceToHost, stream));
        GOF_HIP_CHECK(hipStreamSynchronize(stream));
        U = u32;
        GOF_HIP_CHECK(hipMemsetAsync(w.cscan, 0, (size_t)(U + 1) * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(mt_compact_unique, grid(ne), blk, 0, stream, ne, hi2, lo2, w.uscan, sdf, w.ukeys, w.cscan);
        GOF_LAUNCH_CHECK(stream, 0);
        GOF_HIP_CHECK(device_scan_u32(w.cscan, nullptr, w.cscan, (size_t)U + 1, false, w.tmp, nullptr, stream));
        GOF_HIP_CHECK(hipMemcpyAsync(&u32, w.cscan + U, 4, hipMemcpyDeviceToHost, stream));
        GOF_HIP_CHECK(hipStreamSynchronize(stream));
        E = u32;
    }
    cnt[1] = U; cnt[2] = E;
    GOF_HIP_CHECK(hipMemcpyAsync(w.counters, cnt, sizeof(cnt), hipMemcpyHostToDevice, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    *num_edges_host = E;
    *num_faces_host = F;
    return GOF_OK;
}

int gof_mtets_emit(int64_t V, int64_t Tt, const int64_t* tets, const float* vertices, const float* sdf, const float* scales,
                   const void* tet_ws, size_t tet_ws_bytes, const void* edge_ws, size_t edge_ws_bytes, int64_t num_edges, int64_t num_faces,
                   int64_t* edge_ids, float* edge_pos, float* edge_sdf, float* edge_scales, int64_t* faces, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    (void)V;
    if (Tt <= 0 || (num_edges == 0 && num_faces == 0)) return GOF_OK;
    if (!tets || !vertices || !sdf || !scales || !tet_ws || !edge_ws || !edge_ids || !edge_pos || !edge_sdf || !edge_scales || !faces) {
        set_error("mtets: NULL pointer"); return GOF_E_INVALID; }
    if (tet_ws_bytes < gof_mtets_tet_ws_bytes(Tt)) { set_error("mtets: tet workspace too small"); return GOF_E_WORKSPACE; }
    MtWs w;
    mt_tet_layout(Tt, reinterpret_cast<void*>(align_up(reinterpret_cast<size_t>(tet_ws))), &w);
    int64_t cnt[5];
    GOF_HIP_CHECK(hipMemcpyAsync(cnt, w.counters, sizeof(cnt), hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t U = cnt[1];
    if (cnt[2] != num_edges || cnt[3] + 2 * cnt[4] != num_faces) { set_error("mtets: counts do not match the workspace (run gof_mtets_count first)"); return GOF_E_INVALID; }
    if (cnt[0] > 0 && edge_ws_bytes < gof_mtets_edge_ws_bytes(cnt[0])) { set_error("mtets: edge workspace too small"); return GOF_E_WORKSPACE; }
    mt_edge_layout(cnt[0], reinterpret_cast<void*>(align_up(reinterpret_cast<size_t>(edge_ws))), &w);
    const dim3 blk(256);
    const auto grid = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    if (U > 0) {
        hipLaunchKernelGGL(mt_write_edges, grid(U), blk, 0, stream, U, w.ukeys, w.cscan, vertices, sdf, scales, edge_ids, edge_pos, edge_sdf, edge_scales);
        GOF_LAUNCH_CHECK(stream, 0);
    }
    const int64_t per_chunk = per_chunk_of(Tt);
    const int64_t nchunks = (Tt + per_chunk - 1) / per_chunk;
    if (nchunks > MT_MAX_CHUNKS) { set_error("mtets: too many 32 Mi-tet chunks (%lld)", (long long)nchunks); return GOF_E_INVALID; }
    const uint32_t nbp = (uint32_t)w.nbp;
    hipLaunchKernelGGL(mt_chunk_table, dim3((unsigned)nchunks + 1), blk, 0, stream, Tt, per_chunk, w.tetcase, w.bsum, nbp, w.chunk_tab);
    GOF_LAUNCH_CHECK(stream, 0);
    if (cnt[0] > 0)
        hipLaunchKernelGGL(mt_write_faces, grid(cnt[0]), blk, 0, stream, cnt[0], per_chunk, tets, w.tetcase, w.vt, w.frank, w.chunk_tab, U, w.ukeys, w.cscan, faces);
    GOF_LAUNCH_CHECK(stream, 0);
    return GOF_OK;
}

} // extern "C"
