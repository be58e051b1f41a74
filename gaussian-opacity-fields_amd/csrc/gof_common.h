// gof_common.h -- shared declarations of libgof_hip.so (gfx950 only).
//
// Workspace layouts, launch helpers and the small device math library used by every kernel.
// All kernels are compiled with -ffp-contract=off: every multiply-add that is fused is
// written as an explicit fmaf(), so the device evaluates exactly the fp32/fp64 operation
// sequence documented in DESIGN.md ("arithmetic contract").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gof_hip.h"

// One target: pop_tile reads HW_REG_XCC_ID, the backward used v_permlane16/32_swap, the LDS budgets assume 160 KB per CU.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libgof_hip.so is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace gof {

// ---- compile-time constants of the reference (config.h:15-17, auxiliary.h:21-34) -----------
constexpr int TILE_X = 16;
constexpr int TILE_Y = 16;
constexpr int TILE_PIX = TILE_X * TILE_Y;     // 256 threads = 4 wave64
constexpr int WAVE = 64;
constexpr int MAX_NUM_CONTRIBUTORS = 256;
constexpr int MAX_NUM_PROJECTED = 256;
#define GOF_NEAR_PLANE 0.2
#define GOF_FAR_PLANE 100.0

// ---- per-Gaussian record staged by the blend kernels ---------------------------------------
// One 64-byte line per Gaussian: everything the forward blend needs, so a tile-list entry is
// gathered with four 16-byte loads from one aligned line.
//   f[0..9]  view2gaussian (6 upper-triangle Sigma', 3 B, 1 C)      (forward.cu:268-277)
//   f[10]    opacity * low-pass coefficient  (conic_opacity.w)       (forward.cu:389)
//   f[11..13] rgb (SH colour, clamped at 0, or colors_precomp)       (forward.cu:378-381)
//   f[14..15] means2D (pixel coordinates)                             (forward.cu:387)
struct __attribute__((aligned(64))) SplatRec { float f[16]; };
constexpr int REC_W = 10, REC_RGB = 11, REC_XY = 14;

constexpr size_t ALIGN = 256;
__host__ __device__ inline size_t align_up(size_t x, size_t a = ALIGN) { return (x + a - 1) / a * a; }

// Geometry workspace (replaces GeometryState, rasterizer_impl.h:29-45)
struct GeomWs {
    float* depths;          // [P]
    SplatRec* rec;          // [P]
    float4* conic;          // [P] conic.xyz (2D inverse covariance), w unused -- backward only
    float4* bbox;           // [P] conservative pixel bounding box {xlo, xhi, ylo, yhi} of the alpha >= 1/255 footprint
    float4* fconic;         // [2P] footprint conic in ray space, unit-normalised: {m00, m01, m11, m02}, {m12, m22, q, -} (preprocess.hip)
    uint32_t* tiles_touched;// [P]
    uint2* rect;            // [P] tile rectangle {minx | miny << 16, w | h << 16} of a visible Gaussian, {0, 0} if culled
    uint8_t* clamped;       // [P] bit c set when colour channel c was clamped (forward.cu:67-69)
    uint32_t* flags;        // [4] device-side status words (prefilter violation, ...)
    // depth ordering of the Gaussians (binning.hip): keys = float bits of the view depth (0xFFFFFFFF if culled)
    uint32_t* dkey_a; uint32_t* dkey_b;   // [P]
    uint32_t* dval_a; uint32_t* dval_b;   // [P]  dval_a holds the depth-sorted Gaussian ids after the sort
    uint32_t* order_off;    // [P] exclusive scan of tiles_touched in depth order = first instance of the i-th sorted Gaussian
    uint32_t* inst_first;   // [P] = dkey_a, written by emit_instances: number of Gaussian id's first instance in emission (depth) order -- the backward's record numbering
    uint32_t* sort_tmp;     // radix / scan scratch (rs_tmp_words(P) + scan_tmp_words(P) words)
    uint32_t* total;        // [1] device copy of num_rendered
};
// Image workspace (replaces ImageState, rasterizer_impl.h:57-67)
struct ImageWs {
    uint2* ranges;          // [T]
    uint2* point_ranges;    // [T] integrate only
    float* final_T;         // [4][H*W]  T, dist1, dist2, distortion (forward.cu:591-594)
    uint32_t* n_contrib;    // [2][H*W]  last contributor, max contributor (forward.cu:596-597)
    uint32_t* tile_order;   // [8][tile_queue_stride(T)] dispatch order of the forward-side tile kernels: per XCD queue, by list length (binning.hip: order_tiles)
    uint32_t* tile_queue;   // [TILE_QUEUE_WORDS] [0..7] heads / [8..15] lengths of the forward-side queues, [32..47] the backward's lengths (its heads: scratch)
    uint32_t* tile_cost;    // [T] what blend_forward measured per tile (entries walked): the backward's cost
    uint32_t* tile_order_bw;// [8][tile_queue_stride(T)] dispatch order of blend_backward: by tile_cost (deepest walk first), written after the forward blend
    uint32_t* mask_cursors; // [POOL_SHARDS + 1] cursors of the contributor-mask pool's shards + the requests no shard could serve (cleared by the forward-side order_tiles)
};
constexpr int NXCD = 8;
constexpr int TILE_QUEUE_WORDS = 64;
constexpr int BW_STAGED_WORD = 16;      // queue[16] of an order_tiles call = sum over the tiles of min(cost, list length): for the backward's
                                        // order (tile_queue[32 + 16]) the entries the backward stages = partial records it writes (gof_backward_query)
// A pool with SHARDED cursors.  One cursor for a whole pool is one address for every allocating wave of the GPU: 67 k same-address
// atomics per frame cost blend_forward 0.46 ms at S1M (measured, round 4).  The pool is cut into POOL_SHARDS equal shards with a cursor
// each; a taker starts at the shard its tile number names and moves on while a shard is full.  cursors[POOL_SHARDS] counts the slots
// of requests that found no room anywhere.  (What the host learns: sum over the shards of min(cursor, shard size) + that word.)
constexpr uint32_t POOL_SHARDS = 64;
constexpr uint32_t POOL_NONE = 0xFFFFFFFFu;
__host__ __device__ inline uint32_t pool_shards(uint32_t cap) { return cap >= 16u * POOL_SHARDS ? POOL_SHARDS : 1u; }      // (a small pool is one shard)
__device__ __forceinline__ uint32_t pool_take(uint32_t* __restrict__ cursors, uint32_t cap, uint32_t n, uint32_t start)
{
    const uint32_t shards = pool_shards(cap);
    const uint32_t per = cap / shards;
    for (uint32_t k = 0; k < shards; k++) {
        const uint32_t s = (start + k) & (shards - 1);
        if (__hip_atomic_load(&cursors[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + n > per) continue;      // (a full shard keeps its cursor: look first)
        const uint32_t pos = atomicAdd(&cursors[s], n);
        if (pos + n <= per) return s * per + pos;
    }
    atomicAdd(&cursors[POOL_SHARDS], n);
    return POOL_NONE;
}

// Contributor masks of the forward / backward blend (round 4): one bit per (pixel of the tile, list position), set by blend_forward
// when that entry contributed to the pixel -- exactly the pairs the backward has to visit (backward.cu:763-805).  Rounds 1-3 reserved
// 32 B for EVERY instance (static layout, cmask_base below: still what the opacity-field query uses); the forward only visits ~40 % of
// the lists, so the words now live in a POOL of 8 KB chunks -- [wave 4][word 8][lane 64] = the tile's pixels x one staged batch of 256
// entries -- which a tile takes (pool_take, one atomic) as its blend reaches the batch.  table[slot0(tile) + batch] names the chunk;
// slot0 needs no scan: sum_{t' < t} ceil(len_t' / 256) <= ranges[t].x / 256 + t.  The pool's capacity is whatever the caller's
// binning workspace leaves behind the sort state (gof_binning_bytes_for; in sub-chunks of 2 KB = one wave's share of a chunk); a
// request beyond it is counted but not stored, and the caller learns from gof_backward_query (requested vs held) that this frame's
// forward has to be repeated with more room before its backward.
struct MaskPool {
    uint32_t* table;      // [mask_slots(R, T)]: first sub-chunk of (tile, batch), POOL_NONE if the pool was full
    uint32_t* pool;       // [cap][8][64]
    uint32_t cap;         // sub-chunks the pool holds
};
constexpr uint32_t MASK_SUBCHUNK_WORDS = 8u * 64u;
__host__ __device__ inline size_t mask_slots(size_t R, size_t ntiles) { return R / 256 + ntiles + 2; }
__device__ __forceinline__ size_t mask_slot0(uint32_t range_start, uint32_t tile) { return (size_t)(range_start >> 8) + tile; }

// Binning workspace (replaces BinningState, rasterizer_impl.h:69-79)
struct BinWs {
    uint32_t* vals;  uint32_t* vals_alt;          // [R]  vals = sorted point_list (Gaussian ids, per tile, front to back)
    uint32_t* tiles; uint32_t* tiles_alt;         // [R]  tiles = tile id of every sorted instance
    uint32_t* sort_tmp;                           // rs_tmp_words(R) words
    uint32_t* cmask;                              // [cmask_words(R, T)][256] contributor bit masks of the opacity-field query (static layout: integrate_pixels / integrate_points)
    MaskPool mp;                                  // contributor masks of the forward / backward blend (pool; aliases vals_alt / tiles_alt, dead after the tile sort)
    // query-point variant (integrate): per-point data gathered into LIST order, so the point pass streams it
    float2* pt_xy; float* pt_depth; float* pt_T; float* pt_acc;   // [NI]
    uint32_t* pt_order; uint32_t* pt_queue;                       // [T + 8], [TILE_QUEUE_WORDS]: dispatch order of integrate_points (pop_tile)
};
// Point workspace (replaces PointState, rasterizer_impl.h:47-55)
struct PointWs {
    float4* pos;            // [PN] {pixel x, pixel y, view depth, -} of a point inside the image: ONE 16-byte line for the random read of gather_sorted_points
    uint32_t* tiles_touched; uint32_t* point_offsets;   // offsets: inclusive scan
    float* T_state;         // [PN] running transmittance of each query point (integrate pass 2)
    uint32_t* scan_tmp;     // scan_tmp_words(PN) words
};

// camera matrices stay in device memory (no host copy, no sync); the addresses are wave-uniform so
// the compiler reads them with scalar loads.
struct Cam {
    const float* __restrict__ view;    // [16]
    const float* __restrict__ proj;    // [16]
    const float* __restrict__ campos;  // [3]
};

size_t geom_layout(int32_t P, void* base, GeomWs* out);
size_t image_layout(int32_t W, int32_t H, void* base, ImageWs* out);
enum BinMode { BIN_POINTS = 0, BIN_STATIC_MASKS = 1, BIN_MASK_POOL = 2 };
// BIN_MASK_POOL: `bytes` = size of the caller's buffer (the pool takes what is left; 0 = the full capacity, 4 * mask_slots sub-chunks)
size_t bin_layout(uint32_t R, int32_t W, int32_t H, void* base, BinWs* out, BinMode mode, size_t bytes = 0);
size_t point_layout(int32_t PN, void* base, PointWs* out);

// ---- error handling ----------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define GOF_HIP_CHECK(expr)                                                                   \
    do { hipError_t _e = (expr); if (_e != hipSuccess) {                                      \
        gof::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        return GOF_E_DEVICE; } } while (0)
// after a launch: always catch launch errors; in debug mode also synchronise (CHECK_CUDA semantics)
#define GOF_LAUNCH_CHECK(stream, debug)                                                       \
    do { GOF_HIP_CHECK(hipGetLastError());                                                    \
         if (debug) GOF_HIP_CHECK(hipStreamSynchronize(stream)); } while (0)

// ---- optional per-kernel timing (gof_profile_enable): HIP events recorded on the launch stream ------
struct ProfileScope {
    ProfileScope(const char* name, hipStream_t stream);
    ~ProfileScope();
    int slot;
    hipStream_t stream;
};
#define GOF_PROFILE(name, stream) gof::ProfileScope _gof_prof_scope_##__LINE__(name, stream)

// thread -> pixel map inside a 16x16 tile.  Each wave64 covers an 8x8 pixel quadrant and each 16-lane DPP row of it a 4x4
// pixel block (lane l: row r = l / 16 -> block (r % 2, r / 2) of the quadrant, i = l % 16 -> pixel (i % 4, i / 4) of the block):
// compact footprints meet the fewest splats -- the backward blend walks the contributor union of a ROW, the forward's wave
// exit needs all pixels of the quadrant saturated.  Per-pixel results do not depend on the map.
__device__ __forceinline__ void tile_pixel(uint32_t tid, uint32_t& lx, uint32_t& ly)
{
    const uint32_t wave = tid >> 6, row = (tid >> 4) & 3u, i = tid & 15u;
    lx = (i & 3u) + 4u * (row & 1u) + 8u * (wave & 1u);
    ly = (i >> 2) + 4u * (row >> 1) + 8u * (wave >> 1);
}

// inverse of tile_pixel: thread id of the pixel (lx, ly) of a tile
__device__ __forceinline__ uint32_t tile_thread(uint32_t lx, uint32_t ly)
{
    const uint32_t wave = (lx >> 3) + 2u * (ly >> 3), row = ((lx >> 2) & 1u) + 2u * ((ly >> 2) & 1u);
    return (wave << 6) + (row << 4) + ((ly & 3u) << 2) + (lx & 3u);
}

// Contributor masks, STATIC layout (the opacity-field query: integrate_pixels writes, integrate_points reads): for every pixel of a
// tile one bit per tile-list position.  Tile t's words start at cmask_base(ranges[t].x, t): since
// sum_{t' < t} ceil(len_t' / 32) <= ranges[t].x / 32 + t, the bases need no scan and never overlap.
// Layout [word][thread] (256 threads), so a wave reads / writes 256 contiguous bytes per word.
__host__ __device__ inline size_t cmask_words(size_t R, size_t ntiles) { return R / 32 + ntiles + 2; }
__device__ __forceinline__ size_t cmask_base(uint32_t range_start, uint32_t tile) { return (size_t)(range_start >> 5) + tile; }

// XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs (private L2s), so
// give each XCD a contiguous band of tiles -- neighbouring tiles gather the same splat records.
__device__ __forceinline__ uint32_t xcd_tile_id(uint32_t bid, uint32_t ntiles)
{
    const uint32_t per = (ntiles + NXCD - 1) / NXCD;
    const uint32_t t = (bid % NXCD) * per + bid / NXCD;
    return t;   // may be >= ntiles for the padded tail: caller checks
}
inline uint32_t xcd_padded_tiles(uint32_t ntiles) { return (ntiles + 7) / 8 * 8; }
// Tile assignment for scenes whose tiles do NOT cost the same (round 3; binning.hip: order_tiles).  The tiles are classified by
// cost (half-octave buckets) and every XCD is given the same share of every cost class -- equal tile counts (the dispatcher hands
// each XCD #tiles / 8 workgroups) and equal work -- as spatially contiguous runs of tiles (L2 locality), its queue sorted heaviest
// class first.  A workgroup pops the head of its XCD's queue as it STARTS (one atomic): the i-th workgroup to start on an XCD
// renders that XCD's i-th heaviest tile whatever its blockIdx, so a heavy tile never begins last and the light ones fill the tail.
// (One workgroup per tile, not persistent ones: the SIMD arbiter serves the OLDEST wave first, so the youngest of a set of
// persistent workgroups crawls through its first -- heaviest -- tile until the old ones retire; measured, the first tiles popped
// took the whole kernel.  Contiguous bands of equal tile COUNT per XCD with stealing do not balance either: an XCD owns exactly
// #tiles / 8 workgroups, so none is left to help a heavy band.)  Exactly ntiles pops succeed over a grid of >= ntiles workgroups
// (a workgroup whose own queue is empty tries the others: queue lengths may differ from the workgroup supply by a few tiles), so
// every tile is rendered exactly once; results do not depend on the assignment.  The XCD is read from the hardware
// (HW_REG_XCC_ID; blockIdx % 8 observed equivalent): affinity for speed only.  Must be called by all threads of the workgroup
// (one barrier).  0xFFFFFFFF = no tile left.
// heads[0..7] / lens[0..7] (order_tiles writes queue[0..7] = 0 and queue[8..15] = lengths; the backward keeps its heads in its own
// scratch, cleared per call), order[x * tile_queue_stride(ntiles) + i] = i-th tile of XCD x.
__host__ __device__ __forceinline__ uint32_t tile_queue_stride(uint32_t ntiles) { return (ntiles + NXCD - 1) / NXCD + 128u; }   // + one per cost bucket (rounding of the shares)
__device__ __forceinline__ uint32_t pop_tile(const uint32_t* __restrict__ order, uint32_t* __restrict__ heads, const uint32_t* __restrict__ lens,
                                             uint32_t ntiles, uint32_t* s_slot)
{
#ifdef GOF_STATIC_TILES       // developer A/B (tests/devtools/dev_tile_schedule.py): round 2's static map (contiguous band per XCD, ascending)
    return xcd_tile_id(blockIdx.x, ntiles);
#endif
    if (threadIdx.x == 0) {
        const uint32_t stride = tile_queue_stride(ntiles);
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3u << 11) | (0u << 6) | 20u) & (NXCD - 1);      // hwreg(HW_REG_XCC_ID, 0, 4)
        uint32_t tile = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < NXCD; k++) {
            const uint32_t x = (xcc + k) & (NXCD - 1);
            const uint32_t n = lens[x];
            // (an exhausted head keeps growing: compare first so that the common case costs one atomic)
            if (__hip_atomic_load(&heads[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n) continue;
            const uint32_t pos = atomicAdd(&heads[x], 1u);
            if (pos < n) { tile = order[x * stride + pos]; break; }
        }
        *s_slot = tile;
    }
    __syncthreads();
    return *s_slot;
}

// ---- device math ---------------------------------------------------------------------------------
// Deterministic fp32 exp (Cephes scheme): only IEEE mul / fma / rint / ldexp, <= 1 ulp on
// [-87, 88].  The oracle evaluates the identical sequence, which makes per-pair alpha
// bit-identical between host and device.
// NONPOS: the caller guarantees x <= 0 or NaN (the blend's power, clamped at 0 in front of the call): the upper clamp can never
// fire and is left out (a compare and a select per pair); every result bit is that of the general form.
template <bool NONPOS = false>
__device__ __forceinline__ float gexpf(float x)
{
    x = (x < -87.0f) ? -87.0f : x;     // written as compares so NaN propagates exactly as on the host
    if (!NONPOS) x = (x > 88.0f) ? 88.0f : x;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
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

// Packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE fp32 operations per lane and instruction, the only way to the
// vector unit's peak rate).  Written with vector types so the pairing is by construction, not left to the SLP vectoriser (which
// is switched off for the library: its shuffles cost more than it gains).  Element-wise results are bit-identical to the scalar
// operations; a + b * c is NOT fused unless pk_fma is written.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{ -a.x, -a.y, -a.z }; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return V3{ s * a.x, s * a.y, s * a.z }; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return V3{ a.x * s, a.y * s, a.z * s }; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return V3{ a.x / s, a.y / s, a.z / s }; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z; return tx + ty + tz; }

// column-major 3x3 / 4x4 with the product order used by the reference's matrix library:
// r[c][row] = (a[0][row]*b[c][0] + a[1][row]*b[c][1]) + a[2][row]*b[c][2]
struct M3 { float m[3][3]; };
struct M4 { float m[4][4]; };
__device__ __forceinline__ M3 mk3(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2)
{
    M3 r; r.m[0][0] = x0; r.m[0][1] = y0; r.m[0][2] = z0; r.m[1][0] = x1; r.m[1][1] = y1; r.m[1][2] = z1; r.m[2][0] = x2; r.m[2][1] = y2; r.m[2][2] = z2; return r;
}
__device__ __forceinline__ M3 mul(const M3& a, const M3& b)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int row = 0; row < 3; row++)
            r.m[c][row] = a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1] + a.m[2][row] * b.m[c][2];
    return r;
}
__device__ __forceinline__ V3 mul(const M3& m, V3 v)   // m * v
{
    return V3{ m.m[0][0] * v.x + m.m[1][0] * v.y + m.m[2][0] * v.z,
               m.m[0][1] * v.x + m.m[1][1] * v.y + m.m[2][1] * v.z,
               m.m[0][2] * v.x + m.m[1][2] * v.y + m.m[2][2] * v.z };
}
__device__ __forceinline__ V3 mul(V3 v, const M3& m)   // v * m
{
    return V3{ m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z,
               m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
               m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z };
}
__device__ __forceinline__ M4 mul(const M4& a, const M4& b)
{
    M4 r;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int row = 0; row < 4; row++)
            r.m[c][row] = a.m[0][row] * b.m[c][0] + a.m[1][row] * b.m[c][1] + a.m[2][row] * b.m[c][2] + a.m[3][row] * b.m[c][3];
    return r;
}
__device__ __forceinline__ M3 transpose(const M3& m)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int row = 0; row < 3; row++) r.m[c][row] = m.m[row][c];
    return r;
}
__device__ __forceinline__ M4 transpose(const M4& m)
{
    M4 r;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int row = 0; row < 4; row++) r.m[c][row] = m.m[row][c];
    return r;
}
__device__ __forceinline__ M3 neg(const M3& m)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int row = 0; row < 3; row++) r.m[c][row] = -m.m[c][row];
    return r;
}
__device__ __forceinline__ M3 add(const M3& a, const M3& b)
{
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int row = 0; row < 3; row++) r.m[c][row] = a.m[c][row] + b.m[c][row];
    return r;
}
__device__ __forceinline__ M3 outer(V3 c, V3 r)
{
    M3 m;
    const float rr[3] = { r.x, r.y, r.z };
#pragma unroll
    for (int i = 0; i < 3; i++) { m.m[i][0] = c.x * rr[i]; m.m[i][1] = c.y * rr[i]; m.m[i][2] = c.z * rr[i]; }
    return m;
}

// rotation matrix from a quaternion used as given (forward.cu:138-149)
__device__ __forceinline__ M3 quat_to_R(float r, float x, float y, float z)
{
    return mk3(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}

// tile rectangle of a splat (auxiliary.h:64-74)
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, uint32_t& minx, uint32_t& miny, uint32_t& maxx, uint32_t& maxy, uint32_t gx, uint32_t gy)
{
    minx = min(gx, (uint32_t)max(0, (int)((px - max_radius) / TILE_X)));
    miny = min(gy, (uint32_t)max(0, (int)((py - max_radius) / TILE_Y)));
    maxx = min(gx, (uint32_t)max(0, (int)((px + max_radius + TILE_X - 1) / TILE_X)));
    maxy = min(gy, (uint32_t)max(0, (int)((py + max_radius + TILE_Y - 1) / TILE_Y)));
}

__device__ __forceinline__ V3 transform_point_4x3(V3 p, const float* __restrict__ m)
{
    return V3{ m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
               m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
               m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] };
}

// Everything that depends on a (pixel ray, splat) pair: forward.cu:499-533, backward.cu:771-804.
struct PairEval {
    float n0, n1, n2;      // un-normalised view-space normal Sigma' * ray
    float AAf, BBf;        // fp32 values the reference widens to double (forward.cu:511-512)
    double q;              // BB / AA (fp64, correctly rounded)
    float qf;              // the same quotient in fp32 (backward)
    float t, G, alpha;
    bool skip;
};
// fp32 prelude: normal, AA, BB in the reference's operation order (no contraction)
__device__ __forceinline__ void pair_prelude(const float* __restrict__ v, float rx, float ry, PairEval& p)
{
    p.n0 = v[0] * rx + v[1] * ry + v[2];
    p.n1 = v[1] * rx + v[3] * ry + v[4];
    p.n2 = v[2] * rx + v[4] * ry + v[5];
    p.AAf = rx * p.n0 + ry * p.n1 + p.n2;
    p.BBf = 2 * (v[6] * rx + v[7] * ry + v[8]);
}
// Conservative cull: true only if alpha = min(0.99, w * exp(power)) is CERTAINLY < 1/255, i.e. the
// exact path would `continue` at forward.cu:534 / backward.cu:803.  power is estimated in fp32; the
// bound E covers the fp32 rounding of BB^2/(4 AA) and of the subtraction from CC (a few ulp of the
// larger operand) with a wide safety factor, `log_thr` = ln(1/(255 w)) is computed once per staged
// entry.  Pairs that are not certainly rejected take the exact fp64 path, so results are unchanged.
__device__ __forceinline__ bool pair_certainly_transparent(const PairEval& p, float CC, float log_thr)
{
    if (!(p.AAf > 0.0f)) return false;                       // degenerate quadric: let the exact path decide
    const float qf = p.BBf * __builtin_amdgcn_rcpf(p.AAf);
    const float X = qf * (p.BBf * 0.25f);
    const float power_est = -0.5f * (CC - X);
    const float E = 4e-7f * fmaxf(fabsf(CC), fabsf(X)) + 1e-4f;
    return power_est + E < log_thr;
}
__device__ __forceinline__ float cull_log_threshold(float w)
{
    // ln(1 / (255 w)) with a small downward margin; w <= 0 (or NaN) can never reach 1/255
    return (w > 0.0f) ? -__logf(255.0f * w) - 1e-3f : __builtin_huge_valf();
}
// exact remainder (after pair_prelude): t, min_value, power, exp, alpha
// HW_EXP: exp through v_exp_f32 (<= 1 ulp) instead of the shared deterministic gexpf -- only where alpha is NOT compared with a
// threshold (the backward: its contributors come from the forward's masks), never in the forward / integrate.
template <bool HW_EXP = false>
__device__ __forceinline__ void pair_exact_cc(float CC, float w, PairEval& p);
template <bool HW_EXP = false>
__device__ __forceinline__ void pair_exact(const float* __restrict__ v, float w, PairEval& p) { pair_exact_cc<HW_EXP>(v[9], w, p); }
template <bool HW_EXP>
__device__ __forceinline__ void pair_exact_cc(float CC, float w, PairEval& p)
{
    const double AA = (double)p.AAf, BB = (double)p.BBf;
    // -BB/(2*AA) == -(BB/AA)/2 exactly (power-of-two scaling commutes with rounding), so one
    // fp64 division serves both t and min_value.
    const double q = BB / AA;
    p.q = q;
    p.t = (float)(-q * 0.5);
    p.skip = ((double)p.t <= GOF_NEAR_PLANE);
    const double min_value = (-q) * (BB * 0.25) + (double)CC;
    float power = (float)(-0.5 * min_value);
    if (power > 0.0f) power = 0.0f;
    p.G = HW_EXP ? __builtin_amdgcn_exp2f(power * 1.44269504088896341f) : gexpf<true>(power);
    p.alpha = fminf(0.99f, w * p.G);
    if (p.alpha < 1.0f / 255.0f) p.skip = true;
}
// The same quantities for the BACKWARD (no threshold decision depends on them there): the quotient BB/AA as an fp32 hi + lo
// pair (reciprocal + FMA remainders), the product with BB/4 with its FMA error term, and the cancelling difference CC - product
// taken hi first (exact by Sterbenz for the pairs that matter: min_value << CC).  min_value carries ~1e-6 absolute error -- the
// fp64 path's value to ~5e-7 relative in G -- for 13 fp32 instructions instead of ~23 fp64 ones (incl. a quarter-rate division).
__device__ __forceinline__ void pair_exact_backward_cc(float CC, float w, PairEval& p)
{
    const float ra = __builtin_amdgcn_rcpf(p.AAf);
    float qh = p.BBf * ra;
    qh = fmaf(fmaf(-qh, p.AAf, p.BBf), ra, qh);                 // refined quotient (within 1 ulp)
    const float ql = fmaf(-qh, p.AAf, p.BBf) * ra;               // its remainder: qh + ql = BB/AA to ~2^-45
    const float b4 = p.BBf * 0.25f;
    const float ph = qh * b4;
    const float pl = fmaf(ql, b4, fmaf(qh, b4, -ph));            // (qh + ql) * b4 = ph + pl
    const float min_value = (CC - ph) - pl;
    p.qf = qh;
    p.t = -0.5f * qh;
    float power = -0.5f * min_value;
    if (power > 0.0f) power = 0.0f;
    p.G = __builtin_amdgcn_exp2f(power * 1.44269504088896341f);
    p.alpha = fminf(0.99f, w * p.G);
    p.skip = false;
}
// The FORWARD's default value path (round 4): the exact path's own fp64 arithmetic with the one expensive instruction sequence taken
// out -- the fp64 DIVISION BB / AA (v_rcp_f64 at quarter rate + two Newton steps on the reciprocal + the scaling / fix-up
// instructions: ~14 of the ~37 fp64 instructions of a pair, and a second one for the mapped depth).  The quotient is formed from
// the fp32 reciprocal instead: q0 = fl32(BB * rcp(AA)) (2^-22), then two corrections q <- q + (BB - q AA) y in fp64 with
// y = (double)rcp(AA): the error contracts by |AA y - 1| <= 2^-23 each time, 2^-45 then 2^-68, i.e. q is the FAITHFULLY rounded
// fp64 quotient -- the correctly rounded one except where the true quotient lies within 2^-68 |q| of a rounding boundary (a fraction
// 2^-14 of the pairs), and then its neighbour.  Everything behind it is the exact path's code.  What that last bit of q can change:
// power = fl32(-min_value / 2) moves by one fp32 ulp only if min_value also lies within 2^-52 |BB^2 / 4 AA| of an fp32 rounding
// boundary -- 2^-14 x ~1e-3 at S1M's conditioning, a handful of pairs per 1.2e8; measured: 0 differing n_contrib / contributor
// masks / colour bits on every scene of the test table and at full size (tests/: both modes are run side by side).
// A quotient the fp32 reciprocal cannot start (AA zero / denormal, an overflowing quotient, NaN) takes the true division: the
// degenerate cases keep the reference's IEEE semantics.
__device__ __forceinline__ void pair_nodiv_cc(float CC, float w, PairEval& p)
{
    const float ra = __builtin_amdgcn_rcpf(p.AAf);
    const float qh = p.BBf * ra;
    const double AA = (double)p.AAf, BB = (double)p.BBf;
    double q;
    if (fabsf(qh) < 3e38f) {        // (false for inf / NaN too; a reciprocal that is inf / NaN makes the quotient inf / NaN: no test of its own)
        const double y = (double)ra;
        q = (double)qh;
        q = fma(fma(-q, AA, BB), y, q);
        q = fma(fma(-q, AA, BB), y, q);
    } else {
        q = BB / AA;
    }
    p.q = q;
    p.t = -0.5f * (float)q;                               // == (float)(-q * 0.5): the scaling is exact
    p.skip = p.t < 0.2f;                                  // (double)t <= 0.2  <=>  t < 0.2f  (0.2f is the fp32 above 0.2; NaN: neither)
    // min_value = fl(fl((-q) (BB / 4)) + CC) of the exact path, with its power-of-two scalings moved to where they cost nothing:
    // fl((-q) (BB / 4)) = -fl(q BB) / 4 exactly, and adding an exactly scaled product is what ONE fma with the factor -1/4 does
    // (its single rounding is the addition's); likewise fl32(-min_value / 2) = -fl32(min_value) / 2.  Same bits, an fp64 scaling
    // and an fp64 multiply less per pair.
    const double min_value = fma(q * BB, -0.25, (double)CC);
    float power = -0.5f * (float)min_value;
    if (power > 0.0f) power = 0.0f;
    p.G = gexpf<true>(power);
    p.alpha = fminf(0.99f, w * p.G);
    if (p.alpha < 1.0f / 255.0f) p.skip = true;
}
// mapped depth (forward.cu:545) m = (FAR t - FAR NEAR) / ((FAR - NEAR) t) = c1 - c2 / t in fp32 (v_rcp_f32, one FMA): within
// 1.2e-7 c2 / t + 1 ulp of the exact path's fp64 quotient, i.e. <= 2 ulp at the near plane, 1 ulp from t ~ 1 on (no decision
// depends on it; the backward evaluates the same form)
__device__ __forceinline__ float mapped_depth_fast(float t)
{
    constexpr float c1 = (float)(GOF_FAR_PLANE / (GOF_FAR_PLANE - GOF_NEAR_PLANE));
    constexpr float c2 = (float)(GOF_FAR_PLANE * GOF_NEAR_PLANE / (GOF_FAR_PLANE - GOF_NEAR_PLANE));
    return fmaf(-c2, __builtin_amdgcn_rcpf(t), c1);
}
__device__ __forceinline__ void eval_pair(const float* __restrict__ v, float w, float rx, float ry, PairEval& p)
{
    pair_prelude(v, rx, ry, p);
    pair_exact(v, w, p);
}

} // namespace gof
