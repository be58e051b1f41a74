// api.hip -- host orchestration and the extern "C" boundary of libgof_hip.so (include/gof_hip.h).
//
// Replaces CudaRasterizer::Rasterizer::{forward,backward,integrate,markVisible}
// (reference rasterizer_impl.cu:247-405, 409-526, 530-792, 174-186) and the torch binding's
// buffer management (reference rasterize_points.cu:28-122).  No torch types, no global state,
// every launch on the caller's stream.
#include "gof_common.h"
#include "gof_status.h"
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace gof {

// ---- kernels / helpers defined in the other translation units -------------------------------------

template <int STAGE, int FOOT>
__global__ void preprocess_fwd(int P, int D, int M, const float* means3D, const float* scales, float scale_modifier,
                               const float* rotations, const float* opacities, const float* shs, const float* shs_rest, const float* cov3D_precomp,
                               const float* colors_precomp, const float* v2g_precomp, Cam cam, int W, int H, float tan_fovx,
                               float tan_fovy, float focal_x, float focal_y, float kernel_size, uint32_t gx, uint32_t gy,
                               int prefiltered, int32_t* radii, float* depths, SplatRec* rec, float4* conic_out, float4* bbox_out, float4* fconic_out,
                               uint32_t* tiles_touched, uint2* rect_out, uint8_t* clamped, uint32_t* depth_key, uint32_t* depth_val, uint32_t* flags,
                               uint32_t* zero_ptr, uint32_t zero_n);
template <int MODE>
__global__ void preprocess_bwd(int P, int D, int M, const float* means3D, const int32_t* radii, const float* shs, const float* shs_rest,
                               const uint8_t* clamped, const float* scales, const float* rotations, Cam cam,
                               const float* dL_dv2g, const float* dL_dcolor, float* dL_dmeans, float* dL_dsh, float* dL_dsh_rest,
                               float* dL_dscales, float* dL_drots);
__global__ void preprocess_points(int PN, const float* points3D, Cam cam, int W, int H, float focal_x, float focal_y,
                                  float4* pos, uint32_t* tiles_touched);
__global__ void mark_visible_kernel(int P, const float* means3D, Cam cam, uint8_t* present);
__global__ void sh_grad_pack(int P, const float* dL_dcolor, const uint8_t* clamped, const int32_t* radii, float* packed);
template <int MC>
__global__ void sh_grad_expand(int P, int D, int M, int n_views, const float* means3D, const float* campos, long campos_stride, const float* packed,
                               long packed_stride, float scale, float* out_dc, long stride_dc, float* out_rest, long stride_rest);

uint32_t higher_msb(uint32_t n);
size_t scan_tmp_words(size_t n);
hipError_t device_scan_u32(const uint32_t* in, const uint32_t* idx, uint32_t* out, size_t n, bool inclusive, uint32_t* tmp,
                           const uint32_t** total_dev_out, hipStream_t stream);
size_t rs_tmp_words(size_t n);
const uint32_t* radix_sort_error_flag(const uint32_t* tmp, size_t n, int end_bit);
hipError_t radix_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int end_bit,
                                uint32_t* tmp, uint32_t** keys_res, uint32_t** vals_res, hipStream_t stream, const uint32_t* n_dev = nullptr);
hipError_t radix_sort_pairs_u32_z(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int end_bit,
                                  uint32_t* tmp, uint32_t** keys_res, uint32_t** vals_res, hipStream_t stream, const uint32_t* n_dev,
                                  size_t zero_words_behind, bool first_hist_done, bool scratch_zeroed = false);
size_t radix_zero_words(size_t n, int end_bit);
uint32_t* radix_classic_hist(uint32_t* tmp, size_t n, int end_bit);
uint32_t rs_block_items();
uint32_t rs_units(size_t n);
int radix_passes(int end_bit);
uint32_t emit_instances_grid(uint32_t slots, int P);
__global__ void emit_instances(int P, const uint32_t* order, const uint32_t* order_off, const uint32_t* minxy_sorted, const uint32_t* wh_sorted,
                               uint32_t* tiles, uint32_t* gids, uint32_t gx, uint32_t capacity, uint32_t* inst_first, uint32_t* hist0, uint32_t hist_stride);
uint32_t emit_block_slots();
__global__ void point_keys(int PN, const float4* pos, const uint32_t* offsets, const uint32_t* tiles_touched, uint32_t* keys, uint32_t* vals,
                           uint32_t gx, uint32_t gy);
__global__ void tile_ranges(uint32_t L, const uint32_t* tiles, uint2* ranges, int shift, const uint32_t* n_dev, const uint32_t* sort_error,
                            uint32_t* async_status);
__global__ void rebuild_keys(uint32_t R, const uint32_t* tiles, const uint32_t* gids, const float* depths, uint64_t* keys);

__global__ void blend_forward(const uint2* ranges, const uint32_t* point_list, const SplatRec* rec, const float4* fconic, int W, int H,
                              float focal_x, float focal_y, const float* bg_color, float* final_T, uint32_t* n_contrib,
                              float* out_color, MaskPool masks, uint32_t* mask_cursors, uint32_t gx, uint32_t ntiles, const uint32_t* tile_order, uint32_t* tile_queue,
                              uint32_t* tile_cost);
__global__ void blend_forward_exact(const uint2* ranges, const uint32_t* point_list, const SplatRec* rec, const float4* fconic, int W, int H,
                                    float focal_x, float focal_y, const float* bg_color, float* final_T, uint32_t* n_contrib,
                                    float* out_color, MaskPool masks, uint32_t* mask_cursors, uint32_t gx, uint32_t ntiles, const uint32_t* tile_order, uint32_t* tile_queue,
                                    uint32_t* tile_cost);
__global__ void order_tiles(uint32_t ntiles, const uint2* ranges, const uint32_t* cost_in, uint32_t* order, uint32_t* queue, const uint2* times_ranges,
                            uint32_t* clear_cursors, uint32_t* staged_out, const uint32_t* cursors_in, uint32_t* usage_host);
__global__ void blend_backward(const uint2* ranges, const uint32_t* point_list, const SplatRec* rec, const float4* conic, MaskPool masks,
                               int W, int H, float focal_x, float focal_y, const float* bg_color, const float* final_Ts,
                               const uint32_t* n_contrib, const float* dL_dpixels, const uint2* rect, const uint32_t* inst_off,
                               float4* part16, uint32_t* slot_of, uint32_t* rec_cursor, uint32_t rec_cap,
                               uint32_t gx, uint32_t ntiles, const uint32_t* tile_order,
                               uint32_t* tile_queue, const uint32_t* tile_lens);
__global__ void gather_tile_partials(int P, const uint32_t* inst_off, const uint32_t* tiles_touched, const float4* part16, const float4* conic_w,
                                     const uint32_t* slot_of, float* dL_dmean2D, float* dL_dopacity, float* dL_dcolors, float* dL_dv2g);
__global__ void integrate_pixels(const uint2* gaussian_ranges, const uint32_t* gaussian_list, const SplatRec* rec, const float4* bbox,
                                 const float4* fconic, int W, int H, float focal_x, float focal_y, const float* bg_color, float* final_T,
                                 uint32_t* n_contrib, float* out_color, uint32_t* cmask, uint32_t gx, uint32_t ntiles, const uint32_t* tile_order,
                                 uint32_t* tile_queue, uint32_t* tile_cost);
__global__ void integrate_pixels_capped(const uint2* gaussian_ranges, const uint32_t* gaussian_list, const SplatRec* rec, const float4* bbox,
                                        const float4* fconic, int W, int H, float focal_x, float focal_y, const float* bg_color, float* final_T,
                                        uint32_t* n_contrib, float* out_color, uint32_t* cmask, uint32_t gx, uint32_t ntiles, uint32_t* tile_cost);
__global__ void integrate_rays(const uint2* gaussian_ranges, const uint32_t* gaussian_list, const SplatRec* rec,
                               const float4* fconic, int W, int H, float focal_x, float focal_y, const float* bg_color, float* final_T,
                               uint32_t* n_contrib, float* out_color, uint32_t* cmask, uint32_t gx, uint32_t ntiles, const uint32_t* tile_order,
                               uint32_t* tile_queue, uint32_t* tile_cost);
__global__ void integrate_points(const uint2* gaussian_ranges, const uint2* point_ranges, const uint32_t* gaussian_list,
                                 const uint32_t* point_list, const SplatRec* rec, const float* zfront, int zstride, const uint32_t* cmask, int W, int H,
                                 const uint32_t* pt_key, const float2* pt_ray, const float* pt_depth, float* pt_T, float* pt_acc, const float* base_color, float* out_color,
                                 float* out_alpha_integrated, float* out_color_integrated, const uint32_t* n_contrib, int acc_min, uint32_t gx, uint32_t ntiles,
                                 const uint32_t* tile_order, uint32_t* tile_queue);
__global__ void pack_view_geometry(int P, const SplatRec* rec, const float4* fconic, SplatRec* rec_out, float* zfront_out);
uint32_t gather_scan_tiles(size_t n);
uint32_t gather_scan_threads();
size_t gather_scan_state_words(size_t n);
__global__ void gather_scan_rects(uint32_t n, const uint2* rect, const uint32_t* order, const uint32_t* keys_sorted, uint32_t* minxy_sorted, uint32_t* wh_sorted,
                                  uint32_t* order_off, const uint32_t* sort_error, uint2* ranges, uint32_t ntiles, uint32_t* state, uint32_t* total_host);
__global__ void gather_sorted_points(uint32_t NI, const uint32_t* sorted_ids, const float4* pos, float2* pt_ray, float* pt_depth, int W, int H,
                                     float focal_x, float focal_y);

// ---- error text --------------------------------------------------------------------------------------
static thread_local std::string g_error;
void set_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
}

// ---- asynchronous device status ----------------------------------------------------------------------
// One host-mapped word per DEVICE (pinned, in one allocation visible to every device): a kernel that detects a failure nobody reads
// back in the same call -- the tile sort's bounded look-back poll on lists of <= 2M instances, which has no host read-back behind it --
// writes it through the mapping, and the NEXT forward / backward / integrate call ON THAT DEVICE returns GOF_E_DEVICE (the convention
// of asynchronous errors in the reference's runtime: they surface at a later call).  Written only on failure, so it costs nothing
// otherwise.  Beyond this word (and the thread-local error text, the per-thread second stream) the library keeps no state.
namespace {
constexpr int MAX_DEVICES = 16;
std::mutex g_status_mutex;
volatile uint32_t* g_status_host = nullptr;      // [MAX_DEVICES][16]: a 64-byte line per device
uint32_t* g_status_dev = nullptr;
int current_device_slot()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return (dev >= 0 && dev < MAX_DEVICES) ? dev : 0;
}
}
static uint32_t* async_status_word()
{
    std::lock_guard<std::mutex> lk(g_status_mutex);
    if (!g_status_host) {
        void* p = nullptr;
        if (hipHostMalloc(&p, MAX_DEVICES * 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        std::memset(p, 0, MAX_DEVICES * 64);
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(p); return nullptr; }
        g_status_host = static_cast<volatile uint32_t*>(p);
        g_status_dev = static_cast<uint32_t*>(d);
    }
    return g_status_dev + 16 * current_device_slot();
}
// Device-visible address of `host_words` if that is pinned, device-mapped host memory (hipHostMalloc -- what torch's pin_memory() uses),
// else nullptr: kernels then store a few result words there themselves, where a hipMemcpyAsync would put one or two blit kernels
// (4-5 us each) on the stream.
static uint32_t* device_view_of_pinned(uint32_t* host_words)
{
    if (!host_words) return nullptr;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, host_words, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return static_cast<uint32_t*>(d);
}
static int take_async_status()
{
    std::lock_guard<std::mutex> lk(g_status_mutex);
    if (!g_status_host) return GOF_OK;
    volatile uint32_t* w = g_status_host + 16 * current_device_slot();
    if (*w) {
        *w = 0;
        set_error("an EARLIER call's tile sort on this device timed out waiting for a predecessor block (GPU heavily oversubscribed?): that frame was rendered as background");
        return GOF_E_DEVICE;
    }
    return GOF_OK;
}

// ---- profiling ---------------------------------------------------------------------------------------
namespace {
struct ProfRec { const char* name; hipEvent_t e0, e1; };
std::mutex g_prof_mutex;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
}
ProfileScope::ProfileScope(const char* name, hipStream_t s) : slot(-1), stream(s)
{
    if (!g_prof_on) return;
    ProfRec r; r.name = name;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, stream);      // profiling is best effort: a failed record shows up as a missing sample
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    g_prof.push_back(r);
    slot = (int)g_prof.size() - 1;
}
ProfileScope::~ProfileScope()
{
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    if (slot < (int)g_prof.size()) (void)hipEventRecord(g_prof[slot].e1, stream);
}

// ---- workspace layouts -------------------------------------------------------------------------------
template <typename T>
static inline void carve(char*& p, T*& ptr, size_t count)
{
    p = reinterpret_cast<char*>(align_up(reinterpret_cast<size_t>(p)));
    ptr = reinterpret_cast<T*>(p);
    p += count * sizeof(T);
}

static size_t geom_layout_ex(int32_t P, void* base, GeomWs* out, size_t* forward_bytes_out)
{
    GeomWs g;
    char* p = static_cast<char*>(base);
    const size_t n = (size_t)P;
    carve(p, g.rec, n);
    carve(p, g.conic, n);
    carve(p, g.fconic, 2 * n);
    carve(p, g.depths, n);
    carve(p, g.tiles_touched, n);
    carve(p, g.rect, n);
    carve(p, g.clamped, n);
    carve(p, g.flags, 4);
    carve(p, g.total, 4);
    carve(p, g.dval_a, n);          // depth-sorted Gaussian ids (kept for the render stage)
    carve(p, g.order_off, n);
    carve(p, g.dkey_a, n);
    g.inst_first = g.dkey_a;        // (the sorted depth keys are dead once the depth sort has delivered the order: emit_instances writes over them)
    carve(p, g.dkey_b, n);
    carve(p, g.dval_b, n);
    carve(p, g.sort_tmp, rs_tmp_words(n) + gather_scan_state_words(n));
    // LAST: what only the opacity-field query reads (the footprints' pixel boxes, 16 B per Gaussian) -- a workspace for a forward /
    // backward pair may end here (gof_geom_bytes_forward: round 6), the layout in front of it is the same either way
    const size_t forward_bytes = (size_t)(p - static_cast<char*>(base)) + ALIGN;
    carve(p, g.bbox, n);
    if (out) *out = g;
    if (forward_bytes_out) *forward_bytes_out = forward_bytes;
    return (size_t)(p - static_cast<char*>(base)) + ALIGN;
}
size_t geom_layout(int32_t P, void* base, GeomWs* out) { return geom_layout_ex(P, base, out, nullptr); }
size_t image_layout(int32_t W, int32_t H, void* base, ImageWs* out)
{
    ImageWs im;
    char* p = static_cast<char*>(base);
    const size_t N = (size_t)W * H;
    const size_t T = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    carve(p, im.ranges, T);
    carve(p, im.point_ranges, T);
    carve(p, im.final_T, 4 * N);
    carve(p, im.n_contrib, 2 * N);
    carve(p, im.tile_order, (size_t)NXCD * tile_queue_stride((uint32_t)T));
    carve(p, im.tile_queue, (size_t)TILE_QUEUE_WORDS);
    carve(p, im.tile_cost, T);
    carve(p, im.tile_order_bw, (size_t)NXCD * tile_queue_stride((uint32_t)T));
    carve(p, im.mask_cursors, (size_t)POOL_SHARDS + 2);      // [POOL_SHARDS + 1]: the staged-entry sum of the backward's order (one copy brings all of it to the host)
    if (out) *out = im;
    return (size_t)(p - static_cast<char*>(base)) + ALIGN;
}
size_t bin_layout(uint32_t R, int32_t W, int32_t H, void* base, BinWs* out, BinMode mode, size_t bytes)
{
    BinWs b;
    char* p = static_cast<char*>(base);
    const size_t n = (size_t)R;
    const size_t T = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    carve(p, b.vals, n);            // sorted point_list first: the only part the backward reads
    carve(p, b.tiles, n);
    b.cmask = nullptr; b.pt_xy = nullptr; b.pt_depth = nullptr; b.pt_T = nullptr; b.pt_acc = nullptr; b.pt_order = nullptr; b.pt_queue = nullptr;
    b.mp.table = nullptr; b.mp.pool = nullptr; b.mp.cap = 0;
    if (mode == BIN_MASK_POOL) {
        // [table][sort scratch][vals_alt][tiles_alt] and the pool ON TOP of the two ping-pong buffers: they are dead once the tile
        // sort has left its result in vals / tiles, i.e. before blend_forward writes the first mask word
        carve(p, b.mp.table, mask_slots(n, T));
        carve(p, b.sort_tmp, rs_tmp_words(n));
        char* const pool0 = reinterpret_cast<char*>(align_up(reinterpret_cast<size_t>(p)));
        carve(p, b.vals_alt, n);
        carve(p, b.tiles_alt, n);
        b.mp.pool = reinterpret_cast<uint32_t*>(pool0);
        const size_t sub = (size_t)MASK_SUBCHUNK_WORDS * sizeof(uint32_t);
        const size_t full = 4 * (mask_slots(n, T) + POOL_SHARDS);          // every (tile, batch) a chunk + what the shards' rounding leaves unused
        const size_t head = (size_t)(pool0 - static_cast<char*>(base)) + ALIGN;      // bytes in front of the pool (incl. the caller's alignment slack)
        size_t cap = full;
        if (bytes) cap = bytes > head ? (bytes - head) / sub : 0;
        if (cap > full) cap = full;
        b.mp.cap = (uint32_t)cap;
        char* const pool_end = pool0 + cap * sub;
        if (pool_end > p) p = pool_end;
        if (out) *out = b;
        return (size_t)(p - static_cast<char*>(base)) + ALIGN;
    }
    carve(p, b.vals_alt, n);
    carve(p, b.tiles_alt, n);
    carve(p, b.sort_tmp, rs_tmp_words(n));
    if (mode == BIN_STATIC_MASKS) carve(p, b.cmask, cmask_words(n, T) * TILE_PIX);
    else { carve(p, b.pt_xy, n); carve(p, b.pt_depth, n); carve(p, b.pt_T, n); carve(p, b.pt_acc, n);
           carve(p, b.pt_order, (size_t)NXCD * tile_queue_stride((uint32_t)T)); carve(p, b.pt_queue, (size_t)TILE_QUEUE_WORDS); }
    if (out) *out = b;
    return (size_t)(p - static_cast<char*>(base)) + ALIGN;
}
// size of a BIN_MASK_POOL workspace whose pool holds `subchunks` sub-chunks (never less than the sort state needs)
static size_t bin_static_bytes(uint32_t R, int32_t W, int32_t H) { return bin_layout(R, W, H, nullptr, nullptr, BIN_STATIC_MASKS) + ALIGN; }
static size_t bin_pool_bytes(uint32_t R, int32_t W, int32_t H, size_t subchunks)
{
    BinWs b;
    const size_t full_bytes = bin_layout(R, W, H, nullptr, &b, BIN_MASK_POOL, 0);
    const size_t T = (size_t)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    const size_t full = 4 * (mask_slots((size_t)R, T) + POOL_SHARDS);
    if (subchunks >= full) return full_bytes;
    const size_t head = reinterpret_cast<size_t>(b.mp.pool) + ALIGN;                                   // (base is null: the pointer IS the offset)
    const size_t alt_end = reinterpret_cast<size_t>(b.tiles_alt) + align_up((size_t)R * sizeof(uint32_t)) + ALIGN;
    const size_t want = head + subchunks * (size_t)MASK_SUBCHUNK_WORDS * sizeof(uint32_t);
    return want > alt_end ? want : alt_end;
}
size_t point_layout(int32_t PN, void* base, PointWs* out)
{
    PointWs w;
    char* p = static_cast<char*>(base);
    const size_t n = (size_t)PN;
    carve(p, w.pos, n);
    carve(p, w.tiles_touched, n);
    carve(p, w.point_offsets, n);
    carve(p, w.T_state, n);
    carve(p, w.scan_tmp, scan_tmp_words(n));
    if (out) *out = w;
    return (size_t)(p - static_cast<char*>(base)) + ALIGN;
}

static inline void* aligned_base(const void* ws) { return reinterpret_cast<void*>(align_up(reinterpret_cast<size_t>(ws))); }

static int validate(const GofRasterArgs* a)
{
    if (!a) { set_error("args is NULL"); return GOF_E_INVALID; }
    if (a->P < 0 || a->W <= 0 || a->H <= 0) { set_error("bad P/W/H (%d, %d, %d)", a->P, a->W, a->H); return GOF_E_INVALID; }
    if (a->P == 0) return GOF_OK;
    if (!a->means3D || !a->opacities || !a->background || !a->viewmatrix || !a->projmatrix || !a->campos) {
        set_error("a required pointer is NULL"); return GOF_E_INVALID; }
    if ((a->shs == nullptr) == (a->colors_precomp == nullptr)) { set_error("provide exactly one of shs / colors_precomp"); return GOF_E_INVALID; }
    if (a->shs && (a->M <= 0 || a->D < 0 || a->D > 3 || (a->D + 1) * (a->D + 1) > a->M)) { set_error("SH degree %d does not fit M=%d", a->D, a->M); return GOF_E_INVALID; }
    if (a->shs_rest && (!a->shs || a->M != 16)) { set_error("shs_rest (separate DC / higher-band SH tensors) needs shs and M == 16, got M=%d", a->M); return GOF_E_INVALID; }
    if (!a->cov3D_precomp && (!a->scales || !a->rotations)) { set_error("scales/rotations or cov3D_precomp required"); return GOF_E_INVALID; }
    if (!a->view2gaussian_precomp && (!a->scales || !a->rotations)) { set_error("scales/rotations required to compute view2gaussian"); return GOF_E_INVALID; }
    return GOF_OK;
}

struct Dims { uint32_t gx, gy, ntiles; float focal_x, focal_y; };
static inline Dims dims_of(const GofRasterArgs* a)
{
    Dims d;
    d.gx = (a->W + TILE_X - 1) / TILE_X;
    d.gy = (a->H + TILE_Y - 1) / TILE_Y;
    d.ntiles = d.gx * d.gy;
    d.focal_y = a->H / (2.0f * a->tan_fovy);    // rasterizer_impl.cu:274-275
    d.focal_x = a->W / (2.0f * a->tan_fovx);
    return d;
}

// Sort (tile, id) instances by tile; the instances were emitted into the buffer pair chosen so that the
// result of the final pass lands in (b.tiles, b.vals).
// grid of a tile_ranges launch over up to n items (grid-stride kernel): at most 64 workgroups per CU
static inline uint32_t tile_ranges_grid(uint32_t n) { const uint32_t b = (n + 255u) / 256u; return b < 16384u ? (b ? b : 1u) : 16384u; }
static int sort_by_tile(const BinWs& b, uint32_t n, uint32_t* tiles_in, uint32_t* vals_in, int tile_bits, hipStream_t stream,
                        const uint32_t* n_dev = nullptr, bool first_hist_done = false)
{
    uint32_t* tiles_other = (tiles_in == b.tiles) ? b.tiles_alt : b.tiles;
    uint32_t* vals_other = (vals_in == b.vals) ? b.vals_alt : b.vals;
    uint32_t *kr = nullptr, *vr = nullptr;
    GOF_HIP_CHECK(radix_sort_pairs_u32_z(tiles_in, vals_in, tiles_other, vals_other, n, tile_bits, b.sort_tmp, &kr, &vr, stream, n_dev, 0, first_hist_done));
    if (kr != b.tiles || vr != b.vals) { set_error("internal: sort result in the wrong buffer"); return GOF_E_DEVICE; }
    return GOF_OK;
}

// instance emission in depth order + tile sort + tile ranges, shared by forward and integrate
// n_dev (nullable): the instance count is only known on the device; R is then the CAPACITY of the binning workspace
static int bin_gaussians(const GofRasterArgs* a, const Dims& d, uint32_t R, const GeomWs& g, const BinWs& b, const ImageWs& im,
                         const int32_t* radii, hipStream_t stream, const uint32_t* n_dev = nullptr)
{
    const int dbg = a->debug;
    const int tile_bits = (int)higher_msb(d.ntiles);
    if (R > 0) {
        const bool odd = radix_passes(tile_bits) & 1;
        uint32_t* t_in = odd ? b.tiles_alt : b.tiles;
        uint32_t* v_in = odd ? b.vals_alt : b.vals;
        // Where the tile sort runs as histogram / scan / scatter launches, its blocks are the emission's workgroups: the emission leaves the
        // first pass's histogram (the tile id's low byte: one LDS add per written slot) and that pass starts at its scan.
        // (Measured and dropped, profiles/r05_ab_call6_binning.txt: counting the digits of ALL passes in the emission for a tile sort of
        // single-kernel passes -- emit_instances 0.040 -> 0.084 ms at S1M, and that sort itself lost beyond 2 M pairs: 0.38 vs 0.23 ms.)
        uint32_t* const hist0 = (rs_block_items() == emit_block_slots()) ? radix_classic_hist(b.sort_tmp, R, tile_bits) : nullptr;
        { GOF_PROFILE("emit_instances", stream);
        // one wave per EMIT_SLOTS output slots (R: the instance count, or the workspace's capacity when only the device knows the count)
        hipLaunchKernelGGL(emit_instances, dim3(emit_instances_grid(R, a->P)), dim3(256), 0, stream, a->P, g.dval_a,
                           g.order_off, g.dkey_b, g.dval_b, t_in, v_in, d.gx, R, g.inst_first, hist0, n_dev ? 0u : rs_units((size_t)R)); }
        GOF_LAUNCH_CHECK(stream, dbg);
        { GOF_PROFILE("sort_instances_by_tile", stream);
        int rc = sort_by_tile(b, R, t_in, v_in, tile_bits, stream, n_dev, hist0 != nullptr);
        if (rc) return rc; }
        GOF_LAUNCH_CHECK(stream, dbg);
    }
    // im.ranges: gof_forward_fused ran stage 1 (gather_scan_rects clears the ranges) in this very call.  gof_forward_render and
    // gof_integrate_view are handed an image workspace by the caller: nothing guarantees that stage 1 of THIS frame went through it, and
    // stale ranges would walk point_list out of bounds -- they clear it themselves (5 us on paths that read a count back anyway).
    if (!n_dev) GOF_HIP_CHECK(hipMemsetAsync(im.ranges, 0, (size_t)d.ntiles * sizeof(uint2), stream));
    if (R > 0) {
        GOF_PROFILE("tile_ranges", stream);
        hipLaunchKernelGGL(tile_ranges, dim3(tile_ranges_grid(R)), dim3(256), 0, stream, R, b.tiles, im.ranges, 0, n_dev,
                           radix_sort_error_flag(b.sort_tmp, (size_t)R, tile_bits), async_status_word());
        GOF_LAUNCH_CHECK(stream, dbg);
    }
    // dispatch order of the tile kernels: every XCD an equal share of every cost class, heaviest first (gof_common.h: pop_tile)
    { GOF_PROFILE("order_tiles", stream);
    hipLaunchKernelGGL(order_tiles, dim3(1), dim3(1024), 0, stream, d.ntiles, im.ranges, nullptr, im.tile_order, im.tile_queue, nullptr, im.mask_cursors, nullptr, nullptr, nullptr); }
    GOF_LAUNCH_CHECK(stream, dbg);
    return GOF_OK;
}

// The backward's dispatch order: what a tile costs there is known exactly after the forward blend (the deepest list position one of
// its pixels blended = the entries the backward stages and walks), so the forward call leaves the order in the image workspace.
// usage_host (nullable): device-visible address of the caller's pinned GOF_USAGE_WORDS words (gof_forward_fused)
static void order_tiles_for_backward(const Dims& d, const ImageWs& im, hipStream_t stream, uint32_t* usage_host = nullptr)
{
    GOF_PROFILE("order_tiles_bw", stream);
    // (queue lengths at tile_queue[40..47]; the backward pops from heads in its own scratch, cleared per call -- it may run more than once per forward)
    hipLaunchKernelGGL(order_tiles, dim3(1), dim3(1024), 0, stream, d.ntiles, im.ranges, im.tile_cost, im.tile_order_bw, im.tile_queue + TILE_QUEUE_WORDS / 2, nullptr, nullptr, im.mask_cursors + POOL_SHARDS + 1,
                       im.mask_cursors, usage_host);
}

// The forward blend.  Default: the reference's arithmetic without its two fp64 divisions per pair (blend_forward.hip: pair_nodiv_cc).
// Verification mode (gof_set_forward_exact(1), or GOF_FW_EXACT=1 in the environment when the library is loaded): every pair with the
// divisions -- every output bit the oracle's.
// tile rectangle of a Gaussian intersected with its footprint box (opt-in: gof_set_tight_tile_rects / GOF_TIGHT_RECTS=1; preprocess.hip)
static std::atomic<int> g_tight_rects{ [] { const char* e = getenv("GOF_TIGHT_RECTS"); return (e && e[0] == '1') ? 1 : 0; }() };

// gof_set_integrate_pixel_pass(1) / GOF_INT_PIXELS=1: the opacity-field query's pixel pass in its pixel-centric form (rounds 1-4) instead of
// the ray-centric one (round 5) -- same outputs bit for bit, kept for A/B timing and as the cap fallback (integrate.hip)
static std::atomic<int> g_integrate_pixel_pass{ [] { const char* e = getenv("GOF_INT_PIXELS"); return (e && e[0] == '1') ? 1 : 0; }() };
static std::atomic<int> g_forward_exact{ [] { const char* e = getenv("GOF_FW_EXACT"); return (e && e[0] == '1') ? 1 : 0; }() };
// a mode of THIS call (GofRasterArgs, ABI 12: > 0 on, < 0 off) or, at 0, the process-wide default the setters / the environment gave
static inline bool mode_on(int32_t per_call, const std::atomic<int>& process_default)
{
    return per_call > 0 || (per_call == 0 && process_default.load(std::memory_order_relaxed) != 0);
}
static void launch_blend_forward(const GofRasterArgs* a, const Dims& d, const GeomWs& g, const BinWs& b, const ImageWs& im, float* out_color, hipStream_t stream)
{
    GOF_PROFILE("blend_forward", stream);
    auto* kernel = mode_on(a->forward_exact, g_forward_exact) ? blend_forward_exact : blend_forward;
    hipLaunchKernelGGL(kernel, dim3(xcd_padded_tiles(d.ntiles)), dim3(TILE_PIX), 0, stream,
                       im.ranges, b.vals, g.rec, g.fconic, a->W, a->H, d.focal_x, d.focal_y, a->background,
                       im.final_T, im.n_contrib, out_color, b.mp, im.mask_cursors, d.gx, d.ntiles, im.tile_order, im.tile_queue, im.tile_cost);
}

} // namespace gof

using namespace gof;

extern "C" {

const char* gof_last_error(void) { return g_error.c_str(); }
int gof_set_forward_exact(int on) { return g_forward_exact.exchange(on ? 1 : 0); }
int gof_set_integrate_pixel_pass(int on) { return g_integrate_pixel_pass.exchange(on ? 1 : 0); }
int gof_set_tight_tile_rects(int on) { return g_tight_rects.exchange(on ? 1 : 0); }
int gof_abi_version(void) { return 12; }  // 12: per-call modes in GofRasterArgs (forward_exact, tight_tile_rects, integrate_pixel_pass; round 6); 8: gof_set_forward_exact / gof_set_tight_tile_rects; 9: record pool in the backward scratch, gof_backward_query; 10: gof_forward_fused(usage_pinned_host), backward scratch without its scan (round 4); 11: gof_set_integrate_pixel_pass (round 5)
                                          // 4: densification entry points (gof_train_hip.h), integrate_points bounded by n_contrib; 7: workspace layouts of round 3
                                          // (tile queues in the image / point-binning workspaces, queue heads in the backward scratch: sizes from the same queries)

size_t gof_geom_bytes(int32_t P) { return geom_layout(P < 0 ? 0 : P, nullptr, nullptr) + ALIGN; }
size_t gof_geom_bytes_forward(int32_t P) { size_t fb = 0; geom_layout_ex(P < 0 ? 0 : P, nullptr, nullptr, &fb); return fb + ALIGN; }
size_t gof_image_bytes(int32_t W, int32_t H) { return image_layout(W, H, nullptr, nullptr) + ALIGN; }
// one size for both users of a binning workspace: the opacity-field query (static masks) and the forward blend with a FULL mask pool
size_t gof_binning_bytes(uint32_t R, int32_t W, int32_t H)
{
    const size_t a = bin_layout(R, W, H, nullptr, nullptr, BIN_STATIC_MASKS), b = bin_pool_bytes(R, W, H, (size_t)-1);
    return (a > b ? a : b) + ALIGN;
}
size_t gof_binning_bytes_for(uint32_t R, int32_t W, int32_t H, uint32_t mask_subchunks) { return bin_pool_bytes(R, W, H, mask_subchunks) + ALIGN; }
size_t gof_point_binning_bytes(uint32_t NI, int32_t W, int32_t H) { return bin_layout(NI, W, H, nullptr, nullptr, BIN_POINTS) + ALIGN; }
size_t gof_point_bytes(int32_t PN) { return point_layout(PN < 0 ? 0 : PN, nullptr, nullptr) + ALIGN; }

// ---- a second stream per (thread, device) for the part of the per-Gaussian stage that binning does not wait for ----------------------
// preprocess_fwd is bound by its fp64 arithmetic at four waves per SIMD, and only the blend reads what most
// of it writes: the sync-free forward runs the culls + binning inputs first (stage 1), then the rest (stage 2) on this stream beside
// the depth sort, scan, emission, tile sort and ranges, and the caller's stream waits for it in front of the blend.  The fork and the
// join are events between the two streams: to the caller (and to torch's caching allocator) everything stays ordered on ITS stream.
// (Measured and dropped, profiles/r05_ab_call7_binning.txt: the binning chain on a stream of the device's highest priority with
// stage 2 on the caller's -- the priority changes nothing latency-bound launches feel.)
namespace {
// workgroups per CU of stage 2 of the per-Gaussian kernel beside the binning chain (preprocess.hip: the kernel strides; 0 = one workgroup
// per 256 Gaussians, i.e. every wave slot of the device)
// Measured, interleaved A/B (profiles/r06_ab_call2_stage2_grid.txt; ms per fwd+bwd step, unbounded / 1 / 2 workgroups per CU / no fork):
// S1M 2.404 / 2.388 / 2.403 / 2.419, S1M-clustered 3.255 / 3.241 / 3.261 / 3.293, 6M Gaussians 4.160 / 4.181 / 4.228 / 4.181.  What precedes
// the blend is work-conserving -- stage 2 alone + the chain alone = what the two take side by side, 0.44 ms at S1M, 1.42 ms at 6M -- so
// the bound only decides who waits: at 1 M Gaussians one workgroup per CU lets the depth sort's 245-workgroup passes find their slots
// (0.162 -> 0.145 ms) while stage 2 still ends long before the blend; at 6 M the bounded stage 2 (1.18 ms) becomes the critical path
// (and half / a quarter of a workgroup per CU at S1M: stage 2 0.28 / 0.52 ms, steps 2.42 / 2.56 ms -- r06_ab_call3_*.txt).
// Hence: bounded up to GOF_K1_HEAVY_BOUND_MAX_P Gaussians, unbounded beyond.
#ifndef GOF_K1_HEAVY_WGS_PER_CU
#define GOF_K1_HEAVY_WGS_PER_CU 1
#endif
#ifndef GOF_K1_HEAVY_BOUND_MAX_P
#define GOF_K1_HEAVY_BOUND_MAX_P (2 << 20)
#endif
// 0: stage 2 is not forked (one per-Gaussian kernel in front of the binning chain) -- developer A/B builds only
#ifndef GOF_K1_SPLIT
#define GOF_K1_SPLIT 1
#endif
struct AuxStream {
    hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr, count_ready = nullptr; bool failed = false; int cus = 0;
    ~AuxStream()
    {   // a caller thread that ends gives its stream and events back (thread_local storage: runs at thread exit; at process exit the
        // runtime may already be gone -- errors are swallowed)
        if (count_ready) (void)hipEventDestroy(count_ready);
        if (join) (void)hipEventDestroy(join);
        if (fork) (void)hipEventDestroy(fork);
        if (s) (void)hipStreamDestroy(s);
        (void)hipGetLastError();
    }
};
// the (thread, device)'s stream + events, created at first use; nullptr if the runtime refuses (the forward then runs on one stream)
AuxStream* aux_stream()
{
    static thread_local AuxStream per_device[MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) { (void)hipGetLastError(); return nullptr; }
    AuxStream& a = per_device[dev];
    if (a.failed) return nullptr;
    if (!a.fork) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; }
        if (hipStreamCreateWithPriority(&a.s, hipStreamNonBlocking, least) != hipSuccess || hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&a.join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&a.count_ready, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError(); a.failed = true; return nullptr; }
        if (hipDeviceGetAttribute(&a.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || a.cus <= 0) { (void)hipGetLastError(); a.cus = 256; }
    }
    return &a;
}
// makes `stream` (the caller's) wait for the library stream's work of this call: explicitly in front of the first consumer, and on every
// other way out
struct AuxJoin {
    hipStream_t stream = nullptr;
    hipEvent_t pending = nullptr;
    void now()
    {
        if (!pending) return;
        (void)hipStreamWaitEvent(stream, pending, 0);
        pending = nullptr;
    }
    ~AuxJoin() { now(); }
};
}

// preprocess + depth sort + scan, all asynchronous; *total_dev_out = device address of the instance count
// join (nullable): the caller can take stage 2 of the per-Gaussian kernel on the second stream; it must call join->now() in front of
// the first launch that reads the records / conics / footprints / depths / clamp flags
// full_footprint: the per-Gaussian stage also leaves the footprint's pixel box, q and zfront (preprocess.hip: footprint_bbox<true>) -- what
// the opacity-field query reads; a forward that is followed by the blend needs the conic alone (tight tile rectangles: the box as well)
// geom_bytes: the caller's geometry workspace -- the footprints' pixel boxes are stored only into one of gof_geom_bytes(P)
static int forward_stage1(const GofRasterArgs* a, const GeomWs& g, size_t geom_bytes, const ImageWs& im, int32_t* radii, const uint32_t** total_dev_out, hipStream_t stream,
                          bool full_footprint, uint32_t* total_host_mapped = nullptr, AuxJoin* join = nullptr)
{
    float4* const bbox_out = geom_bytes >= gof_geom_bytes(a->P) ? g.bbox : nullptr;
    const Dims d = dims_of(a);
    const Cam cam = { a->viewmatrix, a->projmatrix, a->campos };
    if (a->prefiltered) GOF_HIP_CHECK(hipMemsetAsync(g.flags, 0, 4 * sizeof(uint32_t), stream));   // only then written and read
    // (SH rows through the LDS as in preprocess_bwd: built in round 5 and measured SLOWER than the per-thread reads -- S1M 0.122 vs
    // 0.110 ms, 6M Gaussians 0.648 vs 0.588 ms: 50 KB of LDS leave three workgroups per CU and two more barriers;
    // profiles/r05_ab_call4_preprocess_fwd.txt -- and removed in round 6)
    const int k1_bits = (a->prefiltered ? 1 : 0) | (mode_on(a->tight_tile_rects, g_tight_rects) ? 2 : 0);
    const bool foot = full_footprint || (k1_bits & 2);
#define GOF_K1_LAUNCH(STAGE, FOOT, STREAM) hipLaunchKernelGGL((preprocess_fwd<STAGE, FOOT>), dim3(k1_grid), dim3(256), 0, STREAM,                 \
                       a->P, a->D, a->M, a->means3D, a->scales, a->scale_modifier, a->rotations, a->opacities, a->shs, a->shs_rest,                          \
                       a->cov3D_precomp, a->colors_precomp, a->view2gaussian_precomp, cam, a->W, a->H, a->tan_fovx, a->tan_fovy,                             \
                       d.focal_x, d.focal_y, a->kernel_size, d.gx, d.gy, k1_bits,                                                                             \
                       radii, g.depths, g.rec, g.conic, bbox_out, g.fconic, g.tiles_touched, g.rect, g.clamped, g.dkey_a, g.dval_a, g.flags, k1_zero_ptr, k1_zero_n)
    // what the depth sort and the fused gather + scan behind it need cleared: cleared by the per-Gaussian kernel, the first launch of the
    // frame, instead of by a memset launch in front of the sort (depth sort 0.147 -> 0.140 ms at S1M, profiles/r05_ab_call8_binning.txt)
    const size_t scan_words = gather_scan_state_words((size_t)a->P);
    const size_t sort_zero = radix_zero_words((size_t)a->P, 32);                 // words at the start of the sort's scratch its single-kernel passes want zero (0: none)
    const bool sort_zero_to_end = sort_zero == rs_tmp_words((size_t)a->P);       // ... reaching the scan's state: one range
    uint32_t* k1_zero_ptr = nullptr;
    uint32_t k1_zero_n = 0;
    if (sort_zero == 0 || sort_zero_to_end) {
        k1_zero_ptr = sort_zero ? g.sort_tmp : g.sort_tmp + rs_tmp_words((size_t)a->P);
        k1_zero_n = (uint32_t)(sort_zero + scan_words);
    }
    // (tight tile rectangles take their tiles_touched from stage 2's footprint box: one kernel then)
    AuxStream* const aux = (GOF_K1_SPLIT && join && !(k1_bits & 2) && !a->debug) ? aux_stream() : nullptr;
    uint32_t k1_grid = (uint32_t)((a->P + 255) / 256);
    { GOF_PROFILE("preprocess_fwd", stream);
    if (aux) GOF_K1_LAUNCH(1, 0, stream);
    else if (foot) GOF_K1_LAUNCH(0, 1, stream);
    else GOF_K1_LAUNCH(0, 0, stream); }
    // (measured, profiles/r05_ab_call5_split_preprocess.txt: S1M 2.459 -> 2.434 ms per step, 6M Gaussians 4.18 -> 4.13, means over three
    // boxes each -- the two streams share the CUs, the depth sort runs 0.108 -> 0.166 ms beside stage 2; forking BEHIND the depth sort
    // instead loses: 2.468 / 4.18.  The device offers no priority below the caller's stream's to put stage 2 on.)
    if (aux) {
        GOF_HIP_CHECK(hipGetLastError());
        GOF_HIP_CHECK(hipEventRecord(aux->fork, stream));
        GOF_HIP_CHECK(hipStreamWaitEvent(aux->s, aux->fork, 0));
        if (GOF_K1_HEAVY_WGS_PER_CU > 0 && a->P <= GOF_K1_HEAVY_BOUND_MAX_P)
            k1_grid = std::min(k1_grid, (uint32_t)(GOF_K1_HEAVY_WGS_PER_CU * aux->cus));      // bounded: the chain's workgroups find free slots
        { GOF_PROFILE("preprocess_fwd_heavy", aux->s);
          if (foot) GOF_K1_LAUNCH(2, 1, aux->s);
          else GOF_K1_LAUNCH(2, 0, aux->s); }
        GOF_HIP_CHECK(hipGetLastError());
        join->stream = stream;
        join->pending = aux->join;
        GOF_HIP_CHECK(hipEventRecord(aux->join, aux->s));
    }
#undef GOF_K1_LAUNCH
    GOF_LAUNCH_CHECK(stream, a->debug);
    // depth order of the Gaussians (4 passes over P; an even number of passes returns to the *_a buffers)
    // (the state words of the fused gather + scan behind it lie right behind the sort's scratch: the sort's own memset clears them too)
    // (Measured and removed, round 6 -- profiles/r06_ab_call3_three_pass_depth_sort.txt: THREE passes of 9 bits under a caller's promise
    // of depth keys below 2^27, detected and redone when broken.  Beside stage 2 the sort's time is set by the interference, not by its
    // passes: 0.139 vs 0.137 ms at S1M, 0.620 vs 0.615 at 6 M Gaussians, step times equal to 0.1 %.)
    uint32_t* const scan_state = g.sort_tmp + rs_tmp_words((size_t)a->P);
    uint32_t *kr = nullptr, *vr = nullptr;          // where the sort leaves its keys / values (kr: read by gather_scan_rects)
    { GOF_PROFILE("sort_gaussians_by_depth", stream);
    GOF_HIP_CHECK(radix_sort_pairs_u32_z(g.dkey_a, g.dval_a, g.dkey_b, g.dval_b, (size_t)a->P, 32, g.sort_tmp, &kr, &vr, stream, nullptr,
                                         k1_zero_ptr ? 0 : scan_words, false, k1_zero_ptr != nullptr));
    if (vr != g.dval_a) { set_error("internal: depth sort result in the wrong buffer"); return GOF_E_DEVICE; } }
    GOF_LAUNCH_CHECK(stream, a->debug);
    // first instance of every depth-sorted Gaussian + the instance count (replaces rasterizer_impl.cu:332)
    { GOF_PROFILE("scan_tiles", stream);
    // dkey_b / dval_b are free after the (even number of) sort passes: they take the depth-ordered rectangles; the counts are
    // scanned in place
    hipLaunchKernelGGL(gather_scan_rects, dim3(gather_scan_tiles((size_t)a->P)), dim3(gather_scan_threads()), 0, stream, (uint32_t)a->P, g.rect, g.dval_a, kr, g.dkey_b, g.dval_b,
                       g.order_off, radix_sort_error_flag(g.sort_tmp, (size_t)a->P, 32), im.ranges, d.ntiles, scan_state, total_host_mapped);
    GOF_HIP_CHECK(hipGetLastError());
    *total_dev_out = scan_state + 1; }
    GOF_LAUNCH_CHECK(stream, a->debug);
    return GOF_OK;
}

static int prepare_impl(const GofRasterArgs* a, void* geom_ws, size_t geom_bytes, void* image_ws, size_t image_bytes,
                        int32_t* radii, uint32_t* num_rendered_host, void* stream_, bool full_footprint)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = validate(a);
    if (!rc) rc = take_async_status();
    if (rc) return rc;
    if (!num_rendered_host) { set_error("num_rendered_host is NULL"); return GOF_E_INVALID; }
    *num_rendered_host = 0;
    if (a->P == 0) return GOF_OK;
    if (!radii || !geom_ws || !image_ws) { set_error("radii / workspace is NULL"); return GOF_E_INVALID; }
    { const size_t need = full_footprint ? gof_geom_bytes(a->P) : gof_geom_bytes_forward(a->P);
      if (geom_bytes < need) { set_error("geometry workspace too small: %zu < %zu", geom_bytes, need); return GOF_E_WORKSPACE; } }
    if (image_bytes < gof_image_bytes(a->W, a->H)) { set_error("image workspace too small"); return GOF_E_WORKSPACE; }
    GeomWs g; ImageWs im;
    geom_layout(a->P, aligned_base(geom_ws), &g);
    image_layout(a->W, a->H, aligned_base(image_ws), &im);
    const uint32_t* total_dev = nullptr;
    rc = forward_stage1(a, g, geom_bytes, im, radii, &total_dev, stream, full_footprint);
    if (rc) return rc;
    // one blocking 4-byte read-back, as the reference (rasterizer_impl.cu:336)
    uint32_t host_words[2] = { 0, 0 };
    GOF_HIP_CHECK(hipMemcpyAsync(&host_words[0], total_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (a->prefiltered)
        GOF_HIP_CHECK(hipMemcpyAsync(&host_words[1], g.flags, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    *num_rendered_host = host_words[0];
    if (host_words[0] >= GOF_SORT_FAILED_COUNT) {
        *num_rendered_host = 0;
        set_error("depth sort: a single-kernel radix pass timed out waiting for a predecessor block (GPU heavily oversubscribed?)");
        return GOF_E_DEVICE;
    }
    if (a->prefiltered && host_words[1]) {
        set_error("Point is filtered although prefiltered is set. This shouldn't happen!");   // auxiliary.h:193-197
        return GOF_E_PREFILTER;
    }
    return GOF_OK;
}

int gof_forward_prepare(const GofRasterArgs* a, void* geom_ws, size_t geom_bytes, void* image_ws, size_t image_bytes,
                        int32_t* radii, uint32_t* num_rendered_host, void* stream_)
{
    return prepare_impl(a, geom_ws, geom_bytes, image_ws, image_bytes, radii, num_rendered_host, stream_, false);
}
// the same stage in front of gof_integrate_view / gof_integrate_run: the footprints complete (pixel box, q, zfront)
int gof_integrate_prepare(const GofRasterArgs* a, void* geom_ws, size_t geom_bytes, void* image_ws, size_t image_bytes,
                          int32_t* radii, uint32_t* num_rendered_host, void* stream_)
{
    return prepare_impl(a, geom_ws, geom_bytes, image_ws, image_bytes, radii, num_rendered_host, stream_, true);
}

// The whole forward without a pipeline bubble: the binning workspace is sized for `capacity` instances chosen by the caller (e.g.
// 1.25 x the count of the previous frame); every launch after the scan is sized for the capacity and reads the actual count on the
// device; the count is copied to PINNED host memory right after the scan and the host waits for THAT event only -- ~0.3 ms into
// the call, with emission, tile sort and the blend already queued behind it.  If the count exceeds the capacity the call returns
// GOF_E_CAPACITY (nothing was written out of bounds) and the caller redoes the frame through gof_forward_prepare/render.
int gof_forward_fused(const GofRasterArgs* a, uint32_t capacity, void* geom_ws, size_t geom_bytes, void* binning_ws, size_t binning_bytes,
                      void* image_ws, size_t image_bytes, int32_t* radii, float* out_color, uint32_t* num_rendered_pinned_host,
                      uint32_t* usage_pinned_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = validate(a);
    if (!rc) rc = take_async_status();
    if (rc) return rc;
    if (!num_rendered_pinned_host || !out_color) { set_error("num_rendered_pinned_host / out_color is NULL"); return GOF_E_INVALID; }
    if (a->P == 0 || a->prefiltered || a->debug) { set_error("gof_forward_fused: empty / prefiltered / debug calls use gof_forward_prepare + gof_forward_render"); return GOF_E_INVALID; }
    if (!radii || !geom_ws || !binning_ws || !image_ws) { set_error("radii / workspace is NULL"); return GOF_E_INVALID; }
    if (geom_bytes < gof_geom_bytes_forward(a->P) || image_bytes < gof_image_bytes(a->W, a->H) || binning_bytes < gof_binning_bytes_for(capacity, a->W, a->H, 0)) {
        set_error("workspace too small (geom %zu, image %zu, binning %zu)", geom_bytes, image_bytes, binning_bytes); return GOF_E_WORKSPACE; }
    GeomWs g; ImageWs im; BinWs b;
    geom_layout(a->P, aligned_base(geom_ws), &g);
    image_layout(a->W, a->H, aligned_base(image_ws), &im);
    bin_layout(capacity, a->W, a->H, aligned_base(binning_ws), &b, BIN_MASK_POOL, binning_bytes);
    const Dims d = dims_of(a);
    const uint32_t* total_dev = nullptr;
    // the two host-bound results -- the instance count (needed ~0.3 ms into the call) and the frame's pool counters (at its end) -- are
    // stored by the kernels that produce them into the caller's pinned memory, if that is device-mapped: no copy launches in the stream
    uint32_t* const count_mapped = device_view_of_pinned(num_rendered_pinned_host);
    uint32_t* const usage_mapped = device_view_of_pinned(usage_pinned_host);
    *num_rendered_pinned_host = 0xFFFFFFFFu;
    AuxJoin heavy;                       // (its destructor joins on every way out of this call)
    rc = forward_stage1(a, g, geom_bytes, im, radii, &total_dev, stream, false, count_mapped, &heavy);
    if (rc) return rc;
    // the event the host waits for: one per (thread, device) -- an event belongs to the device it was created on
    AuxStream* const aux = aux_stream();
    hipEvent_t ev = aux ? aux->count_ready : nullptr;
    bool ev_local = false;
    if (!ev) { GOF_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); ev_local = true; }
    struct EvGuard { hipEvent_t e; bool own; ~EvGuard() { if (own) (void)hipEventDestroy(e); } } ev_guard{ ev, ev_local };
    if (!count_mapped) GOF_HIP_CHECK(hipMemcpyAsync(num_rendered_pinned_host, total_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipEventRecord(ev, stream));
    if (capacity > 0) {
        rc = bin_gaussians(a, d, capacity, g, b, im, radii, stream, total_dev);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(order_tiles, dim3(1), dim3(1024), 0, stream, d.ntiles, im.ranges, nullptr, im.tile_order, im.tile_queue, nullptr, im.mask_cursors, nullptr, nullptr, nullptr);
    }
    heavy.now();                         // the two streams meet: records, conics and footprints are stage 2's, lists and ranges the chain's -- the blend is their first reader
    launch_blend_forward(a, d, g, b, im, out_color, stream);
    GOF_LAUNCH_CHECK(stream, 0);
    order_tiles_for_backward(d, im, stream, usage_mapped);
    if (usage_pinned_host && !usage_mapped)       // (not device-mapped: the copy form, as gof_forward_usage_async)
        GOF_HIP_CHECK(hipMemcpyAsync(usage_pinned_host, im.mask_cursors, (POOL_SHARDS + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipEventSynchronize(ev));
    if (*num_rendered_pinned_host >= GOF_SORT_FAILED_COUNT) {
        set_error("depth sort: a single-kernel radix pass timed out waiting for a predecessor block (GPU heavily oversubscribed?)");
        return GOF_E_DEVICE;
    }
    if (*num_rendered_pinned_host > capacity) {
        set_error("instance count %u exceeds the capacity %u of the binning workspace", *num_rendered_pinned_host, capacity);
        return GOF_E_CAPACITY;
    }
    return GOF_OK;
}

int gof_forward_render(const GofRasterArgs* a, uint32_t R, const int32_t* radii, void* geom_ws, size_t geom_bytes,
                       void* binning_ws, size_t binning_bytes, void* image_ws, size_t image_bytes, float* out_color, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = validate(a);
    if (rc) return rc;
    if (!out_color) { set_error("out_color is NULL"); return GOF_E_INVALID; }
    const size_t HW = (size_t)a->W * a->H;
    if (a->P == 0) {   // rasterize_points.cu:68, 85: zero image, nothing rendered
        GOF_HIP_CHECK(hipMemsetAsync(out_color, 0, GOF_OUTPUT_CHANNELS * HW * sizeof(float), stream));
        return GOF_OK;
    }
    if (!radii || !geom_ws || !binning_ws || !image_ws) { set_error("radii / workspace is NULL"); return GOF_E_INVALID; }
    if (geom_bytes < gof_geom_bytes_forward(a->P) || image_bytes < gof_image_bytes(a->W, a->H) || binning_bytes < gof_binning_bytes_for(R, a->W, a->H, 0)) {
        set_error("workspace too small (geom %zu, image %zu, binning %zu)", geom_bytes, image_bytes, binning_bytes); return GOF_E_WORKSPACE; }
    GeomWs g; ImageWs im; BinWs b;
    geom_layout(a->P, aligned_base(geom_ws), &g);
    image_layout(a->W, a->H, aligned_base(image_ws), &im);
    bin_layout(R, a->W, a->H, aligned_base(binning_ws), &b, BIN_MASK_POOL, binning_bytes);
    const Dims d = dims_of(a);
    rc = bin_gaussians(a, d, R, g, b, im, radii, stream);
    if (rc) return rc;
    launch_blend_forward(a, d, g, b, im, out_color, stream);
    GOF_LAUNCH_CHECK(stream, a->debug);
    order_tiles_for_backward(d, im, stream);
    GOF_LAUNCH_CHECK(stream, a->debug);
    return GOF_OK;
}

// backward scratch (the number of a Gaussian's first instance comes from the forward: GeomWs::inst_first, written by emit_instances --
// until round 4 the backward scanned tiles_touched itself, two launches and 14 us per step):
// the backward's queue heads + the record pool's cursor, then per tile instance (R of them) a slot word (slot + 1 of the instance's
// partial gradient record, 0 = none), and the record POOL: per record one 64-byte line of 16 partial gradients (the 17th follows from them per Gaussian: gather_tile_partials)
// 16.  The pool holds `records` records: R for the worst case (every instance staged), or the number the forward actually staged
// (gof_backward_query: ~30 % of R at S1M) -- 4 + 64 x 0.3 B per instance instead of 68.
struct BwdScratch { uint32_t* queue; uint32_t* slot_of; float4* part16; };
constexpr size_t BWD_QUEUE_BYTES = 256;      // the backward's tile-queue heads ([0..7]) and the record pool's cursor ([BWD_REC_CURSOR]) sit right in front of the slot words: one memset clears both
constexpr int BWD_REC_CURSOR = 16;
constexpr uint32_t BWD_REC_SLACK = 0;        // (a tile takes exactly its staged entries: the sum is what gof_backward_query reports)
static size_t bwd_scratch_layout(int32_t P, uint32_t R, uint32_t records, void* base, BwdScratch* o)
{
    char* p = static_cast<char*>(base);
    const size_t p0 = reinterpret_cast<size_t>(p);
    BwdScratch t;
    (void)P;
    carve(p, t.queue, BWD_QUEUE_BYTES / 4 + (size_t)R + 1);
    t.slot_of = t.queue + BWD_QUEUE_BYTES / 4;
    carve(p, t.part16, 4 * ((size_t)records + 1));
    if (o) *o = t;
    return reinterpret_cast<size_t>(p) - p0;
}
size_t gof_backward_scratch_bytes(int32_t P, uint32_t num_rendered) { return bwd_scratch_layout(P, num_rendered, num_rendered + BWD_REC_SLACK, nullptr, nullptr) + ALIGN; }
size_t gof_backward_scratch_bytes_for(int32_t P, uint32_t num_rendered, uint32_t staged_entries)
{
    return bwd_scratch_layout(P, num_rendered, (staged_entries < num_rendered ? staged_entries : num_rendered) + BWD_REC_SLACK, nullptr, nullptr) + ALIGN;
}
// records a scratch of `bytes` bytes can hold (the inverse of gof_backward_scratch_bytes_for)
static uint32_t bwd_scratch_records(int32_t P, uint32_t R, size_t bytes)
{
    const size_t fixed = bwd_scratch_layout(P, R, 0, nullptr, nullptr) + ALIGN;
    if (bytes < fixed) return 0;
    size_t n = (bytes - fixed) / 64 + 16;                   // (the pool arrays of the zero-record layout already hold one record and their alignment padding)
    if (n > (size_t)R + BWD_REC_SLACK) n = (size_t)R + BWD_REC_SLACK;
    while (n > 0 && bwd_scratch_layout(P, R, (uint32_t)n, nullptr, nullptr) + ALIGN > bytes) n--;      // (the two pool arrays are 256-byte aligned each)
    return (uint32_t)n;
}

// stages: 1 = zero-fill + blend_backward (K8), 2 = preprocess_bwd (K9), 3 = both
static int backward_impl(int stages, const GofRasterArgs* a, uint32_t R, const int32_t* radii, const void* geom_ws, size_t geom_bytes,
                 const void* binning_ws, size_t binning_bytes, const void* image_ws, size_t image_bytes, const float* dL_dout,
                 float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dsh_rest, float* dL_dscales, float* dL_drotations, float* dL_dview2gaussian, void* scratch, size_t scratch_bytes,
                 void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = validate(a);
    if (!rc && (stages & 1)) rc = take_async_status();
    if (rc) return rc;
    if (a->P == 0) return GOF_OK;
    if (!scratch || scratch_bytes < bwd_scratch_layout(a->P, R, 0, nullptr, nullptr) + ALIGN) { set_error("backward scratch missing or too small (gof_backward_scratch_bytes)"); return GOF_E_WORKSPACE; }
    // the record pool is as large as the caller's buffer allows: R records (gof_backward_scratch_bytes: always enough), the forward's
    // staged count (gof_backward_query + gof_backward_scratch_bytes_for), or a guess from earlier frames that the caller verifies
    // afterwards (gof_forward_usage_async): a pool that turns out too small drops the records that do not fit.
    const uint32_t rec_cap = bwd_scratch_records(a->P, R, scratch_bytes);
    BwdScratch ws;
    bwd_scratch_layout(a->P, R, rec_cap, aligned_base(scratch), &ws);
    if (!dL_dout || !dL_dmeans2D || !dL_dcolors || !dL_dopacity || !dL_dmeans3D || !dL_dscales || !dL_drotations || !dL_dview2gaussian ||
        (a->M > 0 && a->shs && !dL_dsh) || !radii) { set_error("a gradient / radii pointer is NULL"); return GOF_E_INVALID; }
    const bool split_sh = a->shs_rest != nullptr;
    if (split_sh != (dL_dsh_rest != nullptr)) { set_error("dL_dsh_rest must be given exactly when args->shs_rest is"); return GOF_E_INVALID; }
    if (!a->scales || !a->rotations) { set_error("backward needs scales and rotations (backward.cu:621)"); return GOF_E_INVALID; }
    if (geom_bytes < gof_geom_bytes_forward(a->P) || image_bytes < gof_image_bytes(a->W, a->H) || binning_bytes < gof_binning_bytes_for(R, a->W, a->H, 0)) {
        set_error("workspace too small"); return GOF_E_WORKSPACE; }
    GeomWs g; ImageWs im; BinWs b;
    geom_layout(a->P, aligned_base(geom_ws), &g);
    image_layout(a->W, a->H, aligned_base(image_ws), &im);
    bin_layout(R, a->W, a->H, aligned_base(binning_ws), &b, BIN_MASK_POOL, binning_bytes);
    const Dims d = dims_of(a);
    const size_t P = (size_t)a->P;
    // torch::zeros of the binding (rasterize_points.cu:161-170): needed for what blend_backward ACCUMULATES into and for the dead
    // dL_dcov3D; preprocess_bwd writes every element of dL_dmeans3D / dL_dscales / dL_drotations (zeros for culled Gaussians)
    // preprocess_bwd<true> (SH rows tiled through LDS) writes every element of dL_dsh itself: no memset for it
    const bool k9_tiled = split_sh || (a->shs && dL_dsh && a->M == 16 && ((reinterpret_cast<uintptr_t>(a->shs) | reinterpret_cast<uintptr_t>(dL_dsh)) & 15) == 0);
    if (stages & 1) {
    { GOF_PROFILE("backward_memsets", stream);
      // exactly adjacent buffers (the Python binding carves dL_dview2gaussian | dL_dcov3D | dL_dmeans2D | dL_dcolors from one
      // allocation) are cleared by ONE memset: each launch costs ~5 us of queue time, the bytes themselves 0.01 ms
      struct Range { char* p; size_t n; } r[6];
      int nr = 0;
      auto add = [&](void* ptr, size_t floats) { if (ptr && floats) { r[nr].p = static_cast<char*>(ptr); r[nr].n = floats * sizeof(float); nr++; } };
      add(dL_dcov3D, 6 * P);           // dL_dmeans2D / dL_dcolors / dL_dopacity / dL_dview2gaussian are written completely by gather_tile_partials
      if (dL_dsh && a->M > 0 && !k9_tiled) add(dL_dsh, 3 * P * (size_t)a->M);
      std::sort(r, r + nr, [](const Range& x, const Range& y) { return x.p < y.p; });
      for (int i = 0; i < nr; ) {
          char* p0 = r[i].p; size_t n = r[i].n; int j = i + 1;
          while (j < nr && r[j].p == p0 + n) { n += r[j].n; j++; }
          GOF_HIP_CHECK(hipMemsetAsync(p0, 0, n, stream));
          i = j;
      }
      if (R > 0) GOF_HIP_CHECK(hipMemsetAsync(ws.queue, 0, BWD_QUEUE_BYTES + 4 * (size_t)R, stream)); }

    if (R > 0) {
        GOF_PROFILE("blend_backward", stream);
        hipLaunchKernelGGL(blend_backward, dim3(xcd_padded_tiles(d.ntiles)), dim3(TILE_PIX), 0, stream,
                           im.ranges, b.vals, g.rec, g.conic, b.mp, a->W, a->H, d.focal_x, d.focal_y, a->background, im.final_T,
                           im.n_contrib, dL_dout, g.rect, g.inst_first, ws.part16, ws.slot_of, ws.queue + BWD_REC_CURSOR, rec_cap, d.gx, d.ntiles,
                           im.tile_order_bw, ws.queue, im.tile_queue + TILE_QUEUE_WORDS / 2 + NXCD);
        GOF_LAUNCH_CHECK(stream, a->debug);
    }
    { GOF_PROFILE("gather_tile_partials", stream);
      // R == 0: tiles_touched is 0 everywhere, the kernel writes zeros
      hipLaunchKernelGGL(gather_tile_partials, dim3((unsigned)(((size_t)a->P * 4 + 255) / 256)), dim3(256), 0, stream, a->P, g.inst_first, g.tiles_touched, ws.part16, g.conic,
                         ws.slot_of, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dview2gaussian);
      GOF_LAUNCH_CHECK(stream, a->debug); }
    }
    if (!(stages & 2)) return GOF_OK;
    const Cam cam = { a->viewmatrix, a->projmatrix, a->campos };
    GOF_PROFILE("preprocess_bwd", stream);
#define GOF_K9_LAUNCH(MODE) hipLaunchKernelGGL(preprocess_bwd<MODE>, dim3((a->P + 255) / 256), dim3(256), 0, stream, a->P, a->D, a->M, a->means3D, \
        radii, a->shs, a->shs_rest, g.clamped, a->scales, a->rotations, cam, dL_dview2gaussian, dL_dcolors, dL_dmeans3D, dL_dsh, dL_dsh_rest,     \
        dL_dscales, dL_drotations)
    if (split_sh) GOF_K9_LAUNCH(2);
    else if (k9_tiled) GOF_K9_LAUNCH(1);
    else GOF_K9_LAUNCH(0);
#undef GOF_K9_LAUNCH
    GOF_LAUNCH_CHECK(stream, a->debug);
    return GOF_OK;
}

#define GOF_BACKWARD_ENTRY(NAME, STAGES)                                                                                                          \
int NAME(const GofRasterArgs* a, uint32_t R, const int32_t* radii, const void* geom_ws, size_t geom_bytes, const void* binning_ws,                \
         size_t binning_bytes, const void* image_ws, size_t image_bytes, const float* dL_dout, float* dL_dmeans2D, float* dL_dcolors,             \
         float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest, float* dL_dscales, float* dL_drotations,    \
         float* dL_dview2gaussian, void* scratch, size_t scratch_bytes, void* stream)                                                             \
{                                                                                                                                                 \
    return backward_impl(STAGES, a, R, radii, geom_ws, geom_bytes, binning_ws, binning_bytes, image_ws, image_bytes, dL_dout, dL_dmeans2D,        \
                         dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dsh_rest, dL_dscales, dL_drotations, dL_dview2gaussian,      \
                         scratch, scratch_bytes, stream);                                                                                         \
}
GOF_BACKWARD_ENTRY(gof_backward, 3)
GOF_BACKWARD_ENTRY(gof_backward_blend, 1)
GOF_BACKWARD_ENTRY(gof_backward_preprocess, 2)
#undef GOF_BACKWARD_ENTRY

// What the backward of the frame in `image_ws` will need, and whether the forward's mask pool was large enough:
//   out3[0] = tile-list entries the backward stages = records its scratch must hold (sum over the tiles of the deepest blended list
//             position, left by the forward's order_tiles_for_backward),
//   out3[1] = sub-chunks of contributor masks the forward asked for, out3[2] = sub-chunks a binning workspace of `binning_bytes`
//             holds: [1] > [2] means masks are missing -- repeat the frame's forward with a binning workspace of
//             gof_binning_bytes_for(R, W, H, out3[1]) or more, then its backward.
// Two forms.  gof_backward_query SYNCHRONISES `stream` (268 bytes read back) and is for a caller that sizes its workspaces BEFORE the
// backward.  gof_forward_usage_async + gof_usage_decode are for a caller that launches the backward optimistically (pools sized from
// earlier frames; a backward whose pools were too small drops what does not fit and is simply repeated): the first enqueues the copy
// of the raw counters (GOF_USAGE_WORDS words) into PINNED host memory behind the frame's forward and returns at once, the second
// -- host arithmetic only -- turns the words into the three numbers once the caller has waited for its own event behind the copy.
static void decode_usage(const uint32_t* words, uint32_t R, int32_t W, int32_t H, size_t binning_bytes, uint32_t* out3)
{
    BinWs b;
    bin_layout(R, W, H, nullptr, &b, BIN_MASK_POOL, binning_bytes);
    out3[2] = b.mp.cap;
    const uint32_t chunks = b.mp.cap / 4u, shards = pool_shards(chunks), per = chunks / shards;      // chunks (of 4 sub-chunks) taken from the shards + chunks no shard had room for
    uint64_t asked = words[POOL_SHARDS];
    for (uint32_t k = 0; k < shards; k++) asked += words[k] < per ? words[k] : per;
    out3[1] = (uint32_t)(4u * asked > 0xFFFFFFFFull ? 0xFFFFFFFFull : 4u * asked);
    if (words[POOL_SHARDS]) out3[1] = out3[1] > b.mp.cap ? out3[1] : b.mp.cap + 4u;                   // (unserved requests: certainly more than held)
    const uint32_t staged = words[POOL_SHARDS + 1];
    out3[0] = staged > R ? R : staged;
}
int gof_usage_decode(const uint32_t* words_host, uint32_t R, int32_t W, int32_t H, size_t binning_bytes, uint32_t* out3_host)
{
    if (!words_host || !out3_host || W <= 0 || H <= 0) { set_error("NULL argument"); return GOF_E_INVALID; }
    decode_usage(words_host, R, W, H, binning_bytes, out3_host);
    return GOF_OK;
}
int gof_forward_usage_async(const GofRasterArgs* a, const void* image_ws, size_t image_bytes, uint32_t* words_pinned_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = validate(a);
    if (rc) return rc;
    if (!words_pinned_host) { set_error("words_pinned_host is NULL"); return GOF_E_INVALID; }
    if (!image_ws || image_bytes < gof_image_bytes(a->W, a->H)) { set_error("image workspace missing or too small"); return GOF_E_WORKSPACE; }
    ImageWs im;
    image_layout(a->W, a->H, aligned_base(const_cast<void*>(image_ws)), &im);
    GOF_HIP_CHECK(hipMemcpyAsync(words_pinned_host, im.mask_cursors, (POOL_SHARDS + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    return GOF_OK;
}
int gof_backward_query(const GofRasterArgs* a, uint32_t R, size_t binning_bytes, const void* image_ws, size_t image_bytes, uint32_t* out3_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = validate(a);
    if (rc) return rc;
    if (!out3_host) { set_error("out3_host is NULL"); return GOF_E_INVALID; }
    out3_host[0] = R; out3_host[1] = 0; out3_host[2] = 0;
    if (a->P == 0 || R == 0) { out3_host[0] = 0; return GOF_OK; }
    uint32_t words[POOL_SHARDS + 2];
    rc = gof_forward_usage_async(a, image_ws, image_bytes, words, stream_);
    if (rc) return rc;
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    decode_usage(words, R, a->W, a->H, binning_bytes, out3_host);
    return GOF_OK;
}

int gof_integrate_prepare_points(const GofRasterArgs* a, int32_t PN, const float* points3D, void* point_ws, size_t point_bytes,
                                 uint32_t* num_integrated_host, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = validate(a);
    if (!rc) rc = take_async_status();       // a tile / point sort of an earlier call on this process that timed out is reported here
    if (rc) return rc;
    if (!num_integrated_host) { set_error("num_integrated_host is NULL"); return GOF_E_INVALID; }
    *num_integrated_host = 0;
    if (PN <= 0) return GOF_OK;
    if (!points3D || !point_ws) { set_error("points3D / workspace is NULL"); return GOF_E_INVALID; }
    if (point_bytes < gof_point_bytes(PN)) { set_error("point workspace too small"); return GOF_E_WORKSPACE; }
    PointWs w;
    point_layout(PN, aligned_base(point_ws), &w);
    const Dims d = dims_of(a);
    const Cam cam = { a->viewmatrix, a->projmatrix, a->campos };
    hipLaunchKernelGGL(preprocess_points, dim3((PN + 255) / 256), dim3(256), 0, stream, PN, points3D, cam, a->W, a->H,
                       d.focal_x, d.focal_y, w.pos, w.tiles_touched);
    GOF_LAUNCH_CHECK(stream, a->debug);
    GOF_HIP_CHECK(device_scan_u32(w.tiles_touched, nullptr, w.point_offsets, (size_t)PN, true, w.scan_tmp, nullptr, stream));
    GOF_LAUNCH_CHECK(stream, a->debug);
    uint32_t n = 0;
    GOF_HIP_CHECK(hipMemcpyAsync(&n, w.point_offsets + PN - 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GOF_HIP_CHECK(hipStreamSynchronize(stream));
    *num_integrated_host = n;
    return GOF_OK;
}

// ---- integrate, Gaussian side of one view: binning + pixel pass.  Leaves in the three workspaces everything
// gof_integrate_points needs (records, sorted list, ranges, contributor masks) and in out_color the base image.
int gof_integrate_view(const GofRasterArgs* a, uint32_t R, const int32_t* radii,
                       void* geom_ws, size_t geom_bytes, void* binning_ws, size_t binning_bytes, void* image_ws, size_t image_bytes,
                       float* out_color, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = validate(a);
    if (!rc) rc = take_async_status();       // a tile / point sort of an earlier call on this process that timed out is reported here
    if (rc) return rc;
    if (a->P == 0) return GOF_OK;
    if (!radii || !out_color) { set_error("an output / radii pointer is NULL"); return GOF_E_INVALID; }
    if (geom_bytes < gof_geom_bytes(a->P) || image_bytes < gof_image_bytes(a->W, a->H) || binning_bytes < bin_static_bytes(R, a->W, a->H)) {
        set_error("workspace too small"); return GOF_E_WORKSPACE; }
    GeomWs g; ImageWs im; BinWs b;
    geom_layout(a->P, aligned_base(geom_ws), &g);
    image_layout(a->W, a->H, aligned_base(image_ws), &im);
    bin_layout(R, a->W, a->H, aligned_base(binning_ws), &b, BIN_STATIC_MASKS);
    const Dims d = dims_of(a);
    rc = bin_gaussians(a, d, R, g, b, im, radii, stream);
    if (rc) return rc;
    GOF_PROFILE("integrate_pixels", stream);
    if (mode_on(a->integrate_pixel_pass, g_integrate_pixel_pass)) {      // the pixel-centric form of rounds 1-4 (args->integrate_pixel_pass / gof_set_integrate_pixel_pass(1): A/B, tests)
        hipLaunchKernelGGL(integrate_pixels, dim3(xcd_padded_tiles(d.ntiles)), dim3(TILE_PIX), 0, stream,
                           im.ranges, b.vals, g.rec, g.bbox, g.fconic, a->W, a->H, d.focal_x, d.focal_y, a->background, im.final_T, im.n_contrib,
                           out_color, b.cmask, d.gx, d.ntiles, im.tile_order, im.tile_queue, im.tile_cost);
        GOF_LAUNCH_CHECK(stream, a->debug);
        return GOF_OK;
    }
    // ray-centric: every distinct sub-ray of a tile once (integrate.hip); the tiles it abandons at the reference's 1024-contributor
    // cap (tile_cost == 0xFFFFFFFF) are rendered by the pixel-centric kernel behind it (one workgroup per tile, most return at once)
    hipLaunchKernelGGL(integrate_rays, dim3(xcd_padded_tiles(d.ntiles)), dim3(576), 0, stream,
                       im.ranges, b.vals, g.rec, g.fconic, a->W, a->H, d.focal_x, d.focal_y, a->background, im.final_T, im.n_contrib,
                       out_color, b.cmask, d.gx, d.ntiles, im.tile_order, im.tile_queue, im.tile_cost);
    GOF_LAUNCH_CHECK(stream, a->debug);
    hipLaunchKernelGGL(integrate_pixels_capped, dim3(d.ntiles), dim3(TILE_PIX), 0, stream,
                       im.ranges, b.vals, g.rec, g.bbox, g.fconic, a->W, a->H, d.focal_x, d.focal_y, a->background, im.final_T, im.n_contrib,
                       out_color, b.cmask, d.gx, d.ntiles, im.tile_cost);
    GOF_LAUNCH_CHECK(stream, a->debug);
    return GOF_OK;
}

// ---- integrate, point side: point binning (createWithKeys + sort + ranges, rasterizer_impl.cu:720-752) + point pass ----
static size_t packed_geom_layout(int32_t P, const void* base, const SplatRec** rec, const float** zfront)
{
    const size_t n = (size_t)(P < 1 ? 1 : P);
    const char* p = static_cast<const char*>(base);
    if (rec) *rec = reinterpret_cast<const SplatRec*>(p);
    const size_t off = (n * sizeof(SplatRec) + ALIGN - 1) & ~(size_t)(ALIGN - 1);
    if (zfront) *zfront = reinterpret_cast<const float*>(p + off);
    return off + ((n * sizeof(float) + ALIGN - 1) & ~(size_t)(ALIGN - 1)) + ALIGN;
}
size_t gof_integrate_packed_geom_bytes(int32_t P) { return packed_geom_layout(P, nullptr, nullptr, nullptr) + ALIGN; }

int gof_integrate_pack_geom(const GofRasterArgs* a, const void* geom_ws, size_t geom_bytes, void* packed, size_t packed_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!a || a->P < 0) { set_error("bad args"); return GOF_E_INVALID; }
    if (a->P == 0) return GOF_OK;
    if (!geom_ws || !packed) { set_error("a workspace is NULL"); return GOF_E_INVALID; }
    if (geom_bytes < gof_geom_bytes(a->P) || packed_bytes < gof_integrate_packed_geom_bytes(a->P)) { set_error("workspace too small"); return GOF_E_WORKSPACE; }
    GeomWs g;
    geom_layout(a->P, aligned_base(const_cast<void*>(geom_ws)), &g);
    const SplatRec* rec; const float* zf;
    packed_geom_layout(a->P, aligned_base(packed), &rec, &zf);
    hipLaunchKernelGGL(pack_view_geometry, dim3((a->P + 255) / 256), dim3(256), 0, stream, a->P, g.rec, g.fconic,
                       const_cast<SplatRec*>(rec), const_cast<float*>(zf));
    GOF_LAUNCH_CHECK(stream, a->debug);
    return GOF_OK;
}

static int integrate_points_impl(const GofRasterArgs* a, uint32_t R, int32_t PN, uint32_t NI, bool packed, bool acc_min,
                         const void* geom_ws, size_t geom_bytes, const void* binning_ws, size_t binning_bytes, void* image_ws, size_t image_bytes,
                         void* point_ws, size_t point_bytes, void* point_binning_ws, size_t point_binning_bytes,
                         const float* base_color, float* out_color, float* out_alpha_integrated, float* out_color_integrated, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc = validate(a);
    if (!rc) rc = take_async_status();       // a tile / point sort of an earlier call on this process that timed out is reported here
    if (rc) return rc;
    if (a->P == 0 || PN <= 0) return GOF_OK;     // rasterize_points.cu:301
    // the accumulating variants may leave out the image and the colour
    if (!base_color || !out_alpha_integrated || (!acc_min && (!out_color || !out_color_integrated))) { set_error("an output pointer is NULL"); return GOF_E_INVALID; }
    if (geom_bytes < (packed ? gof_integrate_packed_geom_bytes(a->P) : gof_geom_bytes(a->P)) || image_bytes < gof_image_bytes(a->W, a->H) ||
        binning_bytes < bin_static_bytes(R, a->W, a->H) ||
        point_bytes < gof_point_bytes(PN) || point_binning_bytes < gof_point_binning_bytes(NI, a->W, a->H)) { set_error("workspace too small"); return GOF_E_WORKSPACE; }
    const SplatRec* rec_ptr; const float* zfront_ptr; int zstride;
    if (packed) { packed_geom_layout(a->P, aligned_base(const_cast<void*>(geom_ws)), &rec_ptr, &zfront_ptr); zstride = 1; }
    else { GeomWs gg; geom_layout(a->P, aligned_base(const_cast<void*>(geom_ws)), &gg); rec_ptr = gg.rec;
           zfront_ptr = reinterpret_cast<const float*>(gg.fconic) + 7; zstride = 8; }        // fconic[2i + 1].w
    ImageWs im; BinWs b, pb; PointWs w;
    image_layout(a->W, a->H, aligned_base(image_ws), &im);
    bin_layout(R, a->W, a->H, aligned_base(const_cast<void*>(binning_ws)), &b, BIN_STATIC_MASKS);
    point_layout(PN, aligned_base(point_ws), &w);
    bin_layout(NI, a->W, a->H, aligned_base(point_binning_ws), &pb, BIN_POINTS);
    const Dims d = dims_of(a);
    // one stable sort of the visible points by (tile, pixel of the tile): 3 radix passes (see point_keys)
    { GOF_PROFILE("bin_points", stream);
    if (NI > 0) {
        const int key_bits = (int)higher_msb(d.ntiles) + 8;
        const bool odd = radix_passes(key_bits) & 1;
        uint32_t* t_in = odd ? pb.tiles_alt : pb.tiles;
        uint32_t* v_in = odd ? pb.vals_alt : pb.vals;
        hipLaunchKernelGGL(point_keys, dim3((PN + 255) / 256), dim3(256), 0, stream, PN, w.pos, w.point_offsets, w.tiles_touched,
                           t_in, v_in, d.gx, d.gy);
        GOF_LAUNCH_CHECK(stream, a->debug);
        rc = sort_by_tile(pb, NI, t_in, v_in, key_bits, stream);
        if (rc) return rc;
        GOF_LAUNCH_CHECK(stream, a->debug);
    }
    GOF_HIP_CHECK(hipMemsetAsync(im.point_ranges, 0, (size_t)d.ntiles * sizeof(uint2), stream));
    if (NI > 0) {
        hipLaunchKernelGGL(tile_ranges, dim3(tile_ranges_grid(NI)), dim3(256), 0, stream, NI, pb.tiles, im.point_ranges, 8, nullptr,
                           radix_sort_error_flag(pb.sort_tmp, (size_t)NI, (int)higher_msb(d.ntiles) + 8), async_status_word());
        GOF_LAUNCH_CHECK(stream, a->debug);
        hipLaunchKernelGGL(gather_sorted_points, dim3((NI + 255) / 256), dim3(256), 0, stream, NI, pb.vals, w.pos, pb.pt_xy, pb.pt_depth, a->W, a->H,
                           d.focal_x, d.focal_y);
        GOF_LAUNCH_CHECK(stream, a->debug);
    } }
    // dispatch order of the point pass: #points of the tile x what its pixels walked (tile_cost, left by integrate_pixels), heaviest first
    hipLaunchKernelGGL(order_tiles, dim3(1), dim3(1024), 0, stream, d.ntiles, im.ranges, im.tile_cost, pb.pt_order, pb.pt_queue, im.point_ranges, nullptr, nullptr, nullptr, nullptr);
    GOF_LAUNCH_CHECK(stream, a->debug);
    GOF_PROFILE("integrate_points", stream);
    hipLaunchKernelGGL(integrate_points, dim3(xcd_padded_tiles(d.ntiles)), dim3(TILE_PIX), 0, stream,
                       im.ranges, im.point_ranges, b.vals, pb.vals, rec_ptr, zfront_ptr, zstride, b.cmask, a->W, a->H, pb.tiles, pb.pt_xy, pb.pt_depth, pb.pt_T, pb.pt_acc,
                       base_color, out_color, out_alpha_integrated, out_color_integrated, im.n_contrib, acc_min ? 1 : 0, d.gx, d.ntiles, pb.pt_order, pb.pt_queue);
    GOF_LAUNCH_CHECK(stream, a->debug);
    return GOF_OK;
}

int gof_integrate_points(const GofRasterArgs* a, uint32_t R, int32_t PN, uint32_t NI,
                         const void* geom_ws, size_t geom_bytes, const void* binning_ws, size_t binning_bytes, void* image_ws, size_t image_bytes,
                         void* point_ws, size_t point_bytes, void* point_binning_ws, size_t point_binning_bytes,
                         const float* base_color, float* out_color, float* out_alpha_integrated, float* out_color_integrated, void* stream_)
{
    return integrate_points_impl(a, R, PN, NI, false, false, geom_ws, geom_bytes, binning_ws, binning_bytes, image_ws, image_bytes, point_ws, point_bytes,
                                 point_binning_ws, point_binning_bytes, base_color, out_color, out_alpha_integrated, out_color_integrated, stream_);
}
int gof_integrate_points_packed(const GofRasterArgs* a, uint32_t R, int32_t PN, uint32_t NI,
                         const void* packed_geom, size_t packed_bytes, const void* binning_ws, size_t binning_bytes, void* image_ws, size_t image_bytes,
                         void* point_ws, size_t point_bytes, void* point_binning_ws, size_t point_binning_bytes,
                         const float* base_color, float* out_color, float* out_alpha_integrated, float* out_color_integrated, void* stream_)
{
    return integrate_points_impl(a, R, PN, NI, true, false, packed_geom, packed_bytes, binning_ws, binning_bytes, image_ws, image_bytes, point_ws, point_bytes,
                                 point_binning_ws, point_binning_bytes, base_color, out_color, out_alpha_integrated, out_color_integrated, stream_);
}

int gof_integrate_points_min(const GofRasterArgs* a, uint32_t R, int32_t PN, uint32_t NI, int32_t packed,
                             const void* geom_ws, size_t geom_bytes, const void* binning_ws, size_t binning_bytes, void* image_ws, size_t image_bytes,
                             void* point_ws, size_t point_bytes, void* point_binning_ws, size_t point_binning_bytes,
                             const float* base_color, float* out_color, float* alpha_min_inout, float* color_min_inout, void* stream_)
{
    return integrate_points_impl(a, R, PN, NI, packed != 0, true, geom_ws, geom_bytes, binning_ws, binning_bytes, image_ws, image_bytes, point_ws,
                                 point_bytes, point_binning_ws, point_binning_bytes, base_color, out_color, alpha_min_inout, color_min_inout, stream_);
}

int gof_integrate_run(const GofRasterArgs* a, uint32_t R, const int32_t* radii, int32_t PN, uint32_t NI,
                      void* geom_ws, size_t geom_bytes, void* binning_ws, size_t binning_bytes, void* image_ws, size_t image_bytes,
                      void* point_ws, size_t point_bytes, void* point_binning_ws, size_t point_binning_bytes,
                      float* out_color, float* out_alpha_integrated, float* out_color_integrated, void* stream_)
{
    if (a && (a->P == 0 || PN <= 0)) { int rc = validate(a); return rc; }     // rasterize_points.cu:301: nothing is launched
    int rc = gof_integrate_view(a, R, radii, geom_ws, geom_bytes, binning_ws, binning_bytes, image_ws, image_bytes, out_color, stream_);
    if (rc) return rc;
    return gof_integrate_points(a, R, PN, NI, geom_ws, geom_bytes, binning_ws, binning_bytes, image_ws, image_bytes, point_ws, point_bytes,
                                point_binning_ws, point_binning_bytes, out_color, out_color, out_alpha_integrated, out_color_integrated, stream_);
}

int gof_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P < 0) { set_error("bad P"); return GOF_E_INVALID; }
    if (P == 0) return GOF_OK;
    if (!means3D || !viewmatrix || !projmatrix || !present) { set_error("a pointer is NULL"); return GOF_E_INVALID; }
    const Cam cam = { viewmatrix, projmatrix, nullptr };
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, cam, present);
    GOF_LAUNCH_CHECK(stream, 0);
    return GOF_OK;
}

// ---- data-parallel training: compressed SH-gradient exchange (preprocess.hip) ---------------------------
int gof_sh_grad_pack(int32_t P, const float* dL_dcolors, const void* geom_ws, size_t geom_bytes, const int32_t* radii, float* packed, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P < 0) { set_error("bad P"); return GOF_E_INVALID; }
    if (P == 0) return GOF_OK;
    if (!dL_dcolors || !geom_ws || !radii || !packed) { set_error("a pointer is NULL"); return GOF_E_INVALID; }
    if (geom_bytes < gof_geom_bytes_forward(P)) { set_error("geometry workspace too small for P = %d", P); return GOF_E_WORKSPACE; }
    GeomWs g;
    geom_layout(P, aligned_base(const_cast<void*>(geom_ws)), &g);
    GOF_PROFILE("sh_grad_pack", stream);
    hipLaunchKernelGGL(sh_grad_pack, dim3((P + 255) / 256), dim3(256), 0, stream, P, dL_dcolors, g.clamped, radii, packed);
    GOF_LAUNCH_CHECK(stream, 0);
    return GOF_OK;
}

int gof_sh_grad_expand(int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D, const float* campos, int64_t campos_stride,
                       const float* packed, int64_t packed_stride, float scale, float* out_dc, int64_t stride_dc, float* out_rest,
                       int64_t stride_rest, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P < 0 || n_views < 1) { set_error("bad P / n_views (%d, %d)", P, n_views); return GOF_E_INVALID; }
    if (D < 0 || D > 3 || M < 1 || M > 16 || (D + 1) * (D + 1) > M) { set_error("bad SH degree / coefficient count (%d, %d)", D, M); return GOF_E_INVALID; }
    if (P == 0) return GOF_OK;
    if (!means3D || !campos || !packed || !out_dc || (M > 1 && !out_rest)) { set_error("a pointer is NULL"); return GOF_E_INVALID; }
    if (n_views > 1 && (packed_stride < (int64_t)3 * P || campos_stride < 3)) { set_error("view strides too small"); return GOF_E_INVALID; }
    GOF_PROFILE("sh_grad_expand", stream);
    const dim3 grid((uint32_t)((P + 255) / 256));
    const size_t lds = 256 * (3 * M + 1) * sizeof(float);
    if (M == 16)
        hipLaunchKernelGGL(sh_grad_expand<16>, grid, dim3(256), lds, stream, P, D, M, n_views, means3D, campos, (long)campos_stride, packed,
                           (long)packed_stride, scale, out_dc, (long)stride_dc, out_rest, (long)stride_rest);
    else
        hipLaunchKernelGGL(sh_grad_expand<0>, grid, dim3(256), lds, stream, P, D, M, n_views, means3D, campos, (long)campos_stride, packed,
                           (long)packed_stride, scale, out_dc, (long)stride_dc, out_rest, (long)stride_rest);
    GOF_LAUNCH_CHECK(stream, 0);
    return GOF_OK;
}

// ---- per-kernel timing ----------------------------------------------------------------------------------
int gof_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    g_prof_on = on != 0;
    return GOF_OK;
}
// Synchronises the recorded events, aggregates per kernel name and writes a JSON object
// {"name": {"calls": n, "total_ms": t}, ...} into buf (NUL-terminated); clears the records.
int gof_profile_report(char* buf, size_t cap)
{
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    struct Agg { const char* name; int calls; double ms; };
    std::vector<Agg> aggs;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            bool found = false;
            for (auto& a : aggs) if (!strcmp(a.name, r.name)) { a.calls++; a.ms += ms; found = true; break; }
            if (!found) aggs.push_back(Agg{ r.name, 1, (double)ms });
        }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    std::string s = "{";
    for (size_t i = 0; i < aggs.size(); i++) {
        char tmp[256];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"calls\": %d, \"total_ms\": %.6f}", i ? ", " : "", aggs[i].name, aggs[i].calls, aggs[i].ms);
        s += tmp;
    }
    s += "}";
    if (!buf || cap < s.size() + 1) { set_error("profile buffer too small"); return GOF_E_INVALID; }
    memcpy(buf, s.c_str(), s.size() + 1);
    return GOF_OK;
}

// ---- introspection ------------------------------------------------------------------------------------
} // extern "C"

namespace gof {
__global__ void unpack_rec(int P, const SplatRec* __restrict__ rec, const float4* __restrict__ conic, const uint8_t* __restrict__ clamped,
                           int what, float* __restrict__ dstf, uint8_t* __restrict__ dstb)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float* f = rec[idx].f;
    switch (what) {
    case 0: dstf[2 * idx] = f[REC_XY]; dstf[2 * idx + 1] = f[REC_XY + 1]; break;                                  // means2D
    case 1: { const float4 c = conic[idx]; dstf[4 * idx] = c.x; dstf[4 * idx + 1] = c.y; dstf[4 * idx + 2] = c.z; dstf[4 * idx + 3] = f[REC_W]; } break;
    case 2: dstf[3 * idx] = f[REC_RGB]; dstf[3 * idx + 1] = f[REC_RGB + 1]; dstf[3 * idx + 2] = f[REC_RGB + 2]; break;
    case 3: for (int i = 0; i < 10; i++) dstf[10 * idx + i] = f[i]; break;
    case 4: { const uint8_t b = clamped[idx]; dstb[3 * idx] = b & 1; dstb[3 * idx + 1] = (b >> 1) & 1; dstb[3 * idx + 2] = (b >> 2) & 1; } break;
    }
}
}

// per tile: number of contributing (pixel, list entry) pairs = set bits of the contributor masks blend_forward left, over the
// words the backward reads (positions below the tile's last contributor)
__global__ void __launch_bounds__(256)
count_contributing_pairs(const uint2* __restrict__ ranges, const uint32_t* __restrict__ n_contrib, const MaskPool masks,
                         int W, int H, uint32_t gx, uint32_t ntiles, uint32_t* __restrict__ out, int hash)
{
    // hash != 0: a position-sensitive checksum of the same words instead of their bit count (two forward modes that agree on it agree
    // on every contributor bit the backward reads)
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    uint32_t lx, ly;
    tile_pixel(tid, lx, ly);
    const uint32_t px = (tile % gx) * TILE_X + lx, py = (tile / gx) * TILE_Y + ly;
    const uint32_t last = (px < (uint32_t)W && py < (uint32_t)H) ? n_contrib[(size_t)W * py + px] : 0u;
    __shared__ uint32_t s_max, s_sum;
    if (tid == 0) { s_max = 0; s_sum = 0; }
    __syncthreads();
    atomicMax(&s_max, last);
    __syncthreads();
    const uint2 range = ranges[tile];
    const uint32_t max_last = min(s_max, range.y - range.x);
    const uint32_t* entry = masks.table + mask_slot0(range.x, tile);
    uint32_t c = 0;
    for (uint32_t w = 0; w < (max_last + 31) / 32; w++) {
        const uint32_t chunk = entry[w >> 3];
        uint32_t word = (chunk == POOL_NONE) ? 0u : masks.pool[((size_t)chunk * 4u + (tid >> 6)) * MASK_SUBCHUNK_WORDS + (w & 7u) * 64u + (tid & 63u)];
        if (w == max_last / 32 && (max_last & 31u)) word &= (1u << (max_last & 31u)) - 1u;      // (positions the backward never stages)
        c += hash ? word * (2u * (w * TILE_PIX + tid) + 1u) + (word >> 7) : (uint32_t)__popc(word);
    }
    atomicAdd(&s_sum, c);
    __syncthreads();
    if (tid == 0) out[tile] = s_sum;
}

extern "C" int64_t gof_debug_fetch(const char* name, const GofRasterArgs* a, uint32_t R, const void* geom_ws, const void* binning_ws,
                                   const void* image_ws, void* dst, size_t dst_bytes, void* stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!name || !a || !dst) { set_error("NULL argument"); return GOF_E_INVALID; }
    GeomWs g; ImageWs im; BinWs b;
    const size_t P = (size_t)a->P, HW = (size_t)a->W * a->H;
    const Dims d = dims_of(a);
    if (geom_ws) geom_layout(a->P, aligned_base(geom_ws), &g);
    if (image_ws) image_layout(a->W, a->H, aligned_base(image_ws), &im);
    if (binning_ws) bin_layout(R, a->W, a->H, aligned_base(binning_ws), &b, BIN_MASK_POOL);      // (the sorted lists lie at the same offsets in every mode)
    const void* src = nullptr; size_t bytes = 0; int64_t count = 0;
    int unpack = -1; size_t per = 0; bool bytes_out = false;
    const std::string n(name);
    if (n == "depths" && geom_ws) { src = g.depths; count = P; bytes = P * 4; }
    else if (n == "tiles_touched" && geom_ws) { src = g.tiles_touched; count = P; bytes = P * 4; }
    else if (n == "means2D" && geom_ws) { unpack = 0; per = 2; }
    else if (n == "conic_opacity" && geom_ws) { unpack = 1; per = 4; }
    else if (n == "rgb" && geom_ws) { unpack = 2; per = 3; }
    else if (n == "view2gaussian" && geom_ws) { unpack = 3; per = 10; }
    else if (n == "clamped" && geom_ws) { unpack = 4; per = 3; bytes_out = true; }
    else if (n == "point_list" && binning_ws) { src = b.vals; count = R; bytes = (size_t)R * 4; }
    else if (n == "point_list_keys" && binning_ws && geom_ws) {
        if (dst_bytes < (size_t)R * 8) { set_error("dst too small"); return GOF_E_INVALID; }
        if (R) hipLaunchKernelGGL(rebuild_keys, dim3((R + 255) / 256), dim3(256), 0, stream, R, b.tiles, b.vals, g.depths, static_cast<uint64_t*>(dst));
        if (hipGetLastError() != hipSuccess) { set_error("rebuild_keys launch failed"); return GOF_E_DEVICE; }
        return (int64_t)R;
    }
    else if ((n == "contrib_pairs" || n == "contrib_hash") && binning_ws && image_ws) {
        if (dst_bytes < (size_t)d.ntiles * 4) { set_error("dst too small"); return GOF_E_INVALID; }
        hipLaunchKernelGGL(count_contributing_pairs, dim3(d.ntiles), dim3(256), 0, stream, im.ranges, im.n_contrib, b.mp, a->W, a->H, d.gx, d.ntiles,
                           static_cast<uint32_t*>(dst), n == "contrib_hash" ? 1 : 0);
        if (hipGetLastError() != hipSuccess) { set_error("count_contributing_pairs launch failed"); return GOF_E_DEVICE; }
        return (int64_t)d.ntiles;
    }
    else if (n == "ranges" && image_ws) { src = im.ranges; count = 2 * (int64_t)d.ntiles; bytes = (size_t)d.ntiles * 8; }
    else if (n == "point_ranges" && image_ws) { src = im.point_ranges; count = 2 * (int64_t)d.ntiles; bytes = (size_t)d.ntiles * 8; }
    else if (n == "tile_cost" && image_ws) { src = im.tile_cost; count = (int64_t)d.ntiles; bytes = (size_t)d.ntiles * 4; }
    else if (n == "tile_order" && image_ws) { src = im.tile_order; count = (int64_t)NXCD * tile_queue_stride(d.ntiles); bytes = (size_t)count * 4; }
    else if (n == "tile_order_bw" && image_ws) { src = im.tile_order_bw; count = (int64_t)NXCD * tile_queue_stride(d.ntiles); bytes = (size_t)count * 4; }
    else if (n == "tile_queue" && image_ws) { src = im.tile_queue; count = TILE_QUEUE_WORDS; bytes = (size_t)count * 4; }
    else if (n == "final_T" && image_ws) { src = im.final_T; count = 4 * HW; bytes = 4 * HW * 4; }
    else if (n == "n_contrib" && image_ws) { src = im.n_contrib; count = 2 * HW; bytes = 2 * HW * 4; }
    else { set_error("unknown array '%s' (or its workspace is NULL)", name); return GOF_E_INVALID; }
    if (unpack >= 0) {
        count = (int64_t)(P * per);
        bytes = P * per * (bytes_out ? 1 : 4);
        if (dst_bytes < bytes) { set_error("dst too small"); return GOF_E_INVALID; }
        if (P) hipLaunchKernelGGL(unpack_rec, dim3((a->P + 255) / 256), dim3(256), 0, stream, a->P, g.rec, g.conic, g.clamped, unpack,
                                  static_cast<float*>(dst), static_cast<uint8_t*>(dst));
        if (hipGetLastError() != hipSuccess) { set_error("unpack launch failed"); return GOF_E_DEVICE; }
        return count;
    }
    if (dst_bytes < bytes) { set_error("dst too small"); return GOF_E_INVALID; }
    if (bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) { set_error("copy failed"); return GOF_E_DEVICE; }
    return count;
}
