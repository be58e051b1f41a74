// tests/hipemu: a host-side stand-in for <hip/hip_runtime.h> -- TEST INFRASTRUCTURE, never part of the product.
//
// The product's kernels (gaussian-opacity-fields_amd/csrc/*.hip) are written for gfx950 only.  This header lets the SAME source
// files be compiled as plain host C++ (clang++ -x c++ -I tests/hipemu/include) into tests/hipemu/_build/libgof_hip_emu.so, in which
// a kernel launch runs every workgroup as a set of cooperative fibers in wave64 lock step:
//   * a workgroup = blockDim fibers on one OS thread; __shared__ = static thread_local; __syncthreads = fiber barrier;
//   * cross-lane operations (ballot, shfl*, DPP, ds_bpermute, readlane, permlane{16,32}_swap) = exchange through a per-wave buffer at a
//     wave barrier, with the hardware's lane semantics; a wave whose lanes reach DIFFERENT cross-lane sites (a divergence the
//     hardware would resolve by exec masks, which fibers cannot) is reported as a deadlock with the waiting sites;
//   * v_rcp_f32 / v_rsq_f32 / v_exp_f32 are replaced by correctly rounded host functions (the product uses them only where results are
//     compared with a tolerance), everything else is the same IEEE fp32 / fp64 arithmetic (-ffp-contract=off, fmaf where written).
// What it is for: running the kernels' logic against the oracle in the CPU test suite (tests/test_hipemu_*.py), under
// AddressSanitizer / UBSan if wanted, and developing kernel changes without a GPU.  What it is NOT: a CPU path of the product --
// nothing under gaussian-opacity-fields_amd/ knows about it, and timing it says nothing.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <sched.h>

#define HIPEMU 1
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __constant__ static const
#define HIP_SYMBOL(x) (&(x))
#define amdgpu_waves_per_eu(...) __unused__          /* (a target attribute the host compiler rejects) */
#define amdgpu_flat_work_group_size(...) __unused__
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5

// ---- vector types ---------------------------------------------------------------------------------------------------------
struct alignas(8) float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct int3 { int x, y, z; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
struct alignas(16) longlong2 { long long x, y; };
struct alignas(32) longlong4 { long long x, y, z, w; };
struct alignas(32) ulonglong4 { unsigned long long x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{ x, y }; }
inline float3 make_float3(float x, float y, float z) { return float3{ x, y, z }; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{ x, y, z, w }; }
inline int2 make_int2(int x, int y) { return int2{ x, y }; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{ x, y, z, w }; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{ x, y }; }
inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return uint3{ x, y, z }; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{ x, y, z, w }; }
inline double2 make_double2(double x, double y) { return double2{ x, y }; }
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- runtime types ----------------------------------------------------------------------------------------------------------
typedef void* hipStream_t;
struct hipemuEvent { std::chrono::steady_clock::time_point t; };
typedef hipemuEvent* hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
constexpr unsigned hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipEventDisableTiming = 2, hipEventDefault = 0;
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
#define hipStreamNonBlocking 1u
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }      // (launches run synchronously: one stream is every stream)
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpyFromSymbol(void* d, const void* sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) { std::memcpy(d, (const char*)sym + off, n); return hipSuccess; }
inline hipError_t hipMemcpyToSymbol(const void* sym, const void* s, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) { std::memcpy((char*)const_cast<void*>(sym) + off, s, n); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
template <class T> inline hipError_t hipHostGetDevicePointer(T** d, void* h, unsigned f) { return hipHostGetDevicePointer((void**)d, h, f); }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemuEvent; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }      // (a small "device": bounded grids stride)
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

// ---- execution model (hipemu_rt.cpp) ------------------------------------------------------------------------------------------
namespace hipemu {
// cross-lane scopes: an operation whose data stays inside a quad (DPP quad_perm) or a 16-lane row (DPP row_*) synchronises only
// that group -- the hardware lets quads / rows of a wave sit in different iterations of a divergent loop (exec mask) as long as the
// lanes an operation reads from take part; the wave scope is for everything else
enum Scope { SCOPE_QUAD = 0, SCOPE_ROW = 1, SCOPE_WAVE = 2 };
struct Fiber {
    void* sp;
    uint3 tid;
    unsigned flat, wave, lane;
    bool done;
    const char* wait_what;
};
struct ScopeState {
    unsigned arrived[16], gen[16], parity[16];     // per group of the scope (16 quads / 4 rows / the wave)
    uint64_t active[2];
    uint64_t xchg[2][64];
    const char* site[64];
};
struct Wave {
    uint64_t alive_mask;
    uint64_t at_barrier;          // lanes waiting in __syncthreads
    ScopeState sc[3];
};
struct Block {
    dim3 grid, dim, bid;
    unsigned nthreads, nwaves, alive, cur;
    unsigned bar_arrived, bar_gen;
    int acc_and, acc_or, acc_count, res_and[2], res_or[2], res_count[2];
    unsigned spin, rng;
    Fiber* fibers;
    Wave* waves;
    void* dyn_lds;
    const std::function<void()>* body;
    void* sched_sp;
    const char* kernel_name;
};
extern thread_local Block* t_block;
extern thread_local Fiber* t_fiber;
void launch(const char* name, dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void yield();
void block_barrier(int pred, int* r_and, int* r_or);
void block_barrier_count(int pred, int* r_count);
// every alive lane of the wave deposits `mine`; returns the wave's 64 deposits (valid until the lane's next cross-lane operation)
// and the mask of lanes that took part
const uint64_t* wave_exchange(uint64_t mine, uint64_t* active, const char* what, Scope scope = SCOPE_WAVE);
inline void* dynamic_lds() { return t_block->dyn_lds; }

template <class T> inline uint64_t to_bits(T v) { static_assert(sizeof(T) <= 8, "cross-lane value wider than 64 bits"); uint64_t b = 0; std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }
template <class T> inline T shfl_from(T v, int src, const char* what)
{
    uint64_t act;
    const uint64_t* x = wave_exchange(to_bits(v), &act, what);
    return from_bits<T>(x[src & 63]);
}
}   // namespace hipemu

#define threadIdx (hipemu::t_fiber->tid)
#define blockIdx (hipemu::t_block->bid)
#define blockDim (hipemu::t_block->dim)
#define gridDim (hipemu::t_block->grid)
constexpr int warpSize = 64;

#define HIPEMU_STR2(x) #x
#define HIPEMU_STR(x) HIPEMU_STR2(x)
#define HIPEMU_AT(op) op " @ " __FILE__ ":" HIPEMU_STR(__LINE__)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(#kernel, dim3(grid), dim3(block), (size_t)(shmem), [&]() { kernel(__VA_ARGS__); })

inline void hipemu_syncthreads() { hipemu::block_barrier(1, nullptr, nullptr); }
inline int hipemu_syncthreads_and(int p) { int r; hipemu::block_barrier(p, &r, nullptr); return r; }
inline int hipemu_syncthreads_or(int p) { int r; hipemu::block_barrier(p, nullptr, &r); return r; }
#define __syncthreads() hipemu_syncthreads()
#define __syncthreads_and(p) hipemu_syncthreads_and(p)
#define __syncthreads_or(p) hipemu_syncthreads_or(p)
inline int hipemu_syncthreads_count(int p) { int r; hipemu::block_barrier_count(p, &r); return r; }
#define __syncthreads_count(p) hipemu_syncthreads_count(p)
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }

// ---- cross-lane -----------------------------------------------------------------------------------------------------------------
inline unsigned long long hipemu_ballot(int pred, const char* what)
{
    uint64_t act;
    const uint64_t* x = hipemu::wave_exchange(pred ? 1u : 0u, &act, what);
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) if (((act >> l) & 1u) && x[l]) m |= 1ull << l;
    return m;
}
#define __ballot(p) hipemu_ballot((p), HIPEMU_AT("ballot"))
// The hardware executes a wave's lanes instruction by instruction; fibers run a lane until its next barrier / cross-lane operation.
// Where a kernel relies on that lock step WITHOUT a cross-lane operation (all lanes read an LDS word, then one of them writes it),
// build_emu.py inserts this wave barrier into its scratch copy of the source (LOCKSTEP_POINTS there lists every such place).
inline void hipemu_wave_sync(const char* what) { uint64_t act; (void)hipemu::wave_exchange(0, &act, what); }
#define HIPEMU_WAVE_SYNC() hipemu_wave_sync(HIPEMU_AT("lock-step point"))
// a wave exchanging data through its own LDS panel: on the hardware the LDS serves a wave's instructions in order and the builtin
// is a compiler barrier only; here it is where the wave's fibers meet
#define __builtin_amdgcn_wave_barrier() hipemu_wave_sync(HIPEMU_AT("wave_barrier"))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
template <class T> inline T hipemu_shfl(T v, int src, int width, const char* what)
{
    const int lane = (int)hipemu::t_fiber->lane;
    return hipemu::shfl_from(v, (src & (width - 1)) + (lane & ~(width - 1)), what);
}
template <class T> inline T hipemu_shfl_xor(T v, int mask, int width, const char* what)
{
    const int lane = (int)hipemu::t_fiber->lane;
    int src = lane ^ mask;
    if (src >= ((lane + width) & ~(width - 1))) src = lane;
    return hipemu::shfl_from(v, src, what);
}
template <class T> inline T hipemu_shfl_up(T v, unsigned d, int width, const char* what)
{
    const int lane = (int)hipemu::t_fiber->lane;
    int src = lane - (int)d;
    if (src < (lane & ~(width - 1))) src = lane;
    return hipemu::shfl_from(v, src, what);
}
template <class T> inline T hipemu_shfl_down(T v, unsigned d, int width, const char* what)
{
    const int lane = (int)hipemu::t_fiber->lane;
    int src = lane + (int)d;
    if (src >= ((lane + width) & ~(width - 1)) || src > 63) src = lane;
    return hipemu::shfl_from(v, src, what);
}
#define HIPEMU_SHFL_SEL(_1, _2, _3, NAME, ...) NAME
#define __shfl(...) HIPEMU_SHFL_SEL(__VA_ARGS__, HIPEMU_SHFL3, HIPEMU_SHFL2)(hipemu_shfl, __VA_ARGS__)
#define __shfl_xor(...) HIPEMU_SHFL_SEL(__VA_ARGS__, HIPEMU_SHFL3, HIPEMU_SHFL2)(hipemu_shfl_xor, __VA_ARGS__)
#define __shfl_up(...) HIPEMU_SHFL_SEL(__VA_ARGS__, HIPEMU_SHFL3, HIPEMU_SHFL2)(hipemu_shfl_up, __VA_ARGS__)
#define __shfl_down(...) HIPEMU_SHFL_SEL(__VA_ARGS__, HIPEMU_SHFL3, HIPEMU_SHFL2)(hipemu_shfl_down, __VA_ARGS__)
#define HIPEMU_SHFL2(f, v, s) f((v), (s), 64, HIPEMU_AT(#f))
#define HIPEMU_SHFL3(f, v, s, w) f((v), (s), (w), HIPEMU_AT(#f))

inline int hipemu_readlane(int v, int lane, const char* what) { return hipemu::shfl_from(v, lane, what); }
#define __builtin_amdgcn_readlane(v, l) hipemu_readlane((v), (l), HIPEMU_AT("readlane"))
inline int hipemu_readfirstlane(int v, const char* what)
{
    uint64_t act;
    const uint64_t* x = hipemu::wave_exchange(hipemu::to_bits(v), &act, what);
    return hipemu::from_bits<int>(x[__builtin_ctzll(act)]);
}
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane((v), HIPEMU_AT("readfirstlane"))
inline int hipemu_ds_bpermute(int addr, int v, const char* what) { return hipemu::shfl_from(v, (addr >> 2) & 63, what); }
#define __builtin_amdgcn_ds_bpermute(a, v) hipemu_ds_bpermute((a), (v), HIPEMU_AT("ds_bpermute"))

// DPP (data-parallel primitives) source-lane selection of gfx9: returns -1 when the control selects no lane (out of row / wave)
inline int hipemu_dpp_src(int l, unsigned ctrl)
{
    const int row = l & ~15, i = l & 15;
    if (ctrl <= 0xFF) return (l & ~3) | (int)((ctrl >> (2 * (l & 3))) & 3u);                     // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; return i + n < 16 ? l + n : -1; }      // row_shl
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; return i >= n ? l - n : -1; }          // row_shr
    if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; return row | ((i - n) & 15); }         // row_ror
    switch (ctrl) {
    case 0x130: return l + 1 < 64 ? l + 1 : -1;      // wave_shl:1
    case 0x134: return (l + 1) & 63;                 // wave_rol:1
    case 0x138: return l >= 1 ? l - 1 : -1;          // wave_shr:1
    case 0x13C: return (l - 1) & 63;                 // wave_ror:1
    case 0x140: return row | (15 - i);               // row_mirror
    case 0x141: return (l & ~7) | (7 - (l & 7));     // row_half_mirror
    case 0x142: return l >= 16 ? row - 1 : -1;       // row_bcast:15
    case 0x143: return l >= 32 ? 31 : -1;            // row_bcast:31
    }
    std::fprintf(stderr, "hipemu: DPP control 0x%x not modelled\n", ctrl);
    std::abort();
}
inline int hipemu_update_dpp(int old, int src, unsigned ctrl, unsigned row_mask, unsigned bank_mask, bool bound_ctrl, const char* what)
{
    uint64_t act;
    const hipemu::Scope scope = ctrl <= 0xFF ? hipemu::SCOPE_QUAD : ((ctrl >= 0x101 && ctrl <= 0x12F) || ctrl == 0x140 || ctrl == 0x141) ? hipemu::SCOPE_ROW : hipemu::SCOPE_WAVE;
    const uint64_t* x = hipemu::wave_exchange(hipemu::to_bits(src), &act, what, scope);
    const int l = (int)hipemu::t_fiber->lane;
    if (!((row_mask >> (l >> 4)) & 1u) || !((bank_mask >> ((l >> 2) & 3)) & 1u)) return old;
    const int s = hipemu_dpp_src(l, ctrl);
    if (s < 0 || !((act >> s) & 1u)) return bound_ctrl ? 0 : old;
    return hipemu::from_bits<int>(x[s]);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc), HIPEMU_AT("update_dpp"))

typedef unsigned hipemu_v2u __attribute__((ext_vector_type(2)));
// v_permlane32_swap vdst, src0: lanes 32-63 of vdst <-> lanes 0-31 of src0; returns {vdst, src0} after the swap
inline hipemu_v2u hipemu_permlane32_swap(unsigned vdst, unsigned src0, const char* what)
{
    uint64_t act;
    const uint64_t* x = hipemu::wave_exchange((uint64_t)vdst | ((uint64_t)src0 << 32), &act, what);
    const int l = (int)hipemu::t_fiber->lane;
    hipemu_v2u r = { vdst, src0 };
    if (l >= 32) r.x = (unsigned)(x[l - 32] >> 32);      // vdst[l] = old src0[l - 32]
    else r.y = (unsigned)x[l + 32];                      // src0[l] = old vdst[l + 32]
    return r;
}
// v_permlane16_swap vdst, src0: odd rows (16 lanes) of vdst <-> even rows of src0
inline hipemu_v2u hipemu_permlane16_swap(unsigned vdst, unsigned src0, const char* what)
{
    uint64_t act;
    const uint64_t* x = hipemu::wave_exchange((uint64_t)vdst | ((uint64_t)src0 << 32), &act, what);
    const int l = (int)hipemu::t_fiber->lane;
    hipemu_v2u r = { vdst, src0 };
    if ((l >> 4) & 1) r.x = (unsigned)(x[l - 16] >> 32);
    else r.y = (unsigned)x[l + 16];
    return r;
}
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipemu_permlane32_swap((a), (b), HIPEMU_AT("permlane32_swap"))
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) hipemu_permlane16_swap((a), (b), HIPEMU_AT("permlane16_swap"))

// ---- scalar builtins ---------------------------------------------------------------------------------------------------------------
#define __builtin_amdgcn_alignbit(hi, lo, s) ((unsigned)(((((uint64_t)(unsigned)(hi)) << 32) | (uint64_t)(unsigned)(lo)) >> ((s) & 31)))
inline unsigned hipemu_bitop3(unsigned a, unsigned b, unsigned c, unsigned tbl)
{
    unsigned r = 0;
    for (int i = 0; i < 32; i++) {
        const unsigned idx = (((a >> i) & 1u) << 2) | (((b >> i) & 1u) << 1) | ((c >> i) & 1u);
        r |= ((tbl >> idx) & 1u) << i;
    }
    return r;
}
#define __builtin_amdgcn_bitop3_b32(a, b, c, t) hipemu_bitop3((a), (b), (c), (t))
inline int hipemu_sbfe(int v, unsigned off, unsigned width)
{
    off &= 31; width &= 31;
    if (width == 0) return 0;
    if (off + width > 32) width = 32 - off;
    return (int)((unsigned)v << (32 - off - width)) >> (32 - width);
}
#define __builtin_amdgcn_sbfe(v, o, w) hipemu_sbfe((v), (o), (w))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_rsqf(x) ((float)(1.0 / std::sqrt((double)(x))))
#define __builtin_amdgcn_exp2f(x) (exp2f(x))
// s_sleep inside a poll loop (the sorts' bounded decoupled look-back): the workgroup polled for runs on another OS thread.  On a busy
// host that thread may be off the CPU for longer than 2^18 bare sched_yield() calls last (seen once: the fiber-order test failed
// while two other 8-thread jobs ran) -- so after a streak of 32768 polls (~10 ms of yielding on an idle core: ordinary waits never get
// there) every further one sleeps 10 us: the poll limit then stands for seconds of waiting, as it does for milliseconds on the GPU.
inline void hipemu_s_sleep()
{
    static thread_local unsigned streak = 0;
    static thread_local std::chrono::steady_clock::time_point last;
    const auto now = std::chrono::steady_clock::now();
    if (now - last > std::chrono::milliseconds(5)) streak = 0;
    last = now;
    if (++streak < 32768u) sched_yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(10));
}
#define __builtin_amdgcn_s_sleep(n) hipemu_s_sleep()
// hwreg(HW_REG_XCC_ID): the dispatcher deals consecutive workgroups round-robin over the 8 XCDs
#define __builtin_amdgcn_s_getreg(r) (hipemu::t_block->bid.x & 7u)
inline unsigned long long wall_clock64() { return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10; }
inline unsigned long long clock64() { return wall_clock64(); }

inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline long long __double_as_longlong(double d) { long long u; std::memcpy(&u, &d, 8); return u; }
inline double __longlong_as_double(long long u) { double d; std::memcpy(&d, &u, 8); return d; }
inline float __logf(float x) { return logf(x); }
inline float __expf(float x) { return expf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return (float)(1.0 / std::sqrt((double)x)); }
inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }

// min / max as HIP's device headers overload them
#define HIPEMU_MINMAX(T) inline T min(T a, T b) { return b < a ? b : a; } inline T max(T a, T b) { return a < b ? b : a; }
HIPEMU_MINMAX(int) HIPEMU_MINMAX(unsigned) HIPEMU_MINMAX(long) HIPEMU_MINMAX(unsigned long) HIPEMU_MINMAX(long long) HIPEMU_MINMAX(unsigned long long)
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
inline unsigned min(unsigned a, int b) { return min(a, (unsigned)b); }
inline unsigned min(int a, unsigned b) { return min((unsigned)a, b); }
inline unsigned max(unsigned a, int b) { return max(a, (unsigned)b); }
inline unsigned max(int a, unsigned b) { return max((unsigned)a, b); }

// ---- atomics (workgroups of one launch run on several OS threads) -----------------------------------------------------------------------
template <class T> inline T atomicAdd(T* p, T v)
{
    if constexpr (std::is_floating_point<T>::value) {
        using U = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
        U* q = reinterpret_cast<U*>(p);
        U old = __atomic_load_n(q, __ATOMIC_RELAXED);
        for (;;) {
            T o; std::memcpy(&o, &old, sizeof(T));
            const T n = o + v;
            U nb; std::memcpy(&nb, &n, sizeof(T));
            if (__atomic_compare_exchange_n(q, &old, nb, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) return o;
        }
    } else {
        return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
    }
}
inline unsigned atomicAdd(unsigned* p, int v) { return __atomic_fetch_add(p, (unsigned)v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, int v) { return __atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned v) { return __atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED); return cmp; }
template <class T> inline T atomicMax(T* p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
    return old;
}
template <class T> inline T atomicMin(T* p, T v)
{
    T old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {}
    return old;
}
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
