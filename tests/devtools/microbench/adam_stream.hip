// Developer microbenchmark (round 6): how fast can the Adam step's 4-read / 3-write stream go on one MI355X?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/adam_stream tests/devtools/microbench/adam_stream.hip && /tmp/adam_stream [elements]
// Variants of csrc/train_epilogue.hip: adam_kernel (V0 = the shipped body), interleaved rounds, minimum and median per variant, GB/s on
// 28 B per element (p, g, m, v read; p, m, v written).  "copy" = a float4 copy of one tensor to another (the guide's 6.29 TB/s figure).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float w1, float beta2, float w2, float step_size, float bc2_sqrt, float eps)
{
    m = m + w1 * (g - m);
    v = v * beta2;
    v = v + (w2 * g) * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p + (step_size * m) / denom;
}

template <int NT_LOAD, int NT_STORE>
struct Mem {
    static __device__ __forceinline__ v4f ld(const float* p) { return NT_LOAD ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)) : *reinterpret_cast<const v4f*>(p); }
    static __device__ __forceinline__ void st(float* p, v4f x) { if (NT_STORE) __builtin_nontemporal_store(x, reinterpret_cast<v4f*>(p)); else *reinterpret_cast<v4f*>(p) = x; }
};

// ITEMS float4 per tensor and thread; PERSIST: grid-stride over the tiles
template <int ITEMS, int NT_LOAD, int NT_STORE, int NT_GRAD, int PERSIST>
__global__ void __launch_bounds__(256)
adam_variant(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ M, float* __restrict__ V, uint64_t n, float w1, float beta2, float w2, float eps)
{
    constexpr uint64_t TILE = 256ull * 4 * ITEMS;
    const uint64_t tiles = n / TILE;
    for (uint64_t tile = blockIdx.x; tile < tiles; tile += PERSIST ? gridDim.x : tiles) {
        const uint64_t start = tile * TILE;
        v4f p[ITEMS], g[ITEMS], m[ITEMS], v[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const uint64_t i = start + (uint64_t)(it * 256 + threadIdx.x) * 4;
            p[it] = Mem<NT_LOAD, 0>::ld(P + i);
            g[it] = Mem<NT_GRAD, 0>::ld(G + i);
            m[it] = Mem<NT_LOAD, 0>::ld(M + i);
            v[it] = Mem<NT_LOAD, 0>::ld(V + i);
        }
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float pp = p[it][c], mm = m[it][c], vv = v[it][c];
                adam_one(pp, g[it][c], mm, vv, w1, beta2, w2, 1e-3f, 0.9f, eps);
                p[it][c] = pp; m[it][c] = mm; v[it][c] = vv;
            }
        }
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const uint64_t i = start + (uint64_t)(it * 256 + threadIdx.x) * 4;
            Mem<0, NT_STORE>::st(P + i, p[it]);
            Mem<0, NT_STORE>::st(M + i, m[it]);
            Mem<0, NT_STORE>::st(V + i, v[it]);
        }
        if (!PERSIST) break;
    }
}

__global__ void __launch_bounds__(256) copy_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, uint64_t n4)
{
    const uint64_t base = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    v4f x[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = src[base + k * 256];
#pragma unroll
    for (int k = 0; k < 4; k++) dst[base + k * 256] = x[k];
}

struct Variant { const char* name; void (*launch)(float*, const float*, float*, float*, uint64_t, hipStream_t); };

template <int ITEMS, int NL, int NS, int NG, int PERSIST>
static void launch(float* P, const float* G, float* M, float* V, uint64_t n, hipStream_t s)
{
    const uint64_t tiles = n / (256ull * 4 * ITEMS);
    const uint32_t grid = PERSIST ? (uint32_t)std::min<uint64_t>(tiles, 256ull * PERSIST) : (uint32_t)tiles;
    hipLaunchKernelGGL((adam_variant<ITEMS, NL, NS, NG, PERSIST>), dim3(grid), dim3(256), 0, s, P, G, M, V, n, 0.1f, 0.999f, 0.001f, 1e-15f);
}

int main(int argc, char** argv)
{
    uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 6000000ull * 59;
    n = n / 8192 * 8192;
    float *P, *G, *M, *V;
    CHECK(hipMalloc(&P, n * 4)); CHECK(hipMalloc(&G, n * 4)); CHECK(hipMalloc(&M, n * 4)); CHECK(hipMalloc(&V, n * 4));
    CHECK(hipMemset(P, 0, n * 4)); CHECK(hipMemset(G, 0, n * 4)); CHECK(hipMemset(M, 0, n * 4)); CHECK(hipMemset(V, 0, n * 4));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    std::vector<Variant> vs = {
        {"V0 shipped: 4 x float4 per tensor and thread", launch<4, 0, 0, 0, 0>},
        {"V1 nontemporal stores", launch<4, 0, 1, 0, 0>},
        {"V2 nontemporal loads + stores", launch<4, 1, 1, 1, 0>},
        {"V3 nontemporal gradient load only", launch<4, 0, 0, 1, 0>},
        {"V4 2 x float4", launch<2, 0, 0, 0, 0>},
        {"V5 8 x float4", launch<8, 0, 0, 0, 0>},
        {"V6 2 x float4, nontemporal loads + stores", launch<2, 1, 1, 1, 0>},
        {"V7 persistent, 8 workgroups per CU, 4 x float4", launch<4, 0, 0, 0, 8>},
        {"V8 persistent, 4 workgroups per CU, 4 x float4, nt", launch<4, 1, 1, 1, 4>},
        {"V9 1 x float4", launch<1, 0, 0, 0, 0>},
    };
    const int rounds = 9, reps = 5;
    std::vector<std::vector<float>> ms(vs.size() + 1);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int r = 0; r < rounds + 1; r++) {
        for (size_t k = 0; k <= vs.size(); k++) {
            CHECK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; i++) {
                if (k < vs.size()) vs[k].launch(P, G, M, V, n, s);
                else {                                  // a float4 copy P -> G: 4 B read + 4 B written per element
                    const uint64_t n4 = n / 4 / 1024 * 1024;
                    hipLaunchKernelGGL(copy_kernel, dim3((uint32_t)(n4 / 1024)), dim3(256), 0, s, (const v4f*)P, (v4f*)G, n4);
                }
            }
            CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
            float t; CHECK(hipEventElapsedTime(&t, e0, e1));
            if (r) ms[k].push_back(t / reps);
        }
    }
    printf("elements %llu (%.2f GB per step)\n", (unsigned long long)n, n * 28.0 / 1e9);
    for (size_t k = 0; k <= vs.size(); k++) {
        std::sort(ms[k].begin(), ms[k].end());
        const float mn = ms[k][0], md = ms[k][ms[k].size() / 2];
        const double bytes = k < vs.size() ? n * 28.0 : n * 8.0;
        printf("%-56s min %.4f ms (%.0f GB/s)  median %.4f ms (%.0f GB/s)\n", k < vs.size() ? vs[k].name : "copy: float4, one tensor to another (8 B per element)", mn, bytes / mn / 1e6, md, bytes / md / 1e6);
    }
    return 0;
}
