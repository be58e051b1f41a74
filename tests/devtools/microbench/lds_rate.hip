// Developer micro-benchmark: cost of ds_read_b128 / ds_read_b32 / ds_add_f32 on gfx950 by address pattern.
// Prints cycles per wave-instruction per CU (2.4 GHz assumed), 4 waves/SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, int pattern)
{
    __shared__ f4 s[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1024; i += 256) s[i] = f4{ (float)i, 1, 2, 3 };
    __syncthreads();
    int idx;
    switch (pattern) {
    case 0: idx = 5; break;                                   // wave-uniform
    case 1: idx = (lane >> 4) * 37 + 3; break;                // one address per 16-lane row
    case 2: idx = lane; break;                                // consecutive 16-byte slots
    case 3: idx = (lane * 197 + 11) & 1023; break;            // scattered
    default: idx = (lane >> 2) * 5; break;                    // one address per quad
    }
    f4 acc = { 0, 0, 0, 0 };
    float a1 = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int j = (idx + r * 64 + it) & 1023;
            if (KIND == 0) { f4 v = s[j]; acc += v; }
            if (KIND == 1) { a1 += ((float*)s)[j * 4]; }
            if (KIND == 2) { atomicAdd(&((float*)s)[j * 4], 1.0f); }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w + a1 == 12345.678f) out[0] = acc.x;
}

template <int KIND> void row(const char* name, float* out)
{
    printf("%-14s", name);
    const char* pn[] = { "uniform", "per-row", "consecutive", "scattered", "per-quad" };
    for (int p = 0; p < 5; p++) {
        const int iters = 2000, blocks = 256 * 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, 10, p);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_cu = (double)iters * 8 * 16;      // 16 waves per CU
        printf(" | %s %.1f", pn[p], ms * 1e-3 * 2.4e9 / insts_per_cu);
    }
    printf("\n");
}

int main()
{
    float* out; hipMalloc(&out, 64);
    printf("cycles per wave-instruction per CU (includes the loop's address arithmetic, ~2-3 VALU)\n");
    row<0>("ds_read_b128", out); row<1>("ds_read_b32", out); row<2>("ds_add_f32", out);
    return 0;
}
