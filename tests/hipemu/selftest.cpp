// tests/hipemu/selftest.cpp -- the emulator checked against the lane semantics it claims (CDNA3/4 ISA: DPP controls, ds_bpermute,
// v_permlane{16,32}_swap; HIP: __shfl*, __ballot, __syncthreads_and/or).  `selftest diverge` must die with the emulator's
// divergence report.  Built and run by tests/test_hipemu_parity.py.
#include <hip/hip_runtime.h>

#include <vector>

static int g_fail = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s (thread %u)\n", __FILE__, __LINE__, #cond, threadIdx.x); __atomic_fetch_add(&g_fail, 1, __ATOMIC_RELAXED); } } while (0)

__global__ void k_cross_lane(unsigned* out)
{
    const unsigned tid = threadIdx.x, lane = tid & 63;
    // ballot / shfl family
    CHECK(__ballot(lane & 1) == 0xAAAAAAAAAAAAAAAAull);
    CHECK(__shfl((int)lane, 5) == 5);
    CHECK(__shfl((int)lane, 5, 16) == (int)((lane & ~15u) + 5));
    CHECK(__shfl_xor((int)lane, 32) == (int)(lane ^ 32));
    CHECK(__shfl_up((int)lane, 3) == (int)(lane >= 3 ? lane - 3 : lane));
    CHECK(__shfl_down((int)lane, 3, 64) == (int)(lane + 3 < 64 ? lane + 3 : lane));
    CHECK(__shfl_down(1.5f * lane, 1) == (lane + 1 < 64 ? 1.5f * (lane + 1) : 1.5f * lane));
    CHECK(__shfl(0x100000000ull + lane, 7) == 0x100000007ull);
    CHECK(__builtin_amdgcn_readlane((int)lane * 3, 9) == 27);
    CHECK(__builtin_amdgcn_ds_bpermute((int)((lane ^ 1) << 2), (int)lane) == (int)(lane ^ 1));
    // DPP
    CHECK(__builtin_amdgcn_update_dpp(0, (int)lane, 0xB1, 0xf, 0xf, false) == (int)(lane ^ 1));             // quad_perm [1,0,3,2]
    CHECK(__builtin_amdgcn_update_dpp(0, (int)lane, 0x4E, 0xf, 0xf, false) == (int)(lane ^ 2));             // quad_perm [2,3,0,1]
    CHECK(__builtin_amdgcn_update_dpp(0, (int)lane, 0x00, 0xf, 0xf, false) == (int)(lane & ~3u));           // quad_perm [0,0,0,0]
    CHECK(__builtin_amdgcn_update_dpp(0, (int)lane, 0x141, 0xf, 0xf, false) == (int)((lane & ~7u) | (7 - (lane & 7))));   // row_half_mirror
    CHECK(__builtin_amdgcn_update_dpp(0, (int)lane, 0x140, 0xf, 0xf, false) == (int)((lane & ~15u) | (15 - (lane & 15)))); // row_mirror
    CHECK(__builtin_amdgcn_update_dpp(0, (int)lane, 0x124, 0xf, 0xf, false) == (int)((lane & ~15u) | ((lane - 4) & 15)));  // row_ror:4
    CHECK(__builtin_amdgcn_update_dpp(-1, (int)lane, 0x111, 0xf, 0xf, false) == ((lane & 15) >= 1 ? (int)lane - 1 : -1));  // row_shr:1, old kept
    CHECK(__builtin_amdgcn_update_dpp(-1, (int)lane, 0x111, 0xf, 0xf, true) == ((lane & 15) >= 1 ? (int)lane - 1 : 0));    // bound_ctrl: 0
    CHECK(__builtin_amdgcn_update_dpp(-1, (int)lane, 0x142, 0xa, 0xf, false) == (((lane >> 4) & 1) ? (int)((lane & ~15u) - 1) : -1));   // row_bcast:15, rows 1 and 3
    CHECK(__builtin_amdgcn_update_dpp(-1, (int)lane, 0x143, 0xc, 0xf, false) == (lane >= 32 ? 31 : -1));                     // row_bcast:31
    CHECK(__builtin_amdgcn_update_dpp(-1, (int)lane, 0x138, 0xf, 0xf, false) == (lane >= 1 ? (int)lane - 1 : -1));           // wave_shr:1
    // permlane swaps: {vdst, src0} after the swap
    {
        const auto r = __builtin_amdgcn_permlane32_swap(100u + lane, 200u + lane, false, false);
        CHECK(r.x == (lane < 32 ? 100u + lane : 200u + lane - 32));
        CHECK(r.y == (lane < 32 ? 100u + lane + 32 : 200u + lane));
        const auto s = __builtin_amdgcn_permlane16_swap(100u + lane, 200u + lane, false, false);
        CHECK(s.x == (((lane >> 4) & 1) ? 200u + lane - 16 : 100u + lane));
        CHECK(s.y == (((lane >> 4) & 1) ? 200u + lane : 100u + lane + 16));
    }
    CHECK(__builtin_amdgcn_alignbit(0x80000001u, 0x40000000u, 31) == 0x00000002u);
    CHECK(__builtin_amdgcn_sbfe(0xF0, 4, 4) == -1);
    CHECK(__builtin_amdgcn_bitop3_b32(0xF0F0F0F0u, 0xCCCCCCCCu, 0xAAAAAAAAu, 0x96) == (0xF0F0F0F0u ^ 0xCCCCCCCCu ^ 0xAAAAAAAAu));
    // workgroup barrier with votes, LDS
    __shared__ unsigned s_sum;
    if (tid == 0) s_sum = 0;
    __syncthreads();
    atomicAdd(&s_sum, tid);
    CHECK(__syncthreads_and(tid < 256) == 1);
    CHECK(__syncthreads_and(tid != 77) == 0);
    CHECK(__syncthreads_or(tid == 77) == 1);
    CHECK(__syncthreads_or(0) == 0);
    CHECK(s_sum == 255u * 256u / 2u);
    // exec-mask semantics: half of every quad's... no: whole waves' lanes 48-63 leave the loop region early and wait at the barrier
    unsigned seen = 0;
    if (lane < 48) {
        for (int it = 0; it < 3; it++) seen += (unsigned)__popcll(__ballot(1));
    }
    __syncthreads();
    CHECK(seen == (lane < 48 ? 3u * 48u : 0u));
    // a loop with quad-uniform trip counts holding quad-scope DPP (gather_tile_partials' shape)
    int acc = 0;
    for (unsigned k = 0; k < (lane >> 2); k++) acc += __builtin_amdgcn_update_dpp(0, 1, 0xB1, 0xf, 0xf, false);
    CHECK(acc == (int)(lane >> 2));
    // lanes that return early no longer take part
    if (tid >= 192 + 32) return;
    const unsigned long long m = __ballot(1);
    CHECK(m == (tid >= 192 ? 0xFFFFFFFFull : ~0ull));
    if (tid == 0) out[blockIdx.x] = 1;
}

__global__ void k_lookback(unsigned* flags, unsigned* out)
{
    // workgroup b waits for workgroup b - 1 (decoupled look-back's dependency direction)
    if (threadIdx.x == 0) {
        unsigned v = 0;
        if (blockIdx.x > 0) { while ((v = __hip_atomic_load(&flags[blockIdx.x - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(1); }
        __hip_atomic_store(&flags[blockIdx.x], v + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        out[blockIdx.x] = v + 1;
    }
}

__global__ void k_diverge()
{
    // lanes of ONE wave at two different cross-lane sites, none of them at the barrier: the hardware would run the two sides one
    // after the other under exec masks; the emulator must refuse loudly
    int x;
    if (threadIdx.x & 1) x = __shfl((int)threadIdx.x, 0);
    else x = (int)__ballot(1);
    if (x == 12345) std::printf("?");
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::strcmp(argv[1], "diverge") == 0) {
        hipLaunchKernelGGL(k_diverge, dim3(1), dim3(64), 0, nullptr);
        return 0;
    }
    std::vector<unsigned> out(40, 0), flags(64, 0), order(64, 0);
    hipLaunchKernelGGL(k_cross_lane, dim3(40), dim3(256), 0, nullptr, out.data());
    for (unsigned v : out) if (v != 1) { std::printf("a workgroup did not finish\n"); g_fail++; }
    hipLaunchKernelGGL(k_lookback, dim3(64), dim3(64), 0, nullptr, flags.data(), order.data());
    for (unsigned i = 0; i < 64; i++) if (order[i] != i + 1) { std::printf("look-back chain broken at %u\n", i); g_fail++; }
    if (g_fail) { std::printf("%d checks failed\n", g_fail); return 1; }
    std::printf("all ok\n");
    return 0;
}
