// Developer micro-benchmark: issue rate of VALU instruction classes on gfx950 as a function of waves per SIMD and of the number of
// independent dependency chains per wave (ILP).  Prints cycles per wave-instruction per SIMD (GPU clock assumed 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND, int ILP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b)
{
    float x[ILP]; f2 p[ILP]; double d[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) { x[i] = threadIdx.x * 1e-3f + i; p[i] = f2{ x[i], x[i] + 1 }; d[i] = x[i]; }
    const f2 A = { a, a }, B = { b, b };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(A), "v"(B));
                if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)a), "v"((double)b));
                if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                if (KIND == 5) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
                if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : );
                if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(A));
                if (KIND == 9) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
                if (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 11) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x[i]) : "v"(a));

                if (KIND == 12) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x[i]) : "v"(a));
                if (KIND == 13) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 14) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 15) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(A));
                if (KIND == 16) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[i]) : "v"(x[i]));
                if (KIND == 17) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(x[i]) : "v"(d[i]));
                if (KIND == 18) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)a));
                if (KIND == 19) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)a));
                if (KIND == 20) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[i]) : "v"((double)a) : "vcc");
                if (KIND == 21) asm volatile("v_div_fmas_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"((double)a));
                if (KIND == 22) asm volatile("v_div_fixup_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"((double)a));
                if (KIND == 23) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(it));
                if (KIND == 24) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[i]));
                if (KIND == 25) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
                if (KIND == 26) asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" : : "v"(x[i]), "v"(a) : "s10", "s11");
                if (KIND == 27) asm volatile("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
                if (KIND == 28) asm volatile("v_or_b32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
                if (KIND == 29) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 30) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 31) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
                if (KIND == 32) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3c088908" : "+v"(x[i]) : "v"(a));
                if (KIND == 33) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x[i]));
                if (KIND == 34) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 35) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 36) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
                if (KIND == 37) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x[i]) : "v"(a));
                if (KIND == 38) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(A), "v"(B));
                if (KIND == 39) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += x[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND, int ILP>
double run(int waves_per_simd, float* out)
{
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;       // 256 CUs x (waves_per_simd x 4 waves = blocks of 256 threads)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, ILP>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 8 * ILP * waves_per_simd;      // wave-instructions each SIMD executes
    return ms * 1e-3 * 2.4e9 / insts_per_simd;
}

template <int KIND> void row(const char* name, float* out)
{
    printf("%-14s", name);
    for (int w : { 1, 4, 8 }) {
        printf(" | w%d:", w);
        printf(" %.2f", run<KIND, 1>(w, out));
        printf(" %.2f", run<KIND, 2>(w, out));
        printf(" %.2f", run<KIND, 4>(w, out));
    }
    printf("\n");
}

int main()
{
    float* out; hipMalloc(&out, 64);
    printf("cycles per wave-instruction per SIMD (2.4 GHz assumed); columns: waves/SIMD x ILP {1,2,4}\n");
    row<0>("v_fma_f32", out); row<7>("v_mul_f32", out); row<1>("v_pk_fma_f32", out); row<8>("v_pk_mul_f32", out); row<2>("v_fma_f64", out);
    row<3>("v_exp_f32", out); row<4>("v_rcp_f32", out); row<9>("v_rcp_f64", out); row<5>("v_add_f32_dpp", out); row<6>("v_cndmask_b32", out);
    row<10>("v_mov_b32", out); row<11>("v_alignbit_b32", out);
    row<12>("cndmask e64 sgpr", out); row<39>("cndmask e64 vcc", out); row<13>("v_bfi_b32", out); row<14>("v_add_f32", out); row<15>("v_pk_add_f32", out);
    row<16>("cvt_f64_f32", out); row<17>("cvt_f32_f64", out); row<18>("v_mul_f64", out); row<19>("v_add_f64", out); row<20>("div_scale_f64", out);
    row<21>("div_fmas_f64", out); row<22>("div_fixup_f64", out); row<23>("v_ldexp_f32", out); row<24>("v_rndne_f32", out); row<25>("v_cmp e32 vcc", out);
    row<26>("v_cmp e64 sgpr", out); row<27>("mov_dpp ror8", out); row<28>("or_dpp mirror", out); row<29>("v_lshl_or_b32", out); row<30>("v_fmac_f32", out);
    row<31>("v_rsq_f32", out); row<32>("v_fmaak_f32", out); row<33>("v_ffbl_b32", out); row<34>("v_max_f32", out); row<35>("v_and_b32", out);
    row<36>("add_dpp ror4", out); row<37>("fma inline 1.0", out); row<38>("pk_fma op_sel", out);
    return 0;
}
