// standalone microbench (not product code): issue rate of v_dot2c_f32_f16 / _bf16 against v_pk_fma_f32 / v_fma_f32 at 1 .. 4 waves per SIMD,
// with 8 or 2 independent accumulation chains -- the question behind fd_kernels_dw5p.h: what does a 15-dot2 output cost?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/dot2_rate.hip -o scratch/dot2_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP, int CH>
__global__ void __launch_bounds__(64) k(float *out, const unsigned *in, int iters)
{
    unsigned a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * (8 + i)]; }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f2 pacc[8]; for (int i = 0; i < 8; ++i) pacc[i] = f2{0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int q = CH == 8 ? c : (c & 1);
                if (OP == 0) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc[q]) : "v"(a[c]), "v"(b[(c + r) & 7]));
                if (OP == 1) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[q]) : "v"(a[c]), "v"(b[(c + r) & 7]));
                if (OP == 2) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[q]) : "v"(a[c]), "v"(b[(c + r) & 7]));
                if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pacc[q]) : "v"(pacc[(c + 1) & 7]), "v"(pacc[(c + 2) & 7]));
                if (OP == 4) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a[c]), "v"(b[(c + r) & 7]));
                if (OP == 5) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[q]) : "v"(a[c]), "v"(b[(c + r) & 7]));   // f16 (high half) x f32 + f32
                if (OP == 6) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[q]) : "v"(a[c]), "v"(b[(c + r) & 7]));   // f16 lo x f16 hi + f32
                if (OP == 7) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[q]) : "v"(a[(c + 1) & 7]), "v"(b[(c + r) & 7]));
                if (OP == 8) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(a[q]) : "v"(a[(c + 1) & 7]), "v"(b[(c + r) & 7]));
                if (OP == 9) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a[q]) : "v"(b[(c + r) & 7]));
                if (OP == 10) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[q]) : "v"(b[(c + r) & 7]));
                if (OP == 11) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a[q]) : "v"(a[(c + 1) & 7]), "v"(b[(c + r) & 7]));
                if (OP == 12) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[q]) : "v"(a[(c + 1) & 7]), "v"(b[(c + r) & 7]));
                if (OP == 13) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pacc[q]) : "v"(pacc[(c + 1) & 7]), "v"(pacc[(c + 2) & 7]));
                if (OP == 14) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[q]) : "v"(a[(c + 1) & 7]), "v"(b[(c + r) & 7]));
            }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i] + pacc[i].x + pacc[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int OP, int CH> void run(const char *name, float *out, const unsigned *in)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int wps : {3}) {
        const int blocks = 1024 * wps;
        hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(64), 0, 0, out, in, 10);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(64), 0, 0, out, in, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double inst_per_simd = (double)wps * iters * 128;
        printf("%-18s chains %d  waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.1 GHz)\n", name, CH, wps, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.1);
    }
}
int main(int argc, char **)
{
    float *out; unsigned *in; CK(hipMalloc(&out, 1024 * 8 * 64 * 4)); CK(hipMalloc(&in, 64 * 16 * 4)); CK(hipMemset(in, 0, 64 * 16 * 4));
    if (argc > 1) {   // random fp16 operands in (-2, 2): the chip clocks to its power budget, and operand toggling is power
        unsigned h[64 * 16]; srand(5);
        for (auto &v : h) { unsigned short a = (unsigned short)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15)), b = (unsigned short)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15)); v = a | ((unsigned)b << 16); }
        CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
        printf("random operands\n");
    }
    run<0, 8>("v_dot2c_f32_f16", out, in); run<2, 8>("v_fmac_f32", out, in); run<3, 8>("v_pk_fma_f32", out, in);
    run<5, 8>("v_fma_mix f16*f32", out, in); run<6, 8>("v_fma_mix f16*f16", out, in); run<7, 8>("v_perm_b32", out, in); run<8, 8>("v_pk_add_f16", out, in);
    run<9, 8>("v_lshlrev_b32", out, in); run<10, 8>("v_cvt_f32_f16", out, in); run<11, 8>("v_cvt_pk_bf16_f32", out, in); run<12, 8>("v_max_f32", out, in);
    run<13, 8>("v_pk_mul_f32", out, in); run<14, 8>("v_cndmask_b32", out, in);
    return 0;
}
