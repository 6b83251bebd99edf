// standalone microbench: fd_dwpw_f32 on the large-map units of the B=32 224x224 plan, with ablations (not product code)
#define FD_DWPW_PROBE
#include "../../fast-depth_amd/csrc/fd_kernels_dwpw_f32.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static float *g_in, *g_skip, *g_out, *g_w;
template <int KS, int S, int MODE, int WM, int NT, int NLD, int ABL>
float run(int B, int Hin, int Win, int C, int N, int TH, int TW, int iters = 30)
{
    const int Ho = Hin / S, Wo = Win / S;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const size_t lds = ((size_t)NLD * 32 * 36 + 2 * 32 * WM * 32 + (size_t)N * C + (size_t)KS * KS * C + C) * 4;
    int tws = 0; while ((1 << tws) < TW) ++tws;
    if (((TH - 1) * S + KS) * ((TW - 1) * S + KS) > NLD * 32 || TH * TW > 32 * WM) return -1.f;
    auto k = fd_dwpw_f32<KS, S, MODE, 2, WM, NT, NLD, 0, ABL>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(512), lds, 0, g_in, g_skip, g_w, g_w + 8192, g_w + 16384, g_w + 100000, g_out, B, Hin, Win, Ho, Wo, C, C, N, TH, tws, tiles_x, tiles_x * tiles_y, 1, 36, fd_dwpw_head{});
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, grid, dim3(512), lds, 0, g_in, g_skip, g_w, g_w + 8192, g_w + 16384, g_w + 100000, g_out, B, Hin, Win, Ho, Wo, C, C, N, TH, tws, tiles_x, tiles_x * tiles_y, 1, 36, fd_dwpw_head{});
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters * 1e3f;
}
template <int KS, int S, int MODE, int WM, int NT, int NLD>
void all(const char *name, int B, int Hin, int Win, int C, int N, int TH, int TW, double mb)
{
    const float f = run<KS, S, MODE, WM, NT, NLD, 0>(B, Hin, Win, C, N, TH, TW, 1);
    unsigned long long pr[8][6]; CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(fd_dwpw_probe), sizeof pr));
    printf("%-10s probe kclk: producer w0 commit %.1f issue %.1f bar1 %.1f dw %.1f bar2 %.1f | consumer w6 mma0 %.1f bar1 %.1f mma1 %.1f retire %.1f bar2 %.1f", name,
           pr[0][4] * 1e-3, pr[0][0] * 1e-3, pr[0][1] * 1e-3, pr[0][2] * 1e-3, pr[0][3] * 1e-3, pr[6][0] * 1e-3, pr[6][1] * 1e-3, pr[6][4] * 1e-3, pr[6][2] * 1e-3, pr[6][3] * 1e-3);
    printf("\n");
    printf("%-10s tile %2dx%-2d: full %6.1f us (%.2f TB/s) | no loads %6.1f | no dw %6.1f | no mfma %6.1f | no stores %6.1f\n", name, TH, TW, f, mb / f * 1e-6,
           run<KS, S, MODE, WM, NT, NLD, 1>(B, Hin, Win, C, N, TH, TW), run<KS, S, MODE, WM, NT, NLD, 2>(B, Hin, Win, C, N, TH, TW),
           run<KS, S, MODE, WM, NT, NLD, 3>(B, Hin, Win, C, N, TH, TW), run<KS, S, MODE, WM, NT, NLD, 4>(B, Hin, Win, C, N, TH, TW));
}
int main()
{
    const int B = 32;
    const size_t big = (size_t)B * 112 * 112 * 64;
    CK(hipMalloc(&g_in, big * 4)); CK(hipMalloc(&g_skip, big * 4)); CK(hipMalloc(&g_out, big * 4)); CK(hipMalloc(&g_w, 1 << 20));
    std::vector<float> h(big); for (auto &v : h) v = (rand() % 1000) * 1e-3f;
    CK(hipMemcpy(g_in, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g_skip, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g_w, h.data(), 1 << 20, hipMemcpyHostToDevice));
    const double px112 = (double)B * 112 * 112, px56 = (double)B * 56 * 56;
    all<3, 1, 0, 4, 2, 6>("conv1", B, 112, 112, 32, 64, 8, 16, px112 * (32 + 64) * 4);
    all<3, 1, 0, 4, 2, 6>("conv1", B, 112, 112, 32, 64, 16, 8, px112 * (32 + 64) * 4);
    all<3, 2, 0, 2, 2, 10>("conv2", B, 112, 112, 64, 128, 8, 8, px112 * 64 * 4 + px56 * 128 * 4);
    all<3, 1, 0, 4, 4, 6>("conv3", B, 56, 56, 128, 128, 14, 8, px56 * 256 * 4);
    all<3, 1, 0, 4, 4, 6>("conv3", B, 56, 56, 128, 128, 8, 16, px56 * 256 * 4);
    all<5, 1, 2, 4, 2, 8>("decode4", B, 56, 56, 128, 64, 14, 8, px56 * (128 / 4 + 128 + 64) * 4);
    all<5, 1, 2, 4, 2, 8>("decode4", B, 56, 56, 128, 64, 8, 16, px56 * (128 / 4 + 128 + 64) * 4);
    all<5, 1, 2, 4, 1, 8>("decode5", B, 112, 112, 64, 32, 8, 16, px112 * (64 / 4 + 64 + 32) * 4);
    all<5, 1, 2, 4, 1, 8>("decode5", B, 112, 112, 64, 32, 16, 8, px112 * (64 / 4 + 64 + 32) * 4);
    return 0;
}
