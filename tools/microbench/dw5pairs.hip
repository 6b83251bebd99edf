// standalone microbench (not product code): the round-6 pixel-pair 5x5 kernel fd_dw5_rows and its two measured alternatives (dw5_variants.h)
// against fd_dwconv<T, 5, 1, 2, ACT, 8> and a CPU reference on the three up2 + skip units of the decoder (B = 32) and the pruned widths (B = 64):
// same inputs, outputs compared, all timed; sweep of the band height.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/dw5pairs.hip -o scratch/dw5pairs
#include "../../fast-depth_amd/csrc/fd_kernels_f32.h"
#include "dw5_variants.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; __builtin_memcpy(&h, &u, 2); return (float)h; }
static unsigned short f2b(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float b2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }

template <typename T> struct cvt;
template <> struct cvt<fd_half> { static unsigned short to(float f) { return f2h(f); } static float from(unsigned short u) { return h2f(u); } static const char *name() { return "f16"; } };
template <> struct cvt<fd_bf16> { static unsigned short to(float f) { return f2b(f); } static float from(unsigned short u) { return b2f(u); } static const char *name() { return "bf16"; } };

template <typename T>
static void run(int B, int H, int C, const std::vector<int> &bhs)
{
    const int W = H, Hs = H / 2, Ws = W / 2;
    const size_t n_hi = (size_t)B * H * W * C, n_lo = (size_t)B * Hs * Ws * C;
    T *low, *skip, *o_old, *o_new; float *wf, *bias; unsigned *wpk;
    CK(hipMalloc(&low, n_lo * 2)); CK(hipMalloc(&skip, n_hi * 2)); CK(hipMalloc(&o_old, n_hi * 2)); CK(hipMalloc(&o_new, n_hi * 2));
    CK(hipMalloc(&wf, 25 * C * 4)); CK(hipMalloc(&bias, C * 4)); CK(hipMalloc(&wpk, 30 * C * 4));
    std::vector<unsigned short> hl(n_lo), hs(n_hi);
    srand(1);
    for (auto &v : hl) v = cvt<T>::to((rand() % 2001 - 600) * 1e-3f);
    for (auto &v : hs) v = cvt<T>::to((rand() % 2001 - 600) * 2e-3f);
    std::vector<float> hw(25 * C), hb(C);
    for (auto &v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
    for (auto &v : hb) v = (rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(low, hl.data(), n_lo * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(skip, hs.data(), n_hi * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(wf, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((fd_pack_dw5_pairs<T>), dim3((30 * C + 255) / 256), dim3(256), 0, 0, wf, wpk, C);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto &&launch, int reps) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < reps; ++i) launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3;
    };
    // old kernel, as the plan configures it (fd_plan_build.h): 8 channels per work-item, 64-channel blocks, 7 x 16 tiles
    const int cb = 64, cbq = 3, th = std::min(H, 7), tw = 16, tiles_x = (W + tw - 1) / tw, tiles_y = (H + th - 1) / th;
    const int th_in = th + 4, tw_in = tw + 4;
    for (int pstr : {72, 80}) {
        const size_t lds = ((size_t)th_in * tw_in * pstr * 2 + 15) / 16 * 16 + ((size_t)25 * cb + cb) * 4;
        dim3 grid(tiles_x * tiles_y, (C + cb - 1) / cb, B);
        const double us = time([&]() { hipLaunchKernelGGL((fd_dwconv<T, 5, 1, 2, 1, 8>), grid, dim3(256), lds, 0, low, skip, wf, bias, o_old, H, W, H, W, C, cbq, th, tw, tiles_x, 0, pstr); }, 20);
        printf("%s B=%d %dx%d C=%d  fd_dwconv<N=8> pitch %d: %.1f us\n", cvt<T>::name(), B, H, W, C, pstr, us);
    }
    std::vector<unsigned short> ho(n_hi), hn(n_hi);
    CK(hipMemcpy(ho.data(), o_old, n_hi * 2, hipMemcpyDeviceToHost));
    const double bytes = (double)(2 * n_hi + n_lo) * 2;
    // CPU reference (double) on two images and a few channels
    auto ref_err = [&](const std::vector<unsigned short> &o) {
        double md = 0;
        for (int n : {0, B - 1}) for (int c : {0, 1, 5, 6, C / 2 + 3, C - 1}) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
            double acc = hb[c];
            for (int ky = 0; ky < 5; ++ky) for (int kx = 0; kx < 5; ++kx) {
                const int iy = y + ky - 2, ix = x + kx - 2;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float v = cvt<T>::from(cvt<T>::to(cvt<T>::from(hl[(((size_t)n * Hs + iy / 2) * Ws + ix / 2) * C + c]) + cvt<T>::from(hs[(((size_t)n * H + iy) * W + ix) * C + c])));
                acc += (double)hw[(ky * 5 + kx) * C + c] * v;
            }
            if (acc < 0) acc = 0;
            md = std::max(md, std::fabs(acc - cvt<T>::from(o[(((size_t)n * H + y) * W + x) * C + c])));
        }
        return md;
    };
    printf("   old kernel vs CPU reference: max|d| %.3g\n", ref_err(ho));
    for (int bh : bhs) {
        const int tiles = (W + 27) / 28, two = ((W + tiles - 1) / tiles + 3) / 4 * 4, bands = (H + bh - 1) / bh;
        const int cblocks = (C + 63) / 64, cbs = ((C + cblocks - 1) / cblocks + 7) / 8 * 8;
        dim3 grid(tiles * bands, cblocks, B);
        CK(hipMemset(o_new, 0xff, n_hi * 2));
        const double us = time([&]() { hipLaunchKernelGGL((fd_dw5_pairs<T, 1>), grid, dim3(256), FD_DW5P_LDS, 0, low, skip, wpk, bias, o_new, H, W, C, cbs, two, tiles, bh); }, 20);
        CK(hipMemcpy(hn.data(), o_new, n_hi * 2, hipMemcpyDeviceToHost));
        double maxd = 0, maxv = 0; size_t bad = 0;
        for (size_t i = 0; i < n_hi; ++i) {
            const double a = cvt<T>::from(ho[i]), b = cvt<T>::from(hn[i]);
            if (!(std::fabs(b) < 1e30)) { ++bad; continue; }
            maxd = std::max(maxd, std::fabs(a - b)); maxv = std::max(maxv, std::fabs(a));
        }
        printf("%s B=%d %dx%d C=%d  fd_dw5_pairs bh=%d (%d WGs, two=%d cbs=%d): %.1f us = %.2f TB/s | vs old: max|d| %.3g of max %.3g, non-finite %zu | vs CPU %.3g\n", cvt<T>::name(), B, H, W, C, bh,
               (int)(grid.x * grid.y * grid.z), two, cbs, us, bytes / us * 1e-6, maxd, maxv, bad, ref_err(hn));
    }
    for (int bh : bhs) for (int cl : {32, 64}) {
        const int spw = 64 / cl, groups = ((W + 3) / 4 + spw - 1) / spw, bands = (H + bh - 1) / bh;
        const int cblocks = (C + 2 * cl - 1) / (2 * cl), cbs = ((C + cblocks - 1) / cblocks + 7) / 8 * 8;
        dim3 grid((groups * bands + FD_DW5R_BLOCK / 64 - 1) / (FD_DW5R_BLOCK / 64), cblocks, B);
        CK(hipMemset(o_new, 0xff, n_hi * 2));
        const double us = cl == 32 ? time([&]() { hipLaunchKernelGGL((fd_dw5_rows<T, 1, 32>), grid, dim3(FD_DW5R_BLOCK), 0, 0, low, skip, wpk, bias, o_new, H, W, C, cbs, groups, bh); }, 20)
                                   : time([&]() { hipLaunchKernelGGL((fd_dw5_rows<T, 1, 64>), grid, dim3(FD_DW5R_BLOCK), 0, 0, low, skip, wpk, bias, o_new, H, W, C, cbs, groups, bh); }, 20);
        CK(hipMemcpy(hn.data(), o_new, n_hi * 2, hipMemcpyDeviceToHost));
        double maxd = 0; size_t bad = 0;
        for (size_t i = 0; i < n_hi; ++i) {
            const double a = cvt<T>::from(ho[i]), b = cvt<T>::from(hn[i]);
            if (!(std::fabs(b) < 1e30)) { ++bad; continue; }
            maxd = std::max(maxd, std::fabs(a - b));
        }
        printf("%s B=%d %dx%d C=%d  fd_dw5_rows<CL=%d> bh=%d (%d WGs, cbs=%d): %.1f us = %.2f TB/s | vs old: max|d| %.3g, non-finite %zu | vs CPU %.3g\n", cvt<T>::name(), B, H, W, C, cl, bh,
               (int)(grid.x * grid.y * grid.z), cbs, us, bytes / us * 1e-6, maxd, bad, ref_err(hn));
    }
    for (int bh : bhs) {
        const int groups = (W + 7) / 8, bands = (H + bh - 1) / bh;
        const int cblocks = (C + 63) / 64, cbs = ((C + cblocks - 1) / cblocks + 7) / 8 * 8;
        dim3 grid((groups * bands + 3) / 4, cblocks, B);
        CK(hipMemset(o_new, 0xff, n_hi * 2));
        const double us = time([&]() { hipLaunchKernelGGL((fd_dw5_dma<T, 1>), grid, dim3(256), 0, 0, low, skip, wpk, bias, o_new, H, W, C, cbs, groups, bh); }, 20);
        CK(hipMemcpy(hn.data(), o_new, n_hi * 2, hipMemcpyDeviceToHost));
        double maxd = 0; size_t bad = 0;
        for (size_t i = 0; i < n_hi; ++i) {
            const double a = cvt<T>::from(ho[i]), b = cvt<T>::from(hn[i]);
            if (!(std::fabs(b) < 1e30)) { ++bad; continue; }
            maxd = std::max(maxd, std::fabs(a - b));
        }
        printf("%s B=%d %dx%d C=%d  fd_dw5_dma bh=%d (%d WGs, cbs=%d): %.1f us = %.2f TB/s | vs old: max|d| %.3g, non-finite %zu | vs CPU %.3g\n", cvt<T>::name(), B, H, W, C, bh,
               (int)(grid.x * grid.y * grid.z), cbs, us, bytes / us * 1e-6, maxd, bad, ref_err(hn));
    }
    CK(hipFree(low)); CK(hipFree(skip)); CK(hipFree(o_old)); CK(hipFree(o_new)); CK(hipFree(wf)); CK(hipFree(bias)); CK(hipFree(wpk));
}

int main(int argc, char **)
{
    { int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)fd_dw5_rows<fd_half, 1, 32>, FD_DW5R_BLOCK, 0)); printf("occupancy API: fd_dw5_rows<f16, 32> blocks of %d per CU: %d\n", FD_DW5R_BLOCK, nb);
      CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)fd_dw5_pairs<fd_half, 1>, 256, FD_DW5P_LDS)); printf("occupancy API: fd_dw5_pairs<f16> blocks of 256 per CU: %d\n", nb); }
    if (argc > 1) { run<fd_half>(32, 112, 64, {20}); run<fd_half>(32, 56, 128, {14}); run<fd_half>(32, 28, 256, {8}); run<fd_bf16>(32, 112, 64, {20}); return 0; }   // quick mode (ablation builds)
    run<fd_half>(32, 112, 64, {112, 56, 28, 20, 14, 10});
    run<fd_half>(32, 56, 128, {56, 28, 20, 14, 10, 8});
    run<fd_half>(32, 28, 256, {28, 14, 10, 8, 6, 4});
    run<fd_bf16>(32, 112, 64, {28, 20, 14});
    run<fd_bf16>(32, 56, 128, {20, 14, 10});
    run<fd_bf16>(32, 28, 256, {14, 10, 6});
    run<fd_half>(64, 112, 56, {28, 20, 14});     // pruned widths (configs[4], B = 64)
    run<fd_half>(64, 56, 120, {20, 14, 10});
    run<fd_half>(64, 28, 200, {14, 10, 6});
    return 0;
}
