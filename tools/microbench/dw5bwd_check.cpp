// standalone check (not product code) of fd_dw5_bwd_rows against a CPU reference in double: the SAME source builds for the GPU
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/microbench/dw5bwd_check.cpp -o scratch/dw5bwd_check
// and for the CPU emulator
//   clang++ -std=c++17 -O1 -DFD_EMU -I tests/hipemu -I fast-depth_amd/csrc tools/microbench/dw5bwd_check.cpp -o scratch/dw5bwd_check_emu
// (round 6: the kernel was exact on the emulator and 4-28 % off on hardware; this is how the difference was located)
#include "../../fast-depth_amd/csrc/fd_kernels_dw5p_bwd.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static unsigned short f2b(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float b2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; }
#ifdef FD_EMU
template <typename X> X *dev(const std::vector<X> &h) { X *p = (X *)malloc(h.size() * sizeof(X)); std::memcpy(p, h.data(), h.size() * sizeof(X)); return p; }
template <typename X> void back(std::vector<X> &h, const X *p) { std::memcpy(h.data(), p, h.size() * sizeof(X)); }
#else
template <typename X> X *dev(const std::vector<X> &h) { X *p; (void)hipMalloc(&p, h.size() * sizeof(X)); (void)hipMemcpy(p, h.data(), h.size() * sizeof(X), hipMemcpyHostToDevice); return p; }
template <typename X> void back(std::vector<X> &h, const X *p) { (void)hipMemcpy(h.data(), p, h.size() * sizeof(X), hipMemcpyDeviceToHost); }
#endif
int main(int argc, char **argv)
{
    const int B = 2, H = argc > 1 ? atoi(argv[1]) : 16, W = argc > 2 ? atoi(argv[2]) : 16, C = argc > 3 ? atoi(argv[3]) : 16, Hs = H / 2, Ws = W / 2;
    const int delta = argc > 4 ? atoi(argv[4]) : -1;          // >= 0: the taps are a delta at this position
    std::vector<unsigned short> G((size_t)B * H * W * C), Z(G.size()), Zs(G.size()), Zl((size_t)B * Hs * Ws * C), SG(G.size(), 0xffff), GL(Zl.size(), 0xffff);
    std::vector<float> coef(4 * C), w(25 * C), stl(4 * C), sts(4 * C), wpart((size_t)64 * 25 * C * 4, 0.f);
    srand(7);
    auto rnd = [] { return (rand() % 2001 - 1000) * 1e-3f; };
    for (auto &v : G) v = f2b(rnd() * 1e-3f);
    for (auto &v : Z) v = f2b(rnd());
    for (auto &v : Zs) v = f2b(rnd() * 3.f);
    for (auto &v : Zl) v = f2b(rnd());
    for (int c = 0; c < C; ++c) {
        coef[0 * C + c] = 1.f + 0.5f * rnd(); coef[1 * C + c] = 1e-4f * rnd(); coef[2 * C + c] = 0.1f * rnd(); coef[3 * C + c] = 1e-4f * rnd();
        stl[0 * C + c] = 1.f + 0.3f * rnd(); stl[1 * C + c] = 0.2f * rnd(); stl[2 * C + c] = 0.1f * rnd(); stl[3 * C + c] = 1.f + 0.2f * rnd();
        sts[0 * C + c] = 1.f + 0.3f * rnd(); sts[1 * C + c] = 0.5f * rnd(); sts[2 * C + c] = 0.f; sts[3 * C + c] = 1.f;
    }
    for (auto &v : w) v = rnd() * 0.2f;
    if (delta >= 0) for (int c = 0; c < C; ++c) for (int t = 0; t < 25; ++t) w[c * 25 + t] = t == delta ? 1.f : 0.f;
    long long *rows = dev(std::vector<long long>(16 * 3 * 2 * ((C + 15) / 16 * 16), 0));
    fd_dw5_bwd_args<fd_bf16> a{};
    a.G = (const fd_bf16 *)dev(G); a.Z = (const fd_bf16 *)dev(Z); a.Zin = (const fd_bf16 *)dev(Zl); a.Zskip = (const fd_bf16 *)dev(Zs);
    a.Gin = (fd_bf16 *)dev(GL); a.SGout = (fd_bf16 *)dev(SG);
    a.coef = dev(coef); a.w = dev(w); a.st_in = dev(stl); a.st_skip = dev(sts); a.wpart = dev(wpart);
    a.sr = fd_stat_rows{rows, 1, (C + 15) / 16 * 16};
    a.H = H; a.W = W; a.C = C; a.groups_x = (W + 7) / 8;
    const int bands = std::max(1, (H + 7) / 14); a.bh_d = a.bh_w = ((H + bands - 1) / bands + 1) / 2 * 2;
    a.wgs_d = a.wgs_w = (a.groups_x * ((H + a.bh_d - 1) / a.bh_d) + 3) / 4;
    hipLaunchKernelGGL((fd_dw5_bwd_rows<fd_bf16, 1, 2>), dim3(a.wgs_d + a.wgs_w, (C + 63) / 64, B), dim3(256), 0, 0, a);
#ifndef FD_EMU
    (void)hipDeviceSynchronize();
#endif
    back(SG, (const unsigned short *)a.SGout); back(GL, (const unsigned short *)a.Gin);
    // reference: dz rounded to bf16, taps rounded to bf16, d_in in double
    std::vector<double> dz(G.size());
    for (size_t i = 0; i < G.size(); ++i) { const int c = (int)(i % C); dz[i] = b2f(f2b(coef[c] * ((b2f(G[i]) - coef[C + c]) - (b2f(Z[i]) - coef[2 * C + c]) * coef[3 * C + c]))); }
    double md = 0, mx = 0; int shown = 0;
    for (int n = 0; n < B; ++n) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int c = 0; c < C; ++c) {
        double s = 0;
        for (int ky = 0; ky < 5; ++ky) for (int kx = 0; kx < 5; ++kx) {
            const int oy = y + 2 - ky, ox = x + 2 - kx;
            if (oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
            s += dz[(((size_t)n * H + oy) * W + ox) * C + c] * b2f(f2b(w[c * 25 + ky * 5 + kx]));
        }
        const double got = b2f(SG[(((size_t)n * H + y) * W + x) * C + c]);
        const double d = std::fabs(got - s);
        if (d > 0.02 * 1e-3 && shown++ < 12) printf("n%d y%d x%d c%d ref %.6g got %.6g\n", n, y, x, c, s, got);
        md = std::max(md, d); mx = std::max(mx, std::fabs(s));
    }
    printf("skip gradient: max|d| %.4g of max %.4g (%d bad)\n", md, mx, shown);
    return 0;
}
