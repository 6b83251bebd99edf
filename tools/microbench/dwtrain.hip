// standalone microbench: fd_dwconv_train (bf16, 3x3 stride 1) with per-phase shader-clock probes (not product code)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFD_DW_PROBE tools/microbench/dwtrain.hip -o scratch/dwtrain/dwtrain
#include "../../fast-depth_amd/csrc/fd_kernels_train.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static void run(int B, int H, int C, int th, int tw, int abl = 0, size_t extra_lds = 0) {
  CK(hipMemcpyToSymbol(HIP_SYMBOL(fd_dw_abl), &abl, 4));
  const int cb = 32, K = 3;
  const size_t n = (size_t)B * H * H * C;
  fd_bf16 *zin, *zout; float *st, *w, *part;
  CK(hipMalloc(&zin, n * 2)); CK(hipMalloc(&zout, n * 2)); CK(hipMalloc(&st, 4 * C * 4)); CK(hipMalloc(&w, 9 * C * 4));
  const int tiles_x = (H + tw - 1) / tw, tiles_y = (H + th - 1) / th;
  CK(hipMalloc(&part, (size_t)tiles_x * tiles_y * B * 2 * C * 4));
  std::vector<unsigned short> h(n); for (auto &v : h) v = 0x3f00 + (rand() & 0xff);
  CK(hipMemcpy(zin, h.data(), n * 2, hipMemcpyHostToDevice));
  std::vector<float> hs(4 * C, 1.0f), hw(9 * C, 0.1f); CK(hipMemcpy(st, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const int th_in = th - 1 + K, tw_in = tw - 1 + K;
  const size_t lds = (std::max((size_t)th_in * tw_in * (cb + 4), (size_t)2048) + (size_t)K * K * cb) * 4 + extra_lds;
  CK(hipFuncSetAttribute((const void *)fd_dwconv_train<fd_bf16, 3, 1, 0, 2, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  dim3 grid(tiles_x * tiles_y, (C + cb - 1) / cb, B);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() { hipLaunchKernelGGL((fd_dwconv_train<fd_bf16, 3, 1, 0, 2, 2, 4>), grid, dim3(256), lds, 0, zin, st, (const fd_bf16 *)nullptr, (const float *)nullptr, w, zout, part, H, H, H, H, C, 3, th, tw, tiles_x, 0, 36, fd_bn_fin{}); };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0, 0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const long wgs = (long)grid.x * grid.y * grid.z;
  std::vector<long long> pr(8 * 16384); CK(hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(fd_dw_probe), pr.size() * 8));
  const long np = std::min<long>(wgs, 16384);
  double d[5] = {0, 0, 0, 0, 0}; long long first = pr[0], last = 0;
  for (long b = 0; b < np; ++b) { for (int k = 0; k < 5; ++k) d[k] += (double)(pr[8 * b + k + 1] - pr[8 * b + k]); first = std::min(first, pr[8 * b]); last = std::max(last, pr[8 * b + 5]); }
  printf("abl %d lds %zu | B=%d %dx%d C=%d tile %dx%d: %ld WGs, %.1f us/launch | per-WG shader clocks: issue+land patch %.0f, LDS commit+barrier %.0f, taps+stores %.0f, barrier %.0f, reduce+partial %.0f | sum %.0f clk = %.2f us at 2.1 GHz | first start -> last end %.0f clk\n",
         abl, lds, B, H, H, C, th, tw, wgs, ms / 20 * 1e3, d[0] / np, d[1] / np, d[2] / np, d[3] / np, d[4] / np, (d[0] + d[1] + d[2] + d[3] + d[4]) / np, (d[0] + d[1] + d[2] + d[3] + d[4]) / np / 2100.0, (double)(last - first));
  CK(hipFree(zin)); CK(hipFree(zout)); CK(hipFree(st)); CK(hipFree(w)); CK(hipFree(part));
}
int main(int argc, char **argv) {
  if (argc > 1) {   // tile-shape sweep
    for (int th : {8, 16, 28}) for (int tw : {16, 32}) run(32, 112, 32, th, tw);
    for (int th : {7, 14, 28}) for (int tw : {16, 28}) run(32, 56, 128, th, tw);
    for (int th : {7, 14, 28}) for (int tw : {16, 28}) run(32, 28, 256, th, tw);
    return 0;
  }
  for (int abl : {0, 1, 2, 3}) run(32, 112, 32, 8, 16, abl);          // conv1.0: full / no stores / no loads / neither
  for (size_t extra : {(size_t)0, (size_t)6000, (size_t)14000, (size_t)27000, (size_t)54000}) run(32, 112, 32, 8, 16, 0, extra);   // residency 5, 4, 3, 2, 1 (by LDS)
  for (int abl : {0, 1, 2, 3}) run(32, 14, 512, 14, 16, abl);          // conv7.0 (whole-frame tiles, as the plan selects)
  for (size_t extra : {(size_t)0, (size_t)8000, (size_t)20000, (size_t)45000, (size_t)100000}) run(32, 14, 512, 14, 16, 0, extra);   // residency by LDS
  run(32, 56, 128, 7, 16);    // conv3.0
  return 0;
}
