// microbench: how fast can the GPU launch short-lived workgroups?  (not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
extern __shared__ float smem[];
template <int NREG>
__global__ void __launch_bounds__(256) busy(float *out, int iters, int use_lds) {
  float acc[NREG];
  for (int i = 0; i < NREG; ++i) acc[i] = threadIdx.x * 0.001f + i;
  if (use_lds) smem[threadIdx.x] = acc[0];
  __syncthreads();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NREG; ++i) acc[i] = fmaf(acc[i], 1.0001f, 0.5f);
  float s = 0; for (int i = 0; i < NREG; ++i) s += acc[i];
  if (use_lds) s += smem[(threadIdx.x + 1) & 255];
  if (s == 12345.678f) out[0] = s;
}
template <int NREG> void run(float *out, dim3 grid, size_t lds, int iters, const char *tag) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void *)busy<NREG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 > 65536 ? 65536 : 65536));
  hipLaunchKernelGGL(busy<NREG>, grid, dim3(256), lds, 0, out, iters, lds > 0);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(busy<NREG>, grid, dim3(256), lds, 0, out, iters, lds > 0);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  const long wgs = (long)grid.x * grid.y * grid.z;
  printf("%-34s grid %5u x %u x %2u = %6ld WGs, lds %6zu, iters %4d: %8.1f us  -> %.1f WG/us, per-WG work ~%.2f us\n", tag, grid.x, grid.y, grid.z, wgs, lds, iters, ms * 1e3, wgs / (ms * 1e3),
         iters * NREG * 4.0 / 2100.0);
}
int main() {
  float *out; CK(hipMalloc(&out, 1024));
  for (int iters : {8, 64, 256}) {
    run<16>(out, dim3(6272, 1, 1), 0, iters, "16 regs, no LDS, 1-D");
    run<16>(out, dim3(98, 2, 32), 0, iters, "16 regs, no LDS, 3-D");
    run<16>(out, dim3(6272, 1, 1), 37888, iters, "16 regs, 37 KB LDS");
    run<64>(out, dim3(6272, 1, 1), 37888, iters / 4 > 0 ? iters / 4 : 1, "64 regs, 37 KB LDS");
    run<16>(out, dim3(1024, 1, 1), 37888, iters * 6, "16 regs, 37 KB LDS, 1024 WGs x6 work");
    run<16>(out, dim3(25088, 1, 1), 0, iters, "16 regs, no LDS, 25088 WGs");
  }
  return 0;
}
