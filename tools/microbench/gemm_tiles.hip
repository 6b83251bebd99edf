// standalone microbench for the pointwise GEMM variants (not product code)
#include "../../fast-depth_amd/csrc/fd_kernels_f32.h"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// ceiling: same grid / waves, MFMAs only (operands: per-lane pseudo-random values, not zeros)
template <int TILES, int CHAIN>
__global__ void __launch_bounds__(256) mfma_only(float* out, int iters) {
  fd_f32x16 acc[TILES]; for (int i=0;i<TILES;++i) for (int r=0;r<16;++r) acc[i][r]=0.f;
  float a[CHAIN], b[CHAIN];
  for (int c=0;c<CHAIN;++c) { a[c] = __sinf(threadIdx.x*0.37f + c)*0.9f; b[c] = __cosf(blockIdx.x*0.11f + threadIdx.x*0.23f + c)*0.9f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAIN; ++c)
#pragma unroll
      for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], b[c], acc[t], 0, 0, 0);
  }
  float s = 0; for (int t=0;t<TILES;++t) for (int r=0;r<16;++r) s += acc[t][r];
  if (s == 12345.678f) out[0] = s;
}
template <int TILES> void ceil_run(float* out) {
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int grids[] = {256, 768, 784, 1024, 2048};
  for (int g : grids) {
    const int iters = 4096 / (16 * TILES) * 4;   // 16384 MFMAs per wave
    hipLaunchKernelGGL((mfma_only<TILES,16>), dim3(g), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e0,0)); for (int i=0;i<5;++i) hipLaunchKernelGGL((mfma_only<TILES,16>), dim3(g), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=5; double fl = (double)g*4*iters*16*TILES*2.0*32*32*2;
    printf("mfma_only tiles=%d grid=%4d: %8.1f us  %6.1f TF/s\n", TILES, g, ms*1e3, fl/ms/1e9);
  }
}

template <int WGM,int WGN,int TM,int TN>
float run(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int iters) {
  constexpr int BM=WGM*TM*32, BN=WGN*TN*32;
  int mt=(M+BM-1)/BM, nt=(N+BN-1)/BN; size_t lds = 3*(BM+BN)*32*4;
  if (lds > 65536) CK(hipFuncSetAttribute((const void*)fd_pw_gemm_f32<WGM,WGN,TM,TN,2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((mt+7)/8*8*nt);
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i=0;i<3;++i) hipLaunchKernelGGL((fd_pw_gemm_f32<WGM,WGN,TM,TN,2>), grid, dim3(64*WGM*WGN), lds, 0, A,W,bias,out,M,N,K,(K+31)/32*32,mt,nt);
  CK(hipEventRecord(e0,0));
  for (int i=0;i<iters;++i) hipLaunchKernelGGL((fd_pw_gemm_f32<WGM,WGN,TM,TN,2>), grid, dim3(64*WGM*WGN), lds, 0, A,W,bias,out,M,N,K,(K+31)/32*32,mt,nt);
  CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/iters*1e3f;
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  const size_t maxA = (size_t)401408*128, maxW = 4096*1024, maxO = (size_t)401408*128;
  float *A,*W,*bias,*out; CK(hipMalloc(&A,maxA*4)); CK(hipMalloc(&W,maxW*4)); CK(hipMalloc(&bias,4096)); CK(hipMalloc(&out,maxO*4));
  std::vector<float> h(maxA); for (auto& v: h) v = (rand()%2001-1000)*1e-3f; CK(hipMemcpy(A,h.data(),maxA*4,hipMemcpyHostToDevice));
  CK(hipMemcpy(W,h.data(),maxW*4,hipMemcpyHostToDevice)); CK(hipMemset(bias,0,4096));
  struct S{int M,N,K;}; S shapes[] = {{401408,64,32},{100352,128,64},{100352,128,128},{25088,256,128},{25088,256,256},{6272,512,256},{6272,512,512},{1568,1024,512},{1568,1024,1024},{1568,512,1024},{6272,256,512},{25088,128,256},{100352,64,128},{401408,32,64}};
  printf("%-18s %11s %11s %11s %11s %11s %11s %11s\n","shape","64x64/4w","64x128/4w","128x64/8w","64x128/8w","128x128/8w","128x128/8w'","256x64/8w");
  int si = -1;
  for (auto s: shapes) { ++si; if (only >= 0 && si != only) continue;
    double fl = 2.0*s.M*s.N*s.K;
    float t[7] = { run<2,2,1,1>(A,W,bias,out,s.M,s.N,s.K,20), run<2,2,1,2>(A,W,bias,out,s.M,s.N,s.K,20), run<4,2,1,1>(A,W,bias,out,s.M,s.N,s.K,20),
                   run<2,4,1,1>(A,W,bias,out,s.M,s.N,s.K,20), run<4,2,1,2>(A,W,bias,out,s.M,s.N,s.K,20), run<2,4,2,1>(A,W,bias,out,s.M,s.N,s.K,20), run<4,2,2,1>(A,W,bias,out,s.M,s.N,s.K,20) };
    char nm[64]; snprintf(nm,64,"%dx%dx%d",s.M,s.N,s.K); printf("%-18s", nm);
    for (int i=0;i<7;++i) printf(" %5.1f(%3.0f)", t[i], fl/t[i]/1e6); printf("\n");
  }
  return 0;
}
