// standalone microbench: fd_pw_gemm16_f32 (16x16x4 MFMA, k-split wave pairs, one workgroup per CU) against the 32x32x2 kernel
// on the pointwise shapes of the network (not product code).  usage: gemm16 [shape index]
#define FD_GEMM16_PROBE 1
#include "../../fast-depth_amd/csrc/fd_kernels_f32.h"
#include "../../fast-depth_amd/csrc/fd_kernels_gemm16_f32.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static hipEvent_t e0, e1;

template <int WGM,int WGN,int TM,int TN>
float run_old(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int iters) {
  constexpr int BM=WGM*TM*32, BN=WGN*TN*32;
  int mt=(M+BM-1)/BM, nt=(N+BN-1)/BN; size_t lds = 3*(BM+BN)*32*4;
  if (lds > 65536) CK(hipFuncSetAttribute((const void*)fd_pw_gemm_f32<WGM,WGN,TM,TN,2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((mt+7)/8*8*nt);
  for (int i=0;i<2;++i) hipLaunchKernelGGL((fd_pw_gemm_f32<WGM,WGN,TM,TN,2>), grid, dim3(64*WGM*WGN), lds, 0, A,W,bias,out,M,N,K,(K+31)/32*32,mt,nt);
  CK(hipEventRecord(e0,0));
  for (int i=0;i<iters;++i) hipLaunchKernelGGL((fd_pw_gemm_f32<WGM,WGN,TM,TN,2>), grid, dim3(64*WGM*WGN), lds, 0, A,W,bias,out,M,N,K,(K+31)/32*32,mt,nt);
  CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/iters*1e3f;
}

template <int TM, int S, int ABL = 0>
float run_new(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int stride, int iters) {
  int mt=(M+stride-1)/stride, nt=(N+63)/64; size_t lds = (size_t)S*(TM*16+64)*32*4;
  CK(hipFuncSetAttribute((const void*)fd_pw_gemm16_f32<TM,S,2,ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((mt+7)/8*8*nt);
  for (int i=0;i<2;++i) hipLaunchKernelGGL((fd_pw_gemm16_f32<TM,S,2,ABL>), grid, dim3(512), lds, 0, A,W,bias,out,M,N,K,(K+31)/32*32,stride,mt,nt, fd_dwfuse{}, fd_g16_train{});
  CK(hipEventRecord(e0,0));
  for (int i=0;i<iters;++i) hipLaunchKernelGGL((fd_pw_gemm16_f32<TM,S,2,ABL>), grid, dim3(512), lds, 0, A,W,bias,out,M,N,K,(K+31)/32*32,stride,mt,nt, fd_dwfuse{}, fd_g16_train{});
  CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/iters*1e3f;
}

static double maxdiff(const float* a, const float* b, size_t n, std::vector<float>& ha, std::vector<float>& hb) {
  ha.resize(n); hb.resize(n);
  CK(hipMemcpy(ha.data(), a, n*4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n*4, hipMemcpyDeviceToHost));
  double d = 0, m = 0; for (size_t i=0;i<n;++i) { d = std::max(d, (double)fabsf(ha[i]-hb[i])); m = std::max(m, (double)fabsf(ha[i])); }
  return d / (m > 0 ? m : 1);
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t maxA = (size_t)401408*128, maxW = 4096*1024, maxO = (size_t)401408*128;
  float *A,*W,*bias,*out,*out2; CK(hipMalloc(&A,maxA*4)); CK(hipMalloc(&W,maxW*4)); CK(hipMalloc(&bias,8192)); CK(hipMalloc(&out,maxO*4)); CK(hipMalloc(&out2,maxO*4));
  std::vector<float> h(maxA); for (auto& v: h) v = (rand()%2001-1000)*1e-3f; CK(hipMemcpy(A,h.data(),maxA*4,hipMemcpyHostToDevice));
  for (auto& v: h) v = (rand()%2001-1000)*2e-4f;
  CK(hipMemcpy(W,h.data(),maxW*4,hipMemcpyHostToDevice)); CK(hipMemcpy(bias,h.data(),8192,hipMemcpyHostToDevice));
  struct S{int M,N,K;}; S shapes[] = {{6272,512,512},{6272,512,256},{1568,1024,1024},{1568,1024,512},{1568,512,1024},{6272,256,512},{25088,256,256},{25088,256,128},{25088,128,256},
                                      {100352,128,128},{100352,128,64},{100352,64,128},{401408,64,32},{401408,32,64},
                                      {12544,408,256},{12544,376,408},{50176,120,256},{200704,56,120},{3136,200,512}};
  std::vector<float> ha, hb;
  if (only == -3) {   // effective shader clock inside the kernel: clock64() ticks per 100 MHz wall tick, main loop only (before the stores)
    for (int abl : {0, 1, 3}) {
      float t = abl == 0 ? run_new<13,4,0>(A,W,bias,out2,6272,512,2048,196,10) : abl == 1 ? run_new<13,4,1>(A,W,bias,out2,6272,512,2048,196,10) : run_new<13,4,3>(A,W,bias,out2,6272,512,2048,196,10);
      std::vector<long long> pr(4*256); CK(hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(fd_gemm16_probe), pr.size()*8));
      double cyc = 0, wall = 0; for (int b = 0; b < 256; ++b) { cyc += (double)(pr[4*b+1]-pr[4*b]); wall += (double)(pr[4*b+3]-pr[4*b+2]); }
      printf("ABL=%d K=2048: %.1f us; per workgroup main loop %.0f shader cycles in %.0f ticks of the constant 100 MHz counter -> %.3f GHz\n", abl, t, cyc/256, wall/256, cyc/wall*0.1);
    }
    return 0;
  }
  if (only == -2 || only == -1) {   // K sweep and ablations on the 6272 x 512 shape (TM 13, stride 196, 4 stages)
    for (int K : {128, 256, 512, 1024, 2048}) {
      const double fl = 2.0*6272*512*K;
      float t0 = run_new<13,4,0>(A,W,bias,out2,6272,512,K,196,20), t1 = run_new<13,4,1>(A,W,bias,out2,6272,512,K,196,20), t2 = run_new<13,4,2>(A,W,bias,out2,6272,512,K,196,20), t3 = run_new<13,4,3>(A,W,bias,out2,6272,512,K,196,20), t4 = run_new<13,4,4>(A,W,bias,out2,6272,512,K,196,20);
      float o = run_old<2,2,1,2>(A,W,bias,out,6272,512,K,20);
      printf("K=%4d  full %.1f us (%.0f TF) | no-dma %.1f (%.0f) | no-dma,no-frags %.1f (%.0f) | mfma only %.1f (%.0f) | full, no stores %.1f | old 64x128 %.1f (%.0f)\n", K, t0, fl/t0/1e6, t1, fl/t1/1e6, t2, fl/t2/1e6, t3, fl/t3/1e6, t4, o, fl/o/1e6);
    }
    fflush(stdout);
    if (only == -2) return 0;
  }
  int si = -1;
  for (auto s: shapes) { ++si; if (only >= 0 && si != only) continue;
    const double fl = 2.0*s.M*s.N*s.K;
    const int nt = (s.N+63)/64;
    float t_old[3] = { run_old<2,2,1,1>(A,W,bias,out,s.M,s.N,s.K,20), run_old<2,2,2,1>(A,W,bias,out,s.M,s.N,s.K,20), run_old<2,2,1,2>(A,W,bias,out,s.M,s.N,s.K,20) };
    run_old<2,2,1,1>(A,W,bias,out,s.M,s.N,s.K,1);
    printf("%dx%dx%d  old 64x64 %.1f us (%.0f TF)  128x64 %.1f (%.0f)  64x128 %.1f (%.0f)\n", s.M,s.N,s.K, t_old[0], fl/t_old[0]/1e6, t_old[1], fl/t_old[1]/1e6, t_old[2], fl/t_old[2]/1e6);
    // candidate strides: full tile, and the strides that make the grid r * 256 workgroups
    const int tms[3] = {13, 7, 4};
    for (int tm : tms) {
      std::vector<int> strides; strides.push_back(tm*16);
      for (int r = 1; r <= 16; ++r) { int mtiles = (256*r + nt - 1) / nt; int st = (s.M + mtiles - 1) / mtiles; if (st <= tm*16 && st * 100 >= tm*16 * (tm == 4 ? 75 : 86)) strides.push_back(st); }
      std::sort(strides.begin(), strides.end()); strides.erase(std::unique(strides.begin(), strides.end()), strides.end());
      for (int st : strides) {
        float t3 = 0, t4 = 0;
        if (tm == 13) { t3 = run_new<13,3>(A,W,bias,out2,s.M,s.N,s.K,st,20); t4 = run_new<13,4>(A,W,bias,out2,s.M,s.N,s.K,st,20); }
        if (tm == 7) { t3 = run_new<7,3>(A,W,bias,out2,s.M,s.N,s.K,st,20); t4 = run_new<7,4>(A,W,bias,out2,s.M,s.N,s.K,st,20); }
        if (tm == 4) { t3 = run_new<4,3>(A,W,bias,out2,s.M,s.N,s.K,st,20); t4 = run_new<4,4>(A,W,bias,out2,s.M,s.N,s.K,st,20); }
        const int mtl = (s.M + st - 1) / st;
        printf("   TM=%2d stride=%3d wgs=%5d (%.2f/CU)  S3 %.1f us (%.0f TF)  S4 %.1f us (%.0f TF)  maxrel %.2e\n", tm, st, mtl*nt, mtl*nt/256.0, t3, fl/t3/1e6, t4, fl/t4/1e6,
               maxdiff(out, out2, (size_t)s.M*s.N, ha, hb));
      }
    }
    fflush(stdout);
  }
  return 0;
}
