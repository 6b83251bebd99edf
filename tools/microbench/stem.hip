// standalone microbench: fd_stem3x3s2 and its ablations (not product code)
#include "../../fast-depth_amd/csrc/fd_kernels_f32.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <int CH> float run(const float* x, const float* wp, const float* bias, float* y, int B, int H, int W, int iters) {
  const int Ho=H/2, Wo=W/2; const int nrows = 2*((255+Wo-1)/Wo)+3; size_t lds = std::max((size_t)3*nrows*(W+8)*4, (size_t)4*64*36*4);
  dim3 grid((Ho*Wo+255)/256, B);
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i=0;i<3;++i) hipLaunchKernelGGL((fd_stem3x3s2<float,2,CH>), grid, dim3(256), lds, 0, x,wp,bias,y,B,H,W,32);
  CK(hipEventRecord(e0,0)); for (int i=0;i<iters;++i) hipLaunchKernelGGL((fd_stem3x3s2<float,2,CH>), grid, dim3(256), lds, 0, x,wp,bias,y,B,H,W,32);
  CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/iters*1e3f;
}
__global__ void copy_k(const fd_f32x4* a, fd_f32x4* b, long n) { long i = (long)blockIdx.x*256+threadIdx.x; if (i<n) b[i]=a[i]; }
int main() {
  const int B=32,H=224,W=224; float *x,*wp,*bias,*y; CK(hipMalloc(&x,(size_t)B*3*H*W*4)); CK(hipMalloc(&wp,27*32*4)); CK(hipMalloc(&bias,128)); CK(hipMalloc(&y,(size_t)B*112*112*32*4));
  std::vector<float> h((size_t)B*3*H*W); for (auto&v:h) v=(rand()%1000)*1e-3f; CK(hipMemcpy(x,h.data(),h.size()*4,hipMemcpyHostToDevice)); CK(hipMemcpy(wp,h.data(),27*32*4,hipMemcpyHostToDevice)); CK(hipMemset(bias,0,128));
  printf("full %.1f us | no stores %.1f | no mfma %.1f | no input loads %.1f\n", run<32>(x,wp,bias,y,B,H,W,50), run<101>(x,wp,bias,y,B,H,W,50), run<102>(x,wp,bias,y,B,H,W,50), run<103>(x,wp,bias,y,B,H,W,50));
  // reference: a plain 16-byte copy of 51 MB (the output size) and of 19 MB
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (long n : {(long)B*112*112*32/4, (long)B*3*H*W/4}) {
    float *s,*d; CK(hipMalloc(&s,n*16)); CK(hipMalloc(&d,n*16));
    for (int i=0;i<3;++i) hipLaunchKernelGGL(copy_k, dim3((n+255)/256), dim3(256), 0, 0, (const fd_f32x4*)s,(fd_f32x4*)d,n);
    CK(hipEventRecord(e0,0)); for (int i=0;i<50;++i) hipLaunchKernelGGL(copy_k, dim3((n+255)/256), dim3(256), 0, 0, (const fd_f32x4*)s,(fd_f32x4*)d,n); CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("copy of %.1f MB: %.1f us (%.2f TB/s read+write)\n", n*16/1e6, ms/50*1e3, 2*n*16/(ms/50*1e-3)/1e12);
  }
  return 0;
}
