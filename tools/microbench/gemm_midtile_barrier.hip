// experiment: software-pipelined variant of fd_pw_gemm_f32 -- the per-K-tile barrier sits in the MIDDLE of the tile's MFMAs and
// the first fragments of the next tile are fetched behind the current tile's last MFMAs (not product code)
#include "../../fast-depth_amd/csrc/fd_kernels_f32.h"
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int WGM, int WGN, int TM, int TN, int ACT, int STAGES>
__global__ void __launch_bounds__(64 * WGM * WGN)
gemm_v4(const float *__restrict__ A, const float *__restrict__ Wt, const float *__restrict__ bias,
        float *__restrict__ out, int M, int N, int K, int K32, int m_tiles, int n_tiles)
{
    constexpr int NW = WGM * WGN;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, BK = 32;
    constexpr int ROWS = BM + BN, STAGE = ROWS * BK, RG = ROWS / 8 / NW;
    constexpr int LEAD = STAGES - 2;                       // tiles issued ahead of the one whose barrier is being passed
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % n_tiles, mt = (slot / n_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;
    const float *src[RG];
    int src_chunk[RG];
    bool src_is_a[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        const int r = (wave + NW * i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        src_chunk[i] = c * 4;
        src_is_a[i] = r < BM;
        if (r < BM) { long row = m0 + r; if (row > M - 1) row = M - 1; src[i] = A + row * K; }
        else { int row = n0 + (r - BM); if (row > N - 1) row = N - 1; src[i] = Wt + (long)row * K32; }
    }
    auto issue = [&](int t) {
        float *dst = smem + (t % STAGES) * STAGE + wave * 8 * BK;
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            int k = t * BK + src_chunk[i];
            if (src_is_a[i] && k >= K) k = 0;
            fd_glds16(src[i] + k, dst + i * NW * 8 * BK);
        }
    };
    fd_f32x16 acc[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        const float bv = col < N ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }
    const int h = lane >> 5;
    int a_off[TM][4], b_off[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) a_off[i][g] = row * BK + (((2 * g + h) ^ ((row >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = BM + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) b_off[j][g] = row * BK + (((2 * g + h) ^ ((row >> 1) & 7)) << 2);
    }
    const int T = K32 / BK;
    // prologue: tiles 0 .. LEAD in flight, tile 0 landed, its first fragments in registers
#pragma unroll
    for (int t = 0; t <= LEAD; ++t) if (t < T) issue(t);
    {
        // wait for tile 0 only: newer groups outstanding = min(LEAD, T-1)
        const int newer = (T - 1 < LEAD) ? T - 1 : LEAD;
        if (newer >= 2) fd_wait_vmcnt<2 * RG>(); else if (newer == 1) fd_wait_vmcnt<RG>(); else fd_wait_vmcnt<0>();
    }
    fd_block_barrier();
    fd_f32x4 a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = fd_ld4(smem + a_off[i][0]);
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = fd_ld4(smem + b_off[j][0]);
    for (int t = 0; t < T; ++t) {
        const float *cur = smem + (t % STAGES) * STAGE;
        const float *nxt = smem + ((t + 1) % STAGES) * STAGE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g == 2) {
                // mid-tile: every wave has left tile t-1 (stage (t+LEAD+1) % STAGES is free) -> refill it; tile t+1 must have
                // landed for everybody before its first fragments are read behind the last MFMAs of this tile
                if (t + 1 < T) {
                    const int newer = (T - 1 - (t + 1) < LEAD - 0) ? T - 1 - (t + 1) : LEAD - 0;   // groups issued after tile t+1 and still allowed in flight
                    // outstanding newer than tile t+1: tiles t+2 .. t+LEAD (issued in earlier iterations)
                    const int nn = newer < (LEAD - 1 > 0 ? LEAD - 1 : 0) ? newer : (LEAD - 1 > 0 ? LEAD - 1 : 0);
                    if (nn >= 1) fd_wait_vmcnt<RG>(); else fd_wait_vmcnt<0>();
                    fd_block_barrier();
                    if (t + 1 + LEAD < T) issue(t + 1 + LEAD);
                }
            }
            if (g < 3) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(g + 1) & 1][i] = fd_ld4(cur + a_off[i][g + 1]);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(g + 1) & 1][j] = fd_ld4(cur + b_off[j][g + 1]);
            } else if (t + 1 < T) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[0][i] = fd_ld4(nxt + a_off[i][0]);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[0][j] = fd_ld4(nxt + b_off[j][0]);
            }
            FD_SCHED_FENCE();
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][q], b[g & 1][j][q], acc[i][j], 0, 0, 0);
            FD_SCHED_FENCE();
        }
    }
    const bool full = m0 + BM <= M;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        if (col >= N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long rbase = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
            float *o = out + rbase * N + col;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * N] = fd_act<ACT>(acc[i][j][r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (rbase + (r & 3) + 8 * (r >> 2) < M) o[((r & 3) + 8 * (r >> 2)) * N] = fd_act<ACT>(acc[i][j][r]);
            }
        }
    }
}

template <int WGM,int WGN,int TM,int TN,int STAGES, bool V4>
float run(const float* A, const float* W, const float* bias, float* out, int M, int N, int K, int iters) {
  constexpr int BM=WGM*TM*32, BN=WGN*TN*32;
  int mt=(M+BM-1)/BM, nt=(N+BN-1)/BN; size_t lds = (size_t)STAGES*(BM+BN)*32*4;
  auto kern = V4 ? (void(*)(const float*,const float*,const float*,float*,int,int,int,int,int,int))gemm_v4<WGM,WGN,TM,TN,2,STAGES>
                 : (void(*)(const float*,const float*,const float*,float*,int,int,int,int,int,int))fd_pw_gemm_f32<WGM,WGN,TM,TN,2>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((mt+7)/8*8*nt);
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i=0;i<3;++i) hipLaunchKernelGGL(kern, grid, dim3(64*WGM*WGN), lds, 0, A,W,bias,out,M,N,K,(K+31)/32*32,mt,nt);
  CK(hipEventRecord(e0,0));
  for (int i=0;i<iters;++i) hipLaunchKernelGGL(kern, grid, dim3(64*WGM*WGN), lds, 0, A,W,bias,out,M,N,K,(K+31)/32*32,mt,nt);
  CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/iters*1e3f;
}

int main() {
  const size_t maxA = (size_t)401408*128, maxW = 4096*1024, maxO = (size_t)401408*128;
  float *A,*W,*bias,*out,*out2; CK(hipMalloc(&A,maxA*4)); CK(hipMalloc(&W,maxW*4)); CK(hipMalloc(&bias,4096*4)); CK(hipMalloc(&out,maxO*4)); CK(hipMalloc(&out2,maxO*4));
  std::vector<float> h(maxA); for (auto& v: h) v = (rand()%2001-1000)*1e-3f; CK(hipMemcpy(A,h.data(),maxA*4,hipMemcpyHostToDevice));
  CK(hipMemcpy(W,h.data()+12345,maxW*4,hipMemcpyHostToDevice)); CK(hipMemcpy(bias,h.data()+777,4096*4,hipMemcpyHostToDevice));
  struct S{int M,N,K;}; S shapes[] = {{6272,512,512},{6272,512,256},{1568,1024,1024},{25088,256,256},{100352,128,128},{100352,128,64},{401408,64,32},{1568,512,1024},{6250,500,200}};
  printf("%-18s %11s %11s %11s %11s %11s %11s\n","shape","base 64x64","v4s3 64x64","v4s4 64x64","base 64x128","v4s3 64x128","v4s4 64x128");
  for (auto s: shapes) {
    double fl = 2.0*s.M*s.N*s.K;
    // correctness: v4 vs base, bitwise
    run<2,2,1,1,3,false>(A,W,bias,out,s.M,s.N,s.K,1); run<2,2,1,1,3,true>(A,W,bias,out2,s.M,s.N,s.K,1);
    std::vector<float> r1((size_t)s.M*s.N), r2((size_t)s.M*s.N); CK(hipMemcpy(r1.data(),out,r1.size()*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(),out2,r2.size()*4,hipMemcpyDeviceToHost));
    size_t bad=0; for (size_t i=0;i<r1.size();++i) if (r1[i]!=r2[i]) ++bad;
    run<2,2,1,1,4,true>(A,W,bias,out2,s.M,s.N,s.K,1); CK(hipMemcpy(r2.data(),out2,r2.size()*4,hipMemcpyDeviceToHost));
    size_t bad4=0; for (size_t i=0;i<r1.size();++i) if (r1[i]!=r2[i]) ++bad4;
    float t[6] = { run<2,2,1,1,3,false>(A,W,bias,out,s.M,s.N,s.K,20), run<2,2,1,1,3,true>(A,W,bias,out,s.M,s.N,s.K,20), run<2,2,1,1,4,true>(A,W,bias,out,s.M,s.N,s.K,20),
                   run<2,2,1,2,3,false>(A,W,bias,out,s.M,s.N,s.K,20), run<2,2,1,2,3,true>(A,W,bias,out,s.M,s.N,s.K,20), run<2,2,1,2,4,true>(A,W,bias,out,s.M,s.N,s.K,20) };
    char nm[64]; snprintf(nm,64,"%dx%dx%d",s.M,s.N,s.K); printf("%-18s", nm);
    for (int i=0;i<6;++i) printf(" %5.1f(%3.0f)", t[i], fl/t[i]/1e6); printf("  mismatches %zu %zu\n", bad, bad4);
  }
  return 0;
}
