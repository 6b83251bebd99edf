// microbench (round 5): what does it cost a producer kernel to ADD its per-tile BatchNorm partial sums into shared rows with 64-bit
// integer no-return atomics, instead of storing one partial row per tile for a finalisation launch?  (not product code)
//   mode 0  stream work only (each workgroup reads `work` KB and stores one value)
//   mode 1  + NCOL device-scope (agent) no-return u64 atomics per workgroup into ONE row
//   mode 2  + NCOL workgroup-scope (= executed in the XCD's own L2) atomics into the row of the XCD the workgroup runs on (HW_REG_XCC_ID)
//   mode 3  mode 1 with every slot padded to its own 128-byte line
//   mode 4  mode 2, then s_waitcnt vmcnt(0) + barrier + one RETURNING device-scope ticket (16 logical shards): the price of a last-arriver tail
// Prints time per launch and checks the sums on the host (mode 2/4: the eight rows are summed on the host AFTER the kernel boundary, which is
// what a consumer kernel would do).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }   // HW_REG_XCC_ID[3:0]

template <int MODE>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ in, float *__restrict__ out, u64 *rows, u64 *tickets, int ncol, int work_vec4_per_thread, int pad) {
  float4 acc = {0, 0, 0, 0};
  const float4 *p = in + (size_t)blockIdx.x * 256 * work_vec4_per_thread + threadIdx.x;
  for (int i = 0; i < work_vec4_per_thread; ++i) { float4 v = p[(size_t)i * 256]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 12345.678f) out[blockIdx.x] = s;
  const int t = threadIdx.x;
  if (MODE == 1 || MODE == 3) {
    if (t < ncol) __hip_atomic_fetch_add(rows + (size_t)t * pad, (u64)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (MODE == 2 || MODE == 4) {
    const int x = xcc_id();
    if (t < ncol) __hip_atomic_fetch_add(rows + (size_t)x * ncol + t, (u64)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        u64 r = __hip_atomic_fetch_add(tickets + (blockIdx.x & 15) * 16, (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r == 0xffffffffffffull) out[0] = 1.f;
      }
    }
  }
}

template <int MODE> double run(const float4 *in, float *out, u64 *rows, u64 *tickets, int G, int ncol, int work_kb, int reps, bool *ok) {
  const int pad = MODE == 3 ? 16 : 1;
  const int wv = work_kb * 1024 / 16 / 256;
  const size_t row_bytes = (size_t)8 * ncol * 8 * 16;
  CK(hipMemset(rows, 0, row_bytes)); CK(hipMemset(tickets, 0, 16 * 16 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(256), 0, 0, in, out, rows, tickets, ncol, wv, pad);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(256), 0, 0, in, out, rows, tickets, ncol, wv, pad);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<u64> h(row_bytes / 8); CK(hipMemcpy(h.data(), rows, row_bytes, hipMemcpyDeviceToHost));
  *ok = true;
  if (MODE != 0) for (int c = 0; c < ncol; ++c) {
    u64 s = 0;
    if (MODE == 1 || MODE == 3) s = h[(size_t)c * pad]; else for (int x = 0; x < 8; ++x) s += h[(size_t)x * ncol + c];
    if (s != (u64)(c + 1) * G * (reps + 1)) { *ok = false; }
  }
  return ms * 1e3 / reps;
}

int main() {
  float4 *in; float *out; u64 *rows, *tickets;
  const size_t in_bytes = (size_t)1 << 30;
  CK(hipMalloc(&in, in_bytes)); CK(hipMemset(in, 0, in_bytes)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&rows, 8 * 2048 * 8 * 16)); CK(hipMalloc(&tickets, 4096));
  printf("%-7s %-5s %-8s | %9s %9s %9s %9s %9s   (us per launch; adds per address = G)\n", "G", "ncol", "work KB", "none", "dev 1row", "xcd 8row", "dev pad", "xcd+tail");
  for (int ncol : {64, 128, 256, 1024}) for (int G : {256, 784, 1568, 3136, 6272, 25088}) for (int work : {0, 16, 64}) {
    if ((size_t)G * work * 1024 > in_bytes) continue;
    bool ok[5]; double t[5];
    t[0] = run<0>(in, out, rows, tickets, G, ncol, work, 20, &ok[0]);
    t[1] = run<1>(in, out, rows, tickets, G, ncol, work, 20, &ok[1]);
    t[2] = run<2>(in, out, rows, tickets, G, ncol, work, 20, &ok[2]);
    t[3] = run<3>(in, out, rows, tickets, G, ncol, work, 20, &ok[3]);
    t[4] = run<4>(in, out, rows, tickets, G, ncol, work, 20, &ok[4]);
    printf("%-7d %-5d %-8d | %9.1f %9.1f %9.1f %9.1f %9.1f   sums %s%s%s%s\n", G, ncol, work, t[0], t[1], t[2], t[3], t[4], ok[1] ? "ok " : "BAD ", ok[2] ? "ok " : "BAD ", ok[3] ? "ok " : "BAD ", ok[4] ? "ok" : "BAD");
  }
  return 0;
}
