// probe of ds_read_b64_tr_b16 on gfx950: which LDS elements does each lane receive?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned short *out, int mode)
{
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = lane * 8;                                   // every lane its own consecutive 8 bytes
    else if (mode == 1) addr = (lane & 15) / 4 * 64 + (lane & 3) * 8 + (lane >> 4) * 1024;   // 16-lane group: 4 rows (pitch 64 B = 32 elems) x 4 chunks
    else addr = (lane >> 2) * 128 + (lane & 3) * 8;                   // row pitch 128 B: lane/4 = row, lane%4 = chunk
    unsigned base = (unsigned)(size_t)lds;                            // LDS byte address of the array
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main()
{
    unsigned short *d; hipMalloc(&d, 64 * 4 * 2);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        std::vector<unsigned short> h(256);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : " | ");
    }
    return 0;
}
