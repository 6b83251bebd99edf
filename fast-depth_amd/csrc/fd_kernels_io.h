// fd_kernels_io.h -- input preparation on the device: the reference's NYU validation transform
// (dataloaders/nyu.py:48-59: Resize(250/480) -> CenterCrop(228, 304) -> Resize(output_size), all nearest-neighbour, then /255)
// is a pure index map, so the host composes the three steps into one row table and one column table
// (fast-depth_amd/dataloaders/nyu.py) and this kernel gathers: raw HWC uint8 frames + raw depth -> the network's NCHW fp32 input
// (+ the transformed depth target).  At 35 k frames/s the CPU transform (PIL per frame) would be the bottleneck.
#pragma once
#include "fd_device.h"

static __global__ void __launch_bounds__(256)
fd_val_transform_u8(const unsigned char *__restrict__ rgb, const float *__restrict__ depth, const int *__restrict__ ymap,
                    const int *__restrict__ xmap, float *__restrict__ x, float *__restrict__ d, int n, int H, int W, int oh, int ow)
{
    const long total = (long)n * oh * ow;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % ow);
        const long t = i / ow;
        const int oy = (int)(t % oh), f = (int)(t / oh);
        const long src = ((long)f * H + ymap[oy]) * W + xmap[ox];
        const unsigned char *p = rgb + src * 3;
        // the reference divides in float64 (np.asfarray(rgb) / 255) and converts to float32 afterwards (dataloader.py:97-99)
        float *o = x + ((long)f * 3 * oh + oy) * ow + ox;
        o[0] = (float)((double)p[0] / 255.0);
        o[(long)oh * ow] = (float)((double)p[1] / 255.0);
        o[2L * oh * ow] = (float)((double)p[2] / 255.0);
        if (d) d[i] = depth[src];
    }
}

// ---- gradient exchange in 16 bits (optional: SURVEY.md 8(e), the 7.92 MB form of the data-parallel all-reduce): a bucket of the flat fp32
// gradient vector -> bfloat16 (round to nearest even) before the collective, and back after it.  4 elements per work-item.
static __global__ void __launch_bounds__(256)
fd_cast_f32_bf16(const float *__restrict__ src, fd_bf16 *__restrict__ dst, long n4, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) fd_st4(dst + 4 * i, fd_ld4(src + 4 * i));
    for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) fd_st1(dst + i, src[i]);
}
static __global__ void __launch_bounds__(256)
fd_cast_bf16_f32(const fd_bf16 *__restrict__ src, float *__restrict__ dst, long n4, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) fd_st4(dst + 4 * i, fd_ld4(src + 4 * i));
    for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = fd_ld1(src + i);
}

