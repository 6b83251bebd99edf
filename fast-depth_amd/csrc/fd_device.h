// fd_device.h -- device-side basics shared by every kernel header.
//
// Product builds (hipcc --offload-arch=gfx950) take the HIP branch.  The FD_EMU branch exists only so
// that tests/hipemu can compile the very same kernel and plan sources for the CPU emulator; it is not
// a backend and is never built into libfastdepth_hip.so.
#pragma once
#ifdef FD_EMU
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#define FD_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

typedef float fd_f32x4 __attribute__((ext_vector_type(4)));
typedef float fd_f32x2 __attribute__((ext_vector_type(2)));
typedef float fd_f32x16 __attribute__((ext_vector_type(16)));

#define FD_ACT_NONE_ 0
#define FD_ACT_RELU_ 1
#define FD_ACT_RELU6_ 2

// activation of the fused Conv-BN-act units: ReLU (models.py:67,74) or ReLU6 (imagenet/mobilenet.py:16-20)
template <int ACT>
__device__ __forceinline__ float fd_act(float v)
{
    if (ACT == FD_ACT_RELU_) return fmaxf(v, 0.0f);
    if (ACT == FD_ACT_RELU6_) return fminf(fmaxf(v, 0.0f), 6.0f);
    return v;
}
template <int ACT>
__device__ __forceinline__ fd_f32x4 fd_act4(fd_f32x4 v)
{
    fd_f32x4 r;
    r.x = fd_act<ACT>(v.x); r.y = fd_act<ACT>(v.y); r.z = fd_act<ACT>(v.z); r.w = fd_act<ACT>(v.w);
    return r;
}
__device__ __forceinline__ fd_f32x4 fd_ld4(const float *p) { return *reinterpret_cast<const fd_f32x4 *>(p); }
__device__ __forceinline__ void fd_st4(float *p, fd_f32x4 v) { *reinterpret_cast<fd_f32x4 *>(p) = v; }
__device__ __forceinline__ fd_f32x4 fd_zero4() { fd_f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
