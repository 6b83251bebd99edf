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

// ---- direct global -> LDS copy (LDS-DMA), counted waits, raw workgroup barrier -------------------------
// fd_glds16: every lane copies 16 bytes from ITS global address to  lds_wave_base + lane*16  (the LDS side is
// wave-uniform base + lane*16 by hardware; a swizzled LDS image is obtained by permuting the per-lane SOURCE).
#ifdef FD_EMU
inline void fd_glds16(const float *g, float *lds_wave_base) { memcpy((char *)lds_wave_base + hipemu::lane_id() * 16, g, 16); }
template <int N> inline void fd_wait_vmcnt() {}
#define FD_SCHED_FENCE() ((void)0)
#define FD_UNIFORM(x) (x)
#define FD_OPAQUE(x) ((void)0)
#define FD_SCALAR_PTR(p) ((void)0)
template <typename P> inline P *fd_uniform_ptr(P *p) { return p; }
inline void fd_store_u32_sbase(char *base, unsigned off, unsigned v) { memcpy(base + off, &v, 4); }
inline void fd_block_barrier_lds() { __syncthreads(); }
inline void fd_block_barrier() { __syncthreads(); }
inline void fd_wave_lds_fence() { __syncthreads(); }         // the emulator runs a wave's lanes one after the other: a full barrier keeps them in step
#else
__device__ __forceinline__ void fd_glds16(const float *g, float *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// wait until at most N of this wave's vector-memory operations (LDS-DMA loads included) are still outstanding
template <int N> __device__ __forceinline__ void fd_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// keeps the compiler's scheduler from moving instructions across this point (source order = issue order)
#define FD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// tells the compiler that the (wave-uniform) integer x lives in a scalar register: address arithmetic derived from it stays on the SALU
#define FD_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// makes the compiler forget what it knows about the integer x: loads addressed through it are not hoisted out of the enclosing loop
// (used to keep per-channel tables out of registers while they are not needed)
#define FD_OPAQUE(x) asm volatile("" : "+v"(x))
// the (wave-uniform) pointer p stays a scalar-register pair the compiler cannot fold lane offsets into: accesses p + (32-bit lane offset) keep the
// "scalar base + vector offset" addressing form instead of a 64-bit address per lane
#define FD_SCALAR_PTR(p) asm volatile("" : "+s"(p))
// 4-byte store to (wave-uniform 64-bit base) + (32-bit lane offset): global_store_dword v_off, v_data, s[base] -- spelled out because the compiler
// re-associates "uniform row base + lane offset" into one 64-bit address per lane and store (v_lshl_add_u64 chains in the tap kernels' epilogues)
__device__ __forceinline__ void fd_store_u32_sbase(char *base, unsigned off, unsigned v)
{
    asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(v), "s"(base) : "memory");
}
// a pointer the caller knows to be wave-uniform, moved into scalar registers (v_readfirstlane of both halves)
template <typename P> __device__ __forceinline__ P *fd_uniform_ptr(P *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<P *>(((unsigned long long)hi << 32) | lo);
}
// a wave's own LDS writes have completed before its following LDS reads are issued (wave-private LDS tiles need no workgroup barrier)
__device__ __forceinline__ void fd_wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// raw s_barrier: unlike __syncthreads() it does not drain vmcnt, so LDS-DMA loads stay in flight across it
__device__ __forceinline__ void fd_block_barrier() { __builtin_amdgcn_s_barrier(); }
// the same, after this wave's own LDS writes/reads have completed (lgkmcnt) -- still without draining vector-memory loads
__device__ __forceinline__ void fd_block_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#endif

// Wave-private LDS / LDS-DMA hand-over inside ONE wave (kernels whose waves share nothing and run different trip counts: no workgroup barrier may be
// used).  Hardware: the wave's own LDS operations (fd_wave_fence) / vector-memory operations incl. LDS-DMA (fd_wave_dma_wait) have completed.
// Emulator: a wave-collective rendezvous (its lanes run one after the other; every lane must have issued its part before any lane reads it).
#ifdef FD_EMU
inline void fd_wave_fence() { (void)__shfl(0.0f, 0); }
inline void fd_wave_dma_wait() { (void)__shfl(0.0f, 0); }
#else
__device__ __forceinline__ void fd_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void fd_wave_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

// XCD-aware placement of a (tiles, channel blocks, images) grid: workgroup number b (x fastest) runs on XCD b % 8 and every XCD has its
// own L2, so with the natural numbering the neighbouring tiles of one image -- whose input halos overlap -- land on eight different L2s and
// every halo is fetched from memory again (measured: 1.4x the algorithmic bytes on the 5x5 decoder layers).  This remapping gives each
// group of eight consecutive workgroup numbers eight different IMAGES: all tiles and channel blocks of an image run on one XCD, and the
// halo re-reads become L2 hits.  Returns the logical (x, y, z) of this workgroup.
struct fd_blk3 { int x, y, z; };
// b = linear workgroup number, per = workgroups per image, nimg = images: returns the image in z and the within-image index in x
__device__ __forceinline__ fd_blk3 fd_xcd_map(unsigned b, unsigned per, unsigned nimg)
{
    const unsigned grp = b / (8u * per), rem = b - grp * 8u * per;
    const unsigned m = nimg - grp * 8u < 8u ? nimg - grp * 8u : 8u;  // images in this group (the last one may be short)
    fd_blk3 r;
    r.z = (int)(grp * 8u + rem % m); r.x = (int)(rem / m); r.y = 0;
    return r;
}
__device__ __forceinline__ fd_blk3 fd_xcd_image_map()      // grid (x, y, images)
{
    const unsigned gx = gridDim.x, gy = gridDim.y;
    fd_blk3 r = fd_xcd_map(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy, gridDim.z);
    r.y = (int)((unsigned)r.x / gx); r.x -= r.y * (int)gx;
    return r;
}
__device__ __forceinline__ fd_blk3 fd_xcd_image_map2()     // grid (x, images)
{
    return fd_xcd_map(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x, gridDim.y);
}

// compile-time integer as a value (selects statically indexed register sets inside generic lambdas)
template <int N> struct fd_int { static constexpr int value = N; };

// Sum over the four lanes {l, l^4, l^8, l^12} of a 16-lane row, result in all four: two row rotates on the VALU's data-parallel-primitive
// path instead of two trips through the LDS crossbar (ds_bpermute); the emulator's butterfly adds the same pairs in the same order.
__device__ __forceinline__ float fd_row_stride4_sum(float v)
{
#ifdef FD_EMU
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 4);
#else
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));   // row_ror:4
#endif
    return v;
}

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
#ifdef FD_EMU
    if (ACT == FD_ACT_RELU6_) return fminf(fmaxf(v, 0.0f), 6.0f);
#else
    if (ACT == FD_ACT_RELU6_) return __builtin_amdgcn_fmed3f(v, 0.0f, 6.0f);     // one v_med3_f32 (same value for every non-NaN input)
#endif
    return v;
}
// the same on a value that comes out of inline asm (fd_dot2_acc): fmaxf / fmed3 would first canonicalise it (a second v_max_f32 per value)
template <int ACT>
__device__ __forceinline__ float fd_act_raw(float v)
{
#ifdef FD_EMU
    return fd_act<ACT>(v);
#else
    float r = v;
    if (ACT == FD_ACT_RELU_) asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    if (ACT == FD_ACT_RELU6_) { const float six = 6.0f; asm("v_med3_f32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(six)); }
    return r;
#endif
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

// ---- 16-bit storage types (activations / pointwise weights; all arithmetic and accumulation stay fp32) ---------------
typedef float fd_f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 fd_half;
struct fd_bf16 { unsigned short v; };                       // raw bfloat16 bits
typedef _Float16 fd_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 fd_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short fd_u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short fd_u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float fd_bf16_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
#ifdef FD_EMU
__device__ __forceinline__ unsigned short fd_f32_to_bf16(float f)
{   // round to nearest even (NaN handling is not needed on this path: inputs are finite)
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// two values -> one 32-bit word (low half = a)
__device__ __forceinline__ unsigned fd_f32x2_to_bf16x2(float a, float b) { return (unsigned)fd_f32_to_bf16(a) | ((unsigned)fd_f32_to_bf16(b) << 16); }
#else
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round to nearest even: the same value as the integer formula above for every finite input)
__device__ __forceinline__ unsigned short fd_f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned fd_f32x2_to_bf16x2(float a, float b)
{
    typedef __bf16 fd_bf16x2_hw __attribute__((ext_vector_type(2)));
    const fd_f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, fd_bf16x2_hw));
}
#endif
// 4 consecutive channels: one 8-byte access
__device__ __forceinline__ fd_f32x4 fd_ld4(const fd_half *p)
{
    const fd_f16x4 h = *reinterpret_cast<const fd_f16x4 *>(p);
    fd_f32x4 r = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    return r;
}
__device__ __forceinline__ void fd_st4(fd_half *p, fd_f32x4 v)
{
    fd_f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    *reinterpret_cast<fd_f16x4 *>(p) = h;
}
__device__ __forceinline__ fd_f32x4 fd_ld4(const fd_bf16 *p)
{
    const fd_u16x4 h = *reinterpret_cast<const fd_u16x4 *>(p);
    fd_f32x4 r = {fd_bf16_to_f32(h.x), fd_bf16_to_f32(h.y), fd_bf16_to_f32(h.z), fd_bf16_to_f32(h.w)};
    return r;
}
__device__ __forceinline__ void fd_st4(fd_bf16 *p, fd_f32x4 v)
{
    typedef unsigned fd_u32x2 __attribute__((ext_vector_type(2)));
    const fd_u32x2 h = {fd_f32x2_to_bf16x2(v.x, v.y), fd_f32x2_to_bf16x2(v.z, v.w)};
    *reinterpret_cast<fd_u32x2 *>(p) = h;
}
// raw (unconverted) 4-channel loads: lets a kernel keep prefetched 16-bit data in half the registers until it is used
__device__ __forceinline__ fd_f32x4 fd_ldraw4(const float *p) { return *reinterpret_cast<const fd_f32x4 *>(p); }
// LDS transpose read (ds_read_b64_tr_b16): every lane passes the address of 4 consecutive 16-bit words; within each 16-lane group those 16 x 4 words
// are a 4 x 16 row-major matrix (lanes 4r .. 4r+3 = row r) and lane j receives its column j -- four values of the SLOW index for one fast index,
// i.e. an MFMA operand piece read straight from an image that was stored the way it came from memory.
#ifdef FD_EMU
inline fd_u16x4 fd_lds_read_tr16(const void *p)
{
    unsigned short o[4];
    hipemu_lds_read_tr16_b64(p, o);
    fd_u16x4 r = {o[0], o[1], o[2], o[3]};
    return r;
}
#else
__device__ __forceinline__ fd_u16x4 fd_lds_read_tr16(const void *p)
{
    typedef short fd_s16x4_hw __attribute__((ext_vector_type(4)));
    const fd_s16x4_hw v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fd_s16x4_hw *)(p));
    return __builtin_bit_cast(fd_u16x4, v);
}
#endif
__device__ __forceinline__ fd_u16x4 fd_ldraw4(const fd_half *p) { return *reinterpret_cast<const fd_u16x4 *>(p); }
__device__ __forceinline__ fd_u16x4 fd_ldraw4(const fd_bf16 *p) { return *reinterpret_cast<const fd_u16x4 *>(p); }
__device__ __forceinline__ fd_f32x4 fd_cvt4(float, fd_f32x4 r) { return r; }
__device__ __forceinline__ fd_f32x4 fd_cvt4(fd_half, fd_u16x4 r)
{
    const fd_f16x4 h = __builtin_bit_cast(fd_f16x4, r);
    fd_f32x4 v = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    return v;
}
__device__ __forceinline__ fd_f32x4 fd_cvt4(fd_bf16, fd_u16x4 r)
{
    fd_f32x4 v = {fd_bf16_to_f32(r.x), fd_bf16_to_f32(r.y), fd_bf16_to_f32(r.z), fd_bf16_to_f32(r.w)};
    return v;
}
__device__ __forceinline__ float fd_ld1(const float *p) { return *p; }
__device__ __forceinline__ float fd_ld1(const fd_half *p) { return (float)*p; }
__device__ __forceinline__ float fd_ld1(const fd_bf16 *p) { return fd_bf16_to_f32(p->v); }
__device__ __forceinline__ void fd_st1(float *p, float v) { *p = v; }
__device__ __forceinline__ void fd_st1(fd_half *p, float v) { *p = (_Float16)v; }
__device__ __forceinline__ void fd_st1(fd_bf16 *p, float v) { p->v = fd_f32_to_bf16(v); }
// Walks a row-major index space of width w in steps of `step` without a division per visit: the staging loops of the LDS-tiled
// depthwise kernels visit patch pixel start, start+step, start+2*step, ... (their integer address arithmetic, not the FMAs, was
// the largest VALU consumer: PMC).
struct fd_px_walk {
    int iy, ix, dy, dx, w;
    __device__ __forceinline__ fd_px_walk(int start, int step, int w_) : w(w_) { iy = start / w_; ix = start - iy * w_; dy = step / w_; dx = step - dy * w_; }
    __device__ __forceinline__ void next() { ix += dx; iy += dy; if (ix >= w) { ix -= w; ++iy; } }
};
__device__ __forceinline__ fd_f32x4 fd_zero4() { fd_f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// Element index of (image n, row y, column x, channel c) of an NHWC tensor [.][H][W][C].  The image part is wave-uniform (n comes from the
// workgroup number: scalar unit); the within-image part fits 32 bits (the plans reject images of >= 2^32 elements) and its two multiplications are
// 24 x 24 bit (y * W + x < 2^24, C < 2^24: v_mul_u32_u24, full rate).  The plain 64-bit expression cost 8 quarter-rate v_mul_lo_u32 / v_mad_u64_u32 per
// staged pixel -- a quarter of the VALU instructions of the depthwise kernels, which PMC shows to be VALU-issue bound (71 % busy).
#ifdef FD_EMU
inline unsigned fd_mul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }      // (masks as the hardware does: an overflow shows up in the CPU tier)
#else
__device__ __forceinline__ unsigned fd_mul24(unsigned a, unsigned b) { return __umul24(a, b); }
#endif
__device__ __forceinline__ long fd_nhwc(int n, int H, int y, int W, int x, int C, int c)
{
    return (long)n * H * W * C + (long)(fd_mul24(fd_mul24((unsigned)y, (unsigned)W) + (unsigned)x, (unsigned)C) + (unsigned)c);
}

// ---- lane vectors of the LDS-tiled depthwise kernels --------------------------------------------------------------------------------
// A work-item of those kernels owns N consecutive channels of a pixel.  fd_lane<T, 4>: the round-1 form -- 4 channels, LDS patches kept in
// fp32 (16 bytes per lane in LDS; 16 bytes per lane in memory only when T is float).  fd_lane<T, 8> (T a 16-bit storage type): 8 channels
// per work-item = 16 bytes per lane in memory AND in LDS, patches kept in the storage type: half the LDS footprint per channel (twice the
// channels per workgroup at the same residency) and half the load / ds_write / ds_read instructions per byte; arithmetic stays fp32 (the
// conversion happens on the LDS read).  A 16-bit plan staged fp32 patches of 4 channels per lane before: its workgroups had the LDS
// footprint and instruction count of the fp32 plan and only their HBM bytes halved (8-byte loads: half the bytes in flight per load).
typedef float fd_f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ fd_f32x8 fd_zero8() { fd_f32x8 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ fd_f32x8 fd_cvt8(fd_half, fd_u16x8 r)
{
    const fd_f16x8 h = __builtin_bit_cast(fd_f16x8, r);
    fd_f32x8 v = {(float)h[0], (float)h[1], (float)h[2], (float)h[3], (float)h[4], (float)h[5], (float)h[6], (float)h[7]};
    return v;
}
__device__ __forceinline__ fd_f32x8 fd_cvt8(fd_bf16, fd_u16x8 r)
{
    // two channels per dword: the low one is a shift, the high one a mask (no unpacking of the 16-bit halves first)
    typedef unsigned fd_u32x4_ __attribute__((ext_vector_type(4)));
    const fd_u32x4_ d = __builtin_bit_cast(fd_u32x4_, r);
    fd_f32x8 v = {__builtin_bit_cast(float, d[0] << 16), __builtin_bit_cast(float, d[0] & 0xffff0000u), __builtin_bit_cast(float, d[1] << 16), __builtin_bit_cast(float, d[1] & 0xffff0000u),
                  __builtin_bit_cast(float, d[2] << 16), __builtin_bit_cast(float, d[2] & 0xffff0000u), __builtin_bit_cast(float, d[3] << 16), __builtin_bit_cast(float, d[3] & 0xffff0000u)};
    return v;
}
__device__ __forceinline__ fd_u16x8 fd_pack8v(fd_half, fd_f32x8 v)
{
    fd_f16x8 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
    return __builtin_bit_cast(fd_u16x8, h);
}
__device__ __forceinline__ fd_u16x8 fd_pack8v(fd_bf16, fd_f32x8 v)
{
    typedef unsigned fd_u32x4_ __attribute__((ext_vector_type(4)));
    const fd_u32x4_ r = {fd_f32x2_to_bf16x2(v[0], v[1]), fd_f32x2_to_bf16x2(v[2], v[3]), fd_f32x2_to_bf16x2(v[4], v[5]), fd_f32x2_to_bf16x2(v[6], v[7])};
    return __builtin_bit_cast(fd_u16x8, r);
}
__device__ __forceinline__ fd_u16x8 fd_sum8(fd_half, fd_u16x8 a, fd_u16x8 b)
{
    return __builtin_bit_cast(fd_u16x8, __builtin_bit_cast(fd_f16x8, a) + __builtin_bit_cast(fd_f16x8, b));
}
__device__ __forceinline__ fd_u16x8 fd_sum8(fd_bf16, fd_u16x8 a, fd_u16x8 b) { return fd_pack8v(fd_bf16{}, fd_cvt8(fd_bf16{}, a) + fd_cvt8(fd_bf16{}, b)); }
template <int ACT>
__device__ __forceinline__ fd_f32x8 fd_act4(fd_f32x8 v)      // (same name as the 4-channel form: the kernels are generic in the lane width)
{
    fd_f32x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = fd_act<ACT>(v[j]);
    return r;
}
// value a T-typed store will hold (the train kernels take their statistics over the STORED, i.e. rounded, tensor)
__device__ __forceinline__ fd_f32x4 fd_round4(float, fd_f32x4 v) { return v; }
__device__ __forceinline__ fd_f32x4 fd_round4(fd_bf16, fd_f32x4 v)
{
    fd_f32x4 r = {fd_bf16_to_f32(fd_f32_to_bf16(v.x)), fd_bf16_to_f32(fd_f32_to_bf16(v.y)), fd_bf16_to_f32(fd_f32_to_bf16(v.z)), fd_bf16_to_f32(fd_f32_to_bf16(v.w))};
    return r;
}
__device__ __forceinline__ fd_f32x4 fd_round4(fd_half, fd_f32x4 v)
{
    fd_f32x4 r = {(float)(_Float16)v.x, (float)(_Float16)v.y, (float)(_Float16)v.z, (float)(_Float16)v.w};
    return r;
}
template <typename T, int N> struct fd_lane;
template <typename T> struct fd_lane<T, 4> {
    static constexpr int n = 4;
    typedef fd_f32x4 vec;
    typedef float lds_t;                                     // element type of the LDS patch images
    typedef decltype(fd_ldraw4((const T *)nullptr)) raw;     // a lane's channels as loaded, before conversion
    static __device__ __forceinline__ vec zero() { return fd_zero4(); }
    static __device__ __forceinline__ vec ld(const T *p) { return fd_ld4(p); }
    static __device__ __forceinline__ raw ldraw(const T *p) { return fd_ldraw4(p); }
    static __device__ __forceinline__ vec cvt(raw r) { return fd_cvt4(T{}, r); }
    static __device__ __forceinline__ void st(T *p, vec v) { fd_st4(p, v); }
    static __device__ __forceinline__ vec ldf(const float *p) { return fd_ld4(p); }          // fp32 tables / taps (global or LDS)
    static __device__ __forceinline__ void stf(float *p, vec v) { fd_st4(p, v); }
    static __device__ __forceinline__ vec lds_ld(const lds_t *p) { return fd_ld4(p); }
    static __device__ __forceinline__ void lds_st(lds_t *p, vec v) { fd_st4(p, v); }
    static __device__ __forceinline__ void lds_st_raw(lds_t *p, raw r) { fd_st4(p, fd_cvt4(T{}, r)); }
    static __device__ __forceinline__ void lds_st_sum(lds_t *p, raw a, raw b) { fd_st4(p, fd_cvt4(T{}, a) + fd_cvt4(T{}, b)); }   // up2(low) + skip
    static __device__ __forceinline__ vec round(vec v) { return fd_round4(T{}, v); }          // value a T store will hold
    static __device__ __forceinline__ vec lds_round(vec v) { return v; }                      // value an LDS patch store will hold
};
template <typename T> struct fd_lane<T, 8> {
    static constexpr int n = 8;
    typedef fd_f32x8 vec;
    typedef T lds_t;
    typedef fd_u16x8 raw;
    static __device__ __forceinline__ vec zero() { return fd_zero8(); }
    static __device__ __forceinline__ raw ldraw(const T *p) { return *reinterpret_cast<const fd_u16x8 *>(p); }
    static __device__ __forceinline__ vec cvt(raw r) { return fd_cvt8(T{}, r); }
    static __device__ __forceinline__ vec ld(const T *p) { return cvt(ldraw(p)); }
    static __device__ __forceinline__ void st(T *p, vec v) { *reinterpret_cast<fd_u16x8 *>(p) = fd_pack8v(T{}, v); }
    static __device__ __forceinline__ vec ldf(const float *p)
    {
        const fd_f32x4 a = fd_ld4(p), b = fd_ld4(p + 4);
        vec v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        return v;
    }
    static __device__ __forceinline__ void stf(float *p, vec v)
    {
        const fd_f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        fd_st4(p, a); fd_st4(p + 4, b);
    }
    static __device__ __forceinline__ vec lds_ld(const lds_t *p) { return cvt(*reinterpret_cast<const fd_u16x8 *>(p)); }
    static __device__ __forceinline__ void lds_st(lds_t *p, vec v) { *reinterpret_cast<fd_u16x8 *>(p) = fd_pack8v(T{}, v); }
    static __device__ __forceinline__ void lds_st_raw(lds_t *p, raw r) { *reinterpret_cast<fd_u16x8 *>(p) = r; }
    // up2(low) + skip, rounded to the storage type.  fp16: four v_pk_add_f16 -- the sum of two fp16 values is exact in fp32, so one fp16 addition
    // rounds exactly as "convert both, add in fp32, round" does, for 4 instructions instead of 16 conversions + 4 packed adds + 4 packed conversions
    static __device__ __forceinline__ void lds_st_sum(lds_t *p, raw a, raw b) { *reinterpret_cast<fd_u16x8 *>(p) = fd_sum8(T{}, a, b); }
    static __device__ __forceinline__ vec round(vec v) { return cvt(fd_pack8v(T{}, v)); }
    static __device__ __forceinline__ vec lds_round(vec v) { return cvt(fd_pack8v(T{}, v)); }
};

// ---- raw buffer access: (128-bit resource = base + byte count) + 32-bit lane offset + scalar offset ----------------------------------------------
// buffer_load_dword v, v_off, s[rsrc], s_off offen: the addressing form the row-walking kernels want (a lane holds a few 32-bit offsets for its whole
// band, everything that changes per row is scalar), and the hardware's range check does the horizontal zero padding: a lane offset >= the byte count
// (FD_BUF_OOB) reads 0 and drops stores.  (Plain pointers were tried first: the compiler re-associates "uniform row base + lane offset" into a
// 64-bit address per lane and access -- 170 v_lshl_add_u64 per 5 rows and twice the offset registers.)
#define FD_BUF_OOB 0x80000000u
#ifdef FD_EMU
struct fd_bufrsrc { char *base; unsigned bytes; };
inline fd_bufrsrc fd_make_rsrc(const void *p, unsigned bytes) { fd_bufrsrc r = {const_cast<char *>(static_cast<const char *>(p)), bytes}; return r; }
inline unsigned fd_buf_ld32(fd_bufrsrc r, unsigned voff, unsigned soff)
{
    if (voff >= r.bytes || (unsigned long long)voff + soff + 4 > r.bytes) return 0u;
    unsigned v; memcpy(&v, r.base + voff + soff, 4); return v;
}
inline void fd_buf_st32(fd_bufrsrc r, unsigned voff, unsigned soff, unsigned v)
{
    if (voff >= r.bytes || (unsigned long long)voff + soff + 4 > r.bytes) return;
    memcpy(r.base + voff + soff, &v, 4);
}
#else
typedef __amdgpu_buffer_rsrc_t fd_bufrsrc;
__device__ __forceinline__ fd_bufrsrc fd_make_rsrc(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ unsigned fd_buf_ld32(fd_bufrsrc r, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0); }
__device__ __forceinline__ void fd_buf_st32(fd_bufrsrc r, unsigned voff, unsigned soff, unsigned v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, (int)soff, 0); }
#endif
// ---- pixel-pair arithmetic of the 16-bit 5x5 kernels (fd_kernels_dw5p.h) ---------------------------------------------------------------
// A "pair" is one 32-bit word holding the SAME channel of two horizontally adjacent pixels (low half = the even pixel) in the storage type.
// Depthwise taps then run on v_dot2_f32_{f16,bf16}: two 16-bit multiply-accumulates into an fp32 accumulator per VALU slot with no conversion
// instructions (tools/microbench/valu_rates.hip: the same 3 shader cycles as one v_pk_fma_f32), against 16-bit taps packed the same way.
#ifdef FD_EMU
// v_perm_b32: result byte k = byte sel[k] of the 8 bytes {s0 (bytes 4-7), s1 (bytes 0-3)}
inline unsigned fd_perm(unsigned s0, unsigned s1, unsigned sel)
{
    const unsigned long long cat = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int k = 0; k < 4; ++k) r |= (unsigned)((cat >> (8 * ((sel >> (8 * k)) & 7u))) & 0xffu) << (8 * k);
    return r;
}
inline float fd_h16_bits_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
inline float fd_dot2(fd_half, unsigned a, unsigned b, float c)
{
    return (float)((double)fd_h16_bits_to_f32((unsigned short)a) * fd_h16_bits_to_f32((unsigned short)b) +
                   (double)fd_h16_bits_to_f32((unsigned short)(a >> 16)) * fd_h16_bits_to_f32((unsigned short)(b >> 16)) + (double)c);
}
inline float fd_dot2(fd_bf16, unsigned a, unsigned b, float c)
{
    return (float)((double)fd_bf16_to_f32((unsigned short)a) * fd_bf16_to_f32((unsigned short)b) +
                   (double)fd_bf16_to_f32((unsigned short)(a >> 16)) * fd_bf16_to_f32((unsigned short)(b >> 16)) + (double)c);
}
#else
__device__ __forceinline__ unsigned fd_perm(unsigned s0, unsigned s1, unsigned sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ float fd_dot2(fd_half, unsigned a, unsigned b, float c)
{
    typedef _Float16 fd_h2_hw __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(fd_h2_hw, a), __builtin_bit_cast(fd_h2_hw, b), c, false);
}
__device__ __forceinline__ float fd_dot2(fd_bf16, unsigned a, unsigned b, float c)
{
    typedef __bf16 fd_b2_hw __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fd_b2_hw, a), __builtin_bit_cast(fd_b2_hw, b), c, false);
}
#endif
// The tap loops of fd_kernels_dw5p.h issue their dot2 in SOURCE order (volatile asm): eight independent accumulation chains interleaved.  Left to
// itself the compiler runs the chains two at a time (its scheduler minimises live registers; the intrinsic is pure, so sched_barrier does not pin it).
// fd_dot2_first starts a chain from a third operand (v_dot2_f32_*: no copy of the bias into the accumulator first).
// HAZARD the compiler cannot see: on gfx90a / gfx940 / gfx950 a register written by a DOT instruction must not be read by a DIFFERENT kind of VALU instruction
// within 3 wait states (LLVM's GCNHazardRecognizer inserts the s_nop for instructions it knows; inline asm is opaque to it).  Chained dot2 on the same
// accumulator are fine (the accumulator operand is forwarded); the FIRST other reader is not -- and the compiler is free to schedule that reader right
// behind the accumulator's last dot2, in between the remaining volatile asm statements.  fd_dot2_done(acc...) is the barrier: a wait of 5 states that every
// accumulator passes THROUGH (in/out operands of empty volatile asm behind the s_nop), so no reader can be placed before it.  (Measured, round 6:
// fd_dw5_bwd_rows lost the last tap of one accumulator on hardware -- 4-28 % errors -- while the CPU emulation was exact; tools/microbench/dw5bwd_check.cpp.)
#ifdef FD_EMU
template <int N> inline void fd_dot2_done(float (&)[N][2]) {}
#else
template <int N> __device__ __forceinline__ void fd_dot2_done(float (&acc)[N][2])
{
    asm volatile("s_nop 4" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) { asm volatile("" : "+v"(acc[i][0])); asm volatile("" : "+v"(acc[i][1])); }
}
#endif
#ifdef FD_EMU
template <typename T> inline void fd_dot2_acc(T, unsigned a, unsigned b, float &acc) { acc = fd_dot2(T{}, a, b, acc); }
template <typename T> inline float fd_dot2_first(T, unsigned a, unsigned b, float c) { return fd_dot2(T{}, a, b, c); }
#else
__device__ __forceinline__ void fd_dot2_acc(fd_half, unsigned a, unsigned b, float &acc) { asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
__device__ __forceinline__ void fd_dot2_acc(fd_bf16, unsigned a, unsigned b, float &acc) { asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
__device__ __forceinline__ float fd_dot2_first(fd_half, unsigned a, unsigned b, float c)
{
    float r;
    asm volatile("v_dot2_f32_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float fd_dot2_first(fd_bf16, unsigned a, unsigned b, float c)
{
    float r;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#endif
#define FD_PERM_LO 0x05040100u     /* fd_perm(b, a, FD_PERM_LO) = (low half of a, low half of b) */
#define FD_PERM_HI 0x07060302u     /* fd_perm(b, a, FD_PERM_HI) = (high half of a, high half of b) */
// two fp32 values -> one word of the storage type (low half = a), round to nearest even
__device__ __forceinline__ unsigned fd_pack2(fd_half, float a, float b)
{
    typedef _Float16 fd_h2_ __attribute__((ext_vector_type(2)));
    const fd_h2_ h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ unsigned fd_pack2(fd_bf16, float a, float b) { return fd_f32x2_to_bf16x2(a, b); }
// element-wise sum of two words of the storage type, rounded to the storage type (fp16: one v_pk_add_f16 -- the sum of two fp16 values is exact
// in fp32, so this is "convert, add in fp32, round"; bf16: that sequence spelled out)
__device__ __forceinline__ unsigned fd_pair_sum(fd_half, unsigned a, unsigned b)
{
    typedef _Float16 fd_h2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(fd_h2_, a) + __builtin_bit_cast(fd_h2_, b));
}
__device__ __forceinline__ unsigned fd_pair_sum(fd_bf16, unsigned a, unsigned b)
{
    return fd_f32x2_to_bf16x2(__builtin_bit_cast(float, a << 16) + __builtin_bit_cast(float, b << 16),
                              __builtin_bit_cast(float, a & 0xffff0000u) + __builtin_bit_cast(float, b & 0xffff0000u));
}

// ---- BatchNorm statistics rows (round 5): order-independent, bit-reproducible accumulation of per-workgroup partial sums ----------
// The train step's BatchNorm reductions -- forward (sum z, sum z^2), backward (sum G, sum G*xhat) -- used to be "one fp32 partial row per
// workgroup -> a finalisation launch that sums 25 ... 6272 rows -> a table": 76 launches at the per-launch floor per step.  Now every
// producer workgroup ADDS its partial sums into a few shared rows with 64-bit INTEGER no-return atomics, and the consumer kernels derive their
// per-channel coefficients from those rows in their prologue (fd_kernels_train.h: fd_stat_table_block).  Integer addition is associative and
// commutative, so the totals -- and everything derived from them -- are the same bits whatever order the workgroups arrive in.
//
// Fixed point without a range / precision compromise: a partial v (fp32) is placed EXACTLY -- mantissa shifted, no rounding -- into one of
// FD_STAT_BINS accumulators chosen by its own binary exponent; bin b holds multiples of 2^-frac[b]:
//     forward  (sums of z, z^2):        |v| < 2^-8  -> frac 56,   2^-8  <= |v| < 2^16 -> frac 32,   |v| >= 2^16 -> frac 8
//     backward (sums of G, G * xhat):   |v| < 2^-32 -> frac 80,   2^-32 <= |v| < 2^-8 -> frac 56,   |v| >= 2^-8  -> frac 32
// so a 24-bit mantissa lands at bit positions < 2^48 of an int64 and 2^14 partials fit with room to spare, across 72 binary orders of magnitude
// (below the lowest bin's 2^-33 * 2^-frac0 ... values lose low bits: < 2^-89 (forward) / 2^-113 (backward) each; fp32 denormals count as 0; above 2^40
// (forward) / 2^16 (backward) per PARTIAL the value saturates -- a diverged network).  The total of a column is then the EXACT sum of the fp32 partials,
// reconstructed in double from the three bins (fd_stat_total).
// One address takes an atomic every ~22 ns whatever the scope (tools/microbench/stat_atomics.hip, MI355X: 6272 workgroups x 128 columns into ONE row
// 135 us, into 8 rows 18 us), so a unit's partials are dealt to nr rows (a power of two <= 16, chosen by the plan so that an address sees <= ~128
// adds (FD_STAT_ADDS_PER_ROW, fd_train_plan.h): row = workgroup number & (nr - 1)) and the consumer adds the nr x 3 integers of a column.
// Layout of a unit's rows: int64 [nr][FD_STAT_BINS][2][cs]  (2 = first / second sum; cs = channel pitch >= C, a multiple of 16: the memory side
// serialises atomics per 128-byte LINE and instruction, so no two (row, bin, sum) slots share a line).  The plan zeroes all rows of a step with one memset.
#define FD_STAT_BINS 3
#define FD_STAT_POISON (1LL << 62)      /* added to the highest bin by an Inf / NaN partial */
#define FD_STAT_MAX_ROWS 16
#define FD_STAT_FWD 0
#define FD_STAT_BWD 1
struct fd_stat_rows { long long *rows; int nr, cs; };    // nr: power of two; cs: channel pitch of the rows (>= C; >= 16 so that every [row][bin][sum] slot starts its own 128-byte line)
#ifdef FD_EMU
inline void fd_atomic_add_i64(long long *p, long long v) { *p += v; }
#else
__device__ __forceinline__ void fd_atomic_add_i64(long long *p, long long v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // result unused: a no-return global_atomic_add_x2
}
#endif
template <int DIR> struct fd_stat_fmt {
    static constexpr int lo = DIR == FD_STAT_FWD ? -8 : -32, hi = DIR == FD_STAT_FWD ? 16 : -8;     // bin boundaries (binary exponents)
    static constexpr int f0 = DIR == FD_STAT_FWD ? 56 : 80, f1 = f0 - 24, f2 = f0 - 48;
};
// adds the fp32 partial v of column c (sum `which`) of workgroup number blk to the unit's rows
template <int DIR>
__device__ __forceinline__ void fd_stat_add(const fd_stat_rows &d, long blk, int /*C*/, int which, int c, float v)
{
    typedef fd_stat_fmt<DIR> F;
    const unsigned u = __builtin_bit_cast(unsigned, v);
    int e = (int)((u >> 23) & 255u) - 127;
    if (e == -127) return;                                  // +-0 and denormals: nothing to add
    if (e == 128) {                                         // Inf / NaN partial (a diverged step): POISON the column -- fd_stat_total returns NaN, as nn.BatchNorm2d's
        fd_atomic_add_i64(d.rows + ((((long)(blk & (d.nr - 1)) * FD_STAT_BINS + 2) * 2 + which) * d.cs + c), FD_STAT_POISON);   // statistics would be
        return;
    }
    const int bin = e < F::lo ? 0 : (e < F::hi ? 1 : 2);
    const int frac = bin == 0 ? F::f0 : (bin == 1 ? F::f1 : F::f2);
    if (e > 47 - frac) e = 47 - frac;                       // (saturation, highest bin only: |partial| >= 2^40 forward / 2^16 backward)
    const long long m = (long long)((u & 0x7fffffu) | 0x800000u);
    const int sh = e + frac - 23;                           // >= 0 except below the lowest bin's exact range
    long long iv = sh >= 0 ? (m << sh) : (sh > -24 ? (m >> -sh) : 0);
    if (u >> 31) iv = -iv;
    if (iv == 0) return;
    fd_atomic_add_i64(d.rows + ((((long)(blk & (d.nr - 1)) * FD_STAT_BINS + bin) * 2 + which) * d.cs + c), iv);
}
// the exact total of column c (sum `which`) over the partials in rows r0, r0 + rstep, ... < nr (r0 = 0, rstep = 1: of the whole unit), as a double
template <int DIR>
__device__ __forceinline__ double fd_stat_total(const long long *__restrict__ rows, int nr, int C /* channel pitch */, int which, int c, int r0, int rstep)
{
    typedef fd_stat_fmt<DIR> F;
    long long a0 = 0, a1 = 0, a2 = 0;
    for (int r = r0; r < nr; r += rstep) {
        const long long *p = rows + (((long)r * FD_STAT_BINS) * 2 + which) * C + c;
        a0 += p[0]; a1 += p[2 * (long)C]; a2 += p[4 * (long)C];
    }
    // (a poisoned column: the highest bin holds at most 2^14 partials below 2^48 in magnitude, i.e. |a2| < 2^62 unless FD_STAT_POISON = 2^62 was added)
    if (a2 >= FD_STAT_POISON / 2 || a2 <= -(FD_STAT_POISON / 2)) return __builtin_nan("");
    return ldexp((double)a0, -F::f0) + ldexp((double)a1, -F::f1) + ldexp((double)a2, -F::f2);
}

// ---- device-coherent accesses ("last arriver" reductions: stream-K partial tiles, fused two-level reductions of the train step) ----
// Device-coherent accesses for partial results and their arrival counters.  MI355X has one L2 per XCD and the L2s are not coherent
// with each other inside a kernel; an agent-scope FENCE would make them so by writing back / invalidating the whole L2
// (buffer_wbl2 / buffer_inv: measured +100 us per launch here, the L2 is full of freshly written activations).  Instead every
// access to the scratch slots and counters is itself agent-scope (sc1: performed at the device coherence point, bypassing the
// non-coherent L2 lines), and the only ordering needed -- partial stores complete before the counter moves -- is an explicit
// s_waitcnt vmcnt(0) (no cache maintenance) followed by the workgroup barrier: the "sc1 payload -> drained vmcnt -> sc1 flag" hand-off
// of MI355X_MICROARCH.md (a workgroup-scope release FENCE is not enough: outside tgsplit mode it does not wait for vmcnt, so the
// counter could overtake the slice stores; inline asm because the compiler drops a fence's wait when it believes the scoreboard empty).
// The reader needs no acquire: its loads are sc1 as well (served by the coherence point, never by its L1 / a stale L2 line).
#ifdef FD_EMU
inline int fd_atomic_inc(int *p) { int o = *p; *p = o + 1; return o; }
inline void fd_store_dev(float *p, float v) { *p = v; }
inline float fd_load_dev(const float *p) { return *p; }
inline void fd_store_dev(int *p, int v) { *p = v; }
inline void fd_store_dev(double *p, double v) { *p = v; }
inline double fd_load_dev(const double *p) { return *p; }
inline void fd_release_wg() {}
inline void fd_acquire_wg() {}
#else
__device__ __forceinline__ int fd_atomic_inc(int *p) { return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fd_store_dev(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float fd_load_dev(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fd_store_dev(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fd_store_dev(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double fd_load_dev(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fd_release_wg() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }          // this wave's sc1 stores have been performed
__device__ __forceinline__ void fd_acquire_wg() { asm volatile("" ::: "memory"); }                            // compiler ordering only: the loads that follow are sc1
#endif
