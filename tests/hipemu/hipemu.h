/*
 * hipemu.h -- a tiny single-threaded CPU emulator of the HIP execution model.  TEST INFRASTRUCTURE ONLY.
 *
 * Purpose: there is no GPU in the build container and GPU minutes are scarce, so the product's HIP
 * kernels (fast-depth_amd/csrc/*.h) and its host-side plan/launch code are ALSO compiled, unchanged,
 * against this header (clang++ -DFD_EMU) into tests/hipemu/_build/libfastdepth_emu.so.  The
 * `-m "not gpu"` tests drive that library through the same C ABI on tiny shapes to catch indexing,
 * tiling, barrier and MFMA-fragment-layout bugs before a kernel ever reaches the MI355X.  It is
 * never part of the product: libfastdepth_hip.so contains none of this, and the Python engine
 * refuses to load the emulation library.
 *
 * Model: one workgroup at a time; each work-item is a ucontext fiber.  __syncthreads() and the
 * wave-collective operations (__shfl_xor, MFMA) are cooperative yields:
 *   - block barrier: a fiber yields with reason BARRIER; the block resumes when every live fiber has.
 *   - wave collective: deposit operands in the wave's exchange area, yield (COLLECTIVE), then every
 *     lane computes its own result from all 64 lanes' deposits, yield again before the area is reused.
 * MFMA lane<->element layouts follow /opt/skills/guides/cdna_hip_programming.md section 3 and were
 * confirmed on an MI355X (scratch/probe, round 1): for v_mfma_f32_32x32x2_f32 lane l holds
 * A[l&31][l>>5], B[l>>5][l&31] and D[(r&3)+8*(r>>2)+4*(l>>5)][l&31] in register r.
 */
#ifndef HIPEMU_H
#define HIPEMU_H

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
#define hipSuccess 0
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static   /* one workgroup runs at a time, its fibers share the static */

namespace hipemu {

enum Reason { RUNNING = 0, BARRIER = 1, COLLECTIVE = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    dim3 tid;
    int reason = RUNNING;
};

struct WaveXchg {
    float f[64][34];   // per-lane deposit area (a, b, c[16] / shuffle value ...)
};

struct State {
    dim3 grid, block, bid;
    std::vector<Fiber> fibers;
    std::vector<WaveXchg> xchg;     // one per wave
    ucontext_t sched;
    Fiber *cur = nullptr;
    int cur_index = 0;
    unsigned char *dyn_smem = nullptr;
    size_t dyn_smem_bytes = 0;
    std::function<void()> body;
};

inline State &st() { static State s; return s; }

inline void yield(int reason)
{
    State &s = st();
    s.cur->reason = reason;
    swapcontext(&s.cur->ctx, &s.sched);
}

inline void fiber_entry()
{
    State &s = st();
    s.body();
    s.cur->reason = DONE;
    swapcontext(&s.cur->ctx, &s.sched);
}

static const size_t kStack = 256 * 1024;

/* Runs `body` once per work-item of every workgroup of the grid. */
inline void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()> &body)
{
    State &s = st();
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads % 64 != 0) { fprintf(stderr, "hipemu: block size %d is not a multiple of 64\n", nthreads); abort(); }
    s.grid = grid; s.block = block; s.body = body;
    if ((int)s.fibers.size() < nthreads) {
        size_t old = s.fibers.size();
        s.fibers.resize(nthreads);
        for (size_t i = old; i < s.fibers.size(); ++i) s.fibers[i].stack = (char *)malloc(kStack);
    }
    s.xchg.resize(nthreads / 64);
    if (s.dyn_smem_bytes < smem_bytes + 64) {
        free(s.dyn_smem);
        s.dyn_smem_bytes = smem_bytes + 64;
        s.dyn_smem = (unsigned char *)aligned_alloc(64, (s.dyn_smem_bytes + 63) / 64 * 64);
    }
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                s.bid = dim3(bx, by, bz);
                memset(s.dyn_smem, 0xCD, smem_bytes);   // poison: reads of never-written LDS show up as garbage
                for (int t = 0; t < nthreads; ++t) {
                    Fiber &f = s.fibers[t];
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.reason = RUNNING;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &s.sched;
                    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
                }
                int live = nthreads;
                while (live > 0) {
                    int at_barrier = 0;
                    live = 0;
                    for (int w = 0; w < nthreads / 64; ++w) {
                        /* run this wave until every lane is at a block barrier or done */
                        for (;;) {
                            int coll = 0, other = 0;
                            for (int l = 0; l < 64; ++l) {
                                Fiber &f = s.fibers[w * 64 + l];
                                if (f.reason == DONE || f.reason == BARRIER) { ++other; continue; }
                                s.cur = &f; s.cur_index = w * 64 + l;
                                swapcontext(&s.sched, &f.ctx);
                                if (f.reason == COLLECTIVE) { ++coll; f.reason = RUNNING; } else ++other;
                            }
                            if (coll == 0) break;
                            if (other != 0 && coll != 0) {
                                /* lanes of one wave diverged around a collective: legal only if the others are done/at barrier
                                   on real hardware too (inactive lanes); we allow it but their deposits are stale. */
                            }
                        }
                        for (int l = 0; l < 64; ++l) {
                            Fiber &f = s.fibers[w * 64 + l];
                            if (f.reason == BARRIER) ++at_barrier;
                            if (f.reason != DONE) ++live;
                        }
                    }
                    if (live > 0 && at_barrier != live) {
                        fprintf(stderr, "hipemu: barrier divergence in block (%u,%u,%u): %d of %d live threads at barrier\n",
                                bx, by, bz, at_barrier, live);
                        abort();
                    }
                    for (int t = 0; t < nthreads; ++t)
                        if (s.fibers[t].reason == BARRIER) s.fibers[t].reason = RUNNING;
                }
            }
}

inline int lane_id() { return st().cur_index & 63; }
inline WaveXchg &wave_xchg() { return st().xchg[st().cur_index >> 6]; }

}  // namespace hipemu

#define threadIdx (hipemu::st().cur->tid)
#define blockIdx (hipemu::st().bid)
#define blockDim (hipemu::st().block)
#define gridDim (hipemu::st().grid)
#define FD_DYN_SMEM(name) unsigned char *name = hipemu::st().dyn_smem

inline void __syncthreads() { hipemu::yield(hipemu::BARRIER); }

inline float __shfl_xor(float v, int mask)
{
    hipemu::WaveXchg &x = hipemu::wave_xchg();
    const int l = hipemu::lane_id();
    x.f[l][0] = v;
    hipemu::yield(hipemu::COLLECTIVE);
    const float r = x.f[l ^ mask][0];
    hipemu::yield(hipemu::COLLECTIVE);
    return r;
}

// ds_read_b64_tr_b16 (gfx950): within each 16-lane group the lanes' 4 x 16-bit words form a 4 x 16 matrix (lanes 4r .. 4r+3 hold row r, each 4
// consecutive elements at ITS OWN address); lane j of the group receives column j (probed on an MI355X: tools/microbench/tr_read_probe.hip)
inline void hipemu_lds_read_tr16_b64(const void *p, unsigned short (&out)[4])
{
    hipemu::WaveXchg &x = hipemu::wave_xchg();
    const int l = hipemu::lane_id();
    memcpy(&x.f[l][0], &p, sizeof p);
    hipemu::yield(hipemu::COLLECTIVE);
    const int g = l & ~15, j = l & 15;
    for (int k = 0; k < 4; ++k) {
        const void *q;
        memcpy(&q, &x.f[g + 4 * k + (j >> 2)][0], sizeof q);
        out[k] = reinterpret_cast<const unsigned short *>(q)[j & 3];
    }
    hipemu::yield(hipemu::COLLECTIVE);
}

inline float __shfl(float v, int src_lane)
{
    hipemu::WaveXchg &x = hipemu::wave_xchg();
    const int l = hipemu::lane_id();
    x.f[l][0] = v;
    hipemu::yield(hipemu::COLLECTIVE);
    const float r = x.f[src_lane & 63][0];
    hipemu::yield(hipemu::COLLECTIVE);
    return r;
}

typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));

/* v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, k-ordered fmaf chain (guide section 3). */
inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int)
{
    hipemu::WaveXchg &x = hipemu::wave_xchg();
    const int l = hipemu::lane_id();
    x.f[l][0] = a; x.f[l][1] = b;
    hipemu::yield(hipemu::COLLECTIVE);
    hipemu_f32x16 d;
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(x.f[row + 32 * k][0], x.f[col + 32 * k][1], acc);
        d[r] = acc;
    }
    hipemu::yield(hipemu::COLLECTIVE);
    return d;
}

/* v_mfma_f32_16x16x4_f32: lane l holds A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)*4+r][l&15]. */
inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int)
{
    hipemu::WaveXchg &x = hipemu::wave_xchg();
    const int l = hipemu::lane_id();
    x.f[l][0] = a; x.f[l][1] = b;
    hipemu::yield(hipemu::COLLECTIVE);
    hipemu_f32x4 d;
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(x.f[row + 16 * k][0], x.f[col + 16 * k][1], acc);
        d[r] = acc;
    }
    hipemu::yield(hipemu::COLLECTIVE);
    return d;
}

/* v_mfma_f32_32x32x16_{f16,bf16}: lane l holds A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31], j = 0..7 (layout confirmed on an
 * MI355X, scratch/probe); D as for the f32 32x32 form.  Products and sums in fp32 (the hardware accumulates in fp32). */
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short hipemu_u16x8 __attribute__((ext_vector_type(8)));
inline hipemu_f32x16 hipemu_mfma_32x32x16_generic(const float (&av)[8], const float (&bv)[8], hipemu_f32x16 c)
{
    hipemu::WaveXchg &x = hipemu::wave_xchg();
    const int l = hipemu::lane_id();
    for (int j = 0; j < 8; ++j) { x.f[l][j] = av[j]; x.f[l][8 + j] = bv[j]; }
    hipemu::yield(hipemu::COLLECTIVE);
    hipemu_f32x16 d;
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc = fmaf(x.f[row + 32 * (k >> 3)][k & 7], x.f[col + 32 * (k >> 3)][8 + (k & 7)], acc);
        d[r] = acc;
    }
    hipemu::yield(hipemu::COLLECTIVE);
    return d;
}
inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x16 c, int, int, int)
{
    float av[8], bv[8];
    for (int j = 0; j < 8; ++j) { av[j] = (float)a[j]; bv[j] = (float)b[j]; }
    return hipemu_mfma_32x32x16_generic(av, bv, c);
}
inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_bf16(hipemu_u16x8 a, hipemu_u16x8 b, hipemu_f32x16 c)
{
    float av[8], bv[8];
    for (int j = 0; j < 8; ++j) {
        unsigned ua = (unsigned)a[j] << 16, ub = (unsigned)b[j] << 16;
        memcpy(&av[j], &ua, 4); memcpy(&bv[j], &ub, 4);
    }
    return hipemu_mfma_32x32x16_generic(av, bv, c);
}

/* v_mfma_f32_16x16x32_{f16,bf16} (gfx950): lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15], j = 0..7; D[(l>>4)*4+r][l&15]
 * (C/D layout: cdna_hip_programming.md, "Fragment layout"; any k numbering common to A and B gives the same sums). */
inline hipemu_f32x4 hipemu_mfma_16x16x32_generic(const float (&av)[8], const float (&bv)[8], hipemu_f32x4 c)
{
    hipemu::WaveXchg &x = hipemu::wave_xchg();
    const int l = hipemu::lane_id();
    for (int j = 0; j < 8; ++j) { x.f[l][j] = av[j]; x.f[l][8 + j] = bv[j]; }
    hipemu::yield(hipemu::COLLECTIVE);
    hipemu_f32x4 d;
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc = fmaf(x.f[row + 16 * (k >> 3)][k & 7], x.f[col + 16 * (k >> 3)][8 + (k & 7)], acc);
        d[r] = acc;
    }
    hipemu::yield(hipemu::COLLECTIVE);
    return d;
}
inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_f16(hipemu_u16x8 a, hipemu_u16x8 b, hipemu_f32x4 c)
{
    float av[8], bv[8];
    const hipemu_f16x8 ha = __builtin_bit_cast(hipemu_f16x8, a), hb = __builtin_bit_cast(hipemu_f16x8, b);
    for (int j = 0; j < 8; ++j) { av[j] = (float)ha[j]; bv[j] = (float)hb[j]; }
    return hipemu_mfma_16x16x32_generic(av, bv, c);
}
inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_bf16(hipemu_u16x8 a, hipemu_u16x8 b, hipemu_f32x4 c)
{
    float av[8], bv[8];
    for (int j = 0; j < 8; ++j) {
        unsigned ua = (unsigned)a[j] << 16, ub = (unsigned)b[j] << 16;
        memcpy(&av[j], &ua, 4); memcpy(&bv[j], &ub, 4);
    }
    return hipemu_mfma_16x16x32_generic(av, bv, c);
}

/* ---- host runtime shim: synchronous, "device" memory is host memory ---- */
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    hipemu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
/* streams / events: everything runs synchronously, so forks and joins are no-ops */
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }

#endif /* HIPEMU_H */
