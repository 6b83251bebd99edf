// fd_kernels_h16.h -- pointwise GEMM for 16-bit storage (fp16 or bf16 activations / weights, fp32 accumulate) on the
// gfx950 2xK matrix instructions v_mfma_f32_32x32x16_{f16,bf16}.
//
// Same skeleton as fd_pw_gemm_f32 (LDS-DMA 3-stage ring issued two K tiles ahead, counted vmcnt + raw s_barrier, 128-byte
// LDS rows with the XOR swizzle on the DMA source and on the fragment reads, XCD-aware 1-D grid, bias-initialised
// accumulators).  A 128-byte row now holds 64 elements, so BK = 64, and a lane's 16-byte chunk IS the MFMA operand
// (8 consecutive k of its row: lanes 0-31 chunk 2s, lanes 32-63 chunk 2s+1 of MFMA step s) -- one ds_read_b128 per operand
// per instruction, no repacking.  In 16-bit the 1x1 layers are HBM-bound: the tile is 64 x 64 so that an
// A panel is streamed once per N tile out of the XCD's L2, and the epilogue transposes the tile through LDS so that the
// NHWC store is 16 bytes per lane.
#pragma once
#include "fd_device.h"

#ifndef FD_EMU
typedef __bf16 fd_bf16x8_hw __attribute__((ext_vector_type(8)));
#endif

__device__ __forceinline__ fd_f32x16 fd_mfma_32x32x16(fd_half, fd_u16x8 a, fd_u16x8 b, fd_f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fd_f16x8, a), __builtin_bit_cast(fd_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ fd_f32x16 fd_mfma_32x32x16(fd_bf16, fd_u16x8 a, fd_u16x8 b, fd_f32x16 c)
{
#ifdef FD_EMU
    return hipemu_mfma_f32_32x32x16_bf16(a, b, c);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fd_bf16x8_hw, a), __builtin_bit_cast(fd_bf16x8_hw, b), c, 0, 0, 0);
#endif
}

// 8 elements of a 16-bit tensor <-> fp32
__device__ __forceinline__ void fd_unpack8(fd_bf16, fd_u16x8 r, float (&f)[8])
{
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fd_bf16_to_f32(r[j]);
}
__device__ __forceinline__ void fd_unpack8(fd_half, fd_u16x8 r, float (&f)[8])
{
    const fd_f16x8 h = __builtin_bit_cast(fd_f16x8, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)h[j];
}
__device__ __forceinline__ fd_u16x8 fd_pack8(fd_bf16, const float (&f)[8])
{
    typedef unsigned fd_u32x4 __attribute__((ext_vector_type(4)));
    const fd_u32x4 r = {fd_f32x2_to_bf16x2(f[0], f[1]), fd_f32x2_to_bf16x2(f[2], f[3]), fd_f32x2_to_bf16x2(f[4], f[5]), fd_f32x2_to_bf16x2(f[6], f[7])};
    return __builtin_bit_cast(fd_u16x8, r);
}
__device__ __forceinline__ fd_u16x8 fd_pack8(fd_half, const float (&f)[8])
{
    fd_f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)f[j];
    return __builtin_bit_cast(fd_u16x8, h);
}
__device__ __forceinline__ fd_u16x8 fd_ld8(const void *p) { return *reinterpret_cast<const fd_u16x8 *>(p); }
__device__ __forceinline__ void fd_st8(void *p, fd_u16x8 v) { *reinterpret_cast<fd_u16x8 *>(p) = v; }

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3 for 16-bit storage, register-window form with EIGHT channels per work-item (fd_dw3_rows loads 4 channels = 8 bytes per lane in
// 16 bit: at the same instruction count it moves half the bytes of the fp32 instance -- 3.5 TB/s against 5.8).  q = x * (C/8) + c8; 16-byte
// loads and stores; the 3 x 3 window, the taps and the accumulators are two 4-channel vectors each.  C % 8 == 0.
// ------------------------------------------------------------------------------------------------
template <typename T, int S, int ACT>
__global__ void __launch_bounds__(256)
fd_dw3_rows8(const T *__restrict__ in, const float *__restrict__ wp, const float *__restrict__ bias,
             T *__restrict__ out, int H, int W, int Ho, int Wo, int C, int TH)
{
    const int CG = C >> 3;
    const fd_blk3 blk = fd_xcd_image_map();
    const int q = blk.x * 256 + threadIdx.x;
    if (q >= Wo * CG) return;
    const int xo = q / CG, c8 = q - xo * CG;
    const int n = blk.z;
    const int oy0 = blk.y * TH;
    const int oy1 = (oy0 + TH < Ho) ? oy0 + TH : Ho;
    fd_f32x4 w[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) { w[t][0] = fd_ld4(wp + (long)t * C + c8 * 8); w[t][1] = fd_ld4(wp + (long)t * C + c8 * 8 + 4); }
    const fd_f32x4 b0 = fd_ld4(bias + c8 * 8), b1 = fd_ld4(bias + c8 * 8 + 4);
    const T *img = in + (long)n * H * W * C + c8 * 8;
    const int x0 = xo * S - 1;
    const bool okl = x0 >= 0, okr = (x0 + 2) < W;
    const int xl = okl ? x0 : x0 + 1, xr = okr ? x0 + 2 : x0 + 1;       // clamped: the three loads are always issued, the padding is a select
    struct v8 { fd_f32x4 lo, hi; };
    auto cvt = [&](fd_u16x8 r, bool ok) {
        float f[8];
        fd_unpack8(T{}, r, f);
        v8 v;
        v.lo = fd_f32x4{f[0], f[1], f[2], f[3]}; v.hi = fd_f32x4{f[4], f[5], f[6], f[7]};
        if (!ok) { v.lo = fd_zero4(); v.hi = fd_zero4(); }
        return v;
    };
    auto load_row = [&](int iy, v8 &l, v8 &c, v8 &r) {
        const bool oky = iy >= 0 && iy < H;
        const int qy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        const T *p = img + (long)qy * W * C;
        const fd_u16x8 vl = fd_ld8(p + (long)xl * C), vc = fd_ld8(p + (long)(x0 + 1) * C), vr = fd_ld8(p + (long)xr * C);
        l = cvt(vl, oky && okl); c = cvt(vc, oky); r = cvt(vr, oky && okr);
    };
    T *o = out + (((long)n * Ho + oy0) * Wo) * C + (long)q * 8;
    v8 r0l, r0c, r0r, r1l, r1c, r1r, r2l, r2c, r2r;
    load_row(S * oy0 - 1, r0l, r0c, r0r);
    if (S == 1) load_row(oy0, r1l, r1c, r1r);
    for (int oy = oy0; oy < oy1; ++oy) {
        if (S == 1) load_row(oy + 1, r2l, r2c, r2r);
        else { load_row(2 * oy, r1l, r1c, r1r); load_row(2 * oy + 1, r2l, r2c, r2r); }
        fd_f32x4 a0 = b0, a1 = b1;
        a0 += r0l.lo * w[0][0]; a0 += r0c.lo * w[1][0]; a0 += r0r.lo * w[2][0];
        a0 += r1l.lo * w[3][0]; a0 += r1c.lo * w[4][0]; a0 += r1r.lo * w[5][0];
        a0 += r2l.lo * w[6][0]; a0 += r2c.lo * w[7][0]; a0 += r2r.lo * w[8][0];
        a1 += r0l.hi * w[0][1]; a1 += r0c.hi * w[1][1]; a1 += r0r.hi * w[2][1];
        a1 += r1l.hi * w[3][1]; a1 += r1c.hi * w[4][1]; a1 += r1r.hi * w[5][1];
        a1 += r2l.hi * w[6][1]; a1 += r2c.hi * w[7][1]; a1 += r2r.hi * w[8][1];
        a0 = fd_act4<ACT>(a0); a1 = fd_act4<ACT>(a1);
        const float f[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        fd_st8(o, fd_pack8(T{}, f));
        o += (long)Wo * C;
        if (S == 1) { r0l = r1l; r0c = r1c; r0r = r1r; r1l = r2l; r1c = r2c; r1r = r2r; }
        else { r0l = r2l; r0c = r2c; r0r = r2r; }
    }
}

// depth of the LDS-DMA ring of the 16-bit GEMMs (first-generation inference kernel, train forward, train backward-data)
#ifndef FD_H16_STAGES
#define FD_H16_STAGES 3
#endif
// the network head (Cout -> 1 pointwise + activation, nearest x2) evaluated on this GEMM's output tile instead of a launch of its own
struct fd_pw_head { const float *w, *b; float *y; int act, up, h, w_; };
template <typename T, int ACT, int HEAD>
__device__ __forceinline__ void
fd_pw_gemm_h16_body(const T *__restrict__ A, const T *__restrict__ Wt, const float *__restrict__ bias, T *__restrict__ out,
               int M, int N, int K, int K64, int m_tiles, int n_tiles, const fd_pw_head hd)
{
    constexpr int BM = 64, BN = 64, BK = 64;                // BK elements = 128 bytes per LDS row
    constexpr int ROWS = BM + BN, STAGE = ROWS * 128;        // bytes per stage
    constexpr int RG = ROWS / 8 / 4;
    FD_DYN_SMEM(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % n_tiles, mt = (slot / n_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const T *src[RG];
    int src_k[RG];
    bool src_is_a[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        const int r = (wave + 4 * i) * 8 + (lane >> 3);
        src_k[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;        // element offset of the chunk that lands in LDS slot (r, lane&7)
        src_is_a[i] = r < BM;
        if (r < BM) { long row = m0 + r; if (row > M - 1) row = M - 1; src[i] = A + row * K; }
        else { int row = n0 + (r - BM); if (row > N - 1) row = N - 1; src[i] = Wt + (long)row * K64; }
    }
    auto issue = [&](int t) {
        unsigned char *dst = smem + (t % FD_H16_STAGES) * STAGE + wave * 8 * 128;
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            int k = t * BK + src_k[i];
            if (src_is_a[i] && k >= K) k = 0;                // ragged K: finite data; the zero-padded weights annihilate it
            fd_glds16(reinterpret_cast<const float *>(src[i] + k), reinterpret_cast<float *>(dst + i * 4 * 8 * 128));
        }
    };
    fd_f32x16 acc;
    {
        const int col = n0 + wn * 32 + (lane & 31);
        const float bv = col < N ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bv;
    }
    const int h = lane >> 5;
    int a_off[4], b_off[4];                                   // byte offsets within a stage
    {
        const int ra = wm * 32 + (lane & 31), rb = BM + wn * 32 + (lane & 31);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a_off[s] = ra * 128 + (((2 * s + h) ^ ((ra >> 1) & 7)) << 4);
            b_off[s] = rb * 128 + (((2 * s + h) ^ ((rb >> 1) & 7)) << 4);
        }
    }
    const int Tn = K64 / BK;
    issue(0);
    if (FD_H16_STAGES > 2 && Tn > 1) issue(1);
    for (int t = 0; t < Tn; ++t) {
        if (FD_H16_STAGES > 2 && t + 1 < Tn) fd_wait_vmcnt<RG>(); else fd_wait_vmcnt<0>();
        fd_block_barrier();
        if (t + FD_H16_STAGES - 1 < Tn) issue(t + FD_H16_STAGES - 1);
        const unsigned char *cur = smem + (t % FD_H16_STAGES) * STAGE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const fd_u16x8 a = *reinterpret_cast<const fd_u16x8 *>(cur + a_off[s]);
            const fd_u16x8 b = *reinterpret_cast<const fd_u16x8 *>(cur + b_off[s]);
            acc = fd_mfma_32x32x16(T{}, a, b, acc);
        }
    }
    // epilogue: activation, conversion, transpose through LDS (each wave owns a [32][32+8] T tile), 16-byte stores
    __syncthreads();                                          // every wave is done reading the operand stages
    T *tile = reinterpret_cast<T *>(smem) + wave * 32 * 40;
    const int col = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        fd_st1(tile + row * 40 + col, fd_act<ACT>(acc[r]));
    }
    __syncthreads();
    if (HEAD) {
        // N <= 32: the wn == 0 waves' tiles hold every channel of their 32 pixels (rounded to T like the stored tensor the separate head kernel
        // would read); lane = pixel: y = act_h(sum_c tile[c] * w_h[c] + b_h), written as a 2x2 block of the full-resolution output (or 1:1)
        if (wn == 0 && lane < 32) {
            const long p = m0 + wm * 32 + lane;
            if (p < M) {
                float acc1 = hd.b[0];
                for (int c = 0; c < N; ++c) acc1 = fmaf(fd_ld1(tile + lane * 40 + c), hd.w[c], acc1);
                const float v = hd.act == 2 ? fminf(fmaxf(acc1, 0.0f), 6.0f) : (hd.act == 1 ? fmaxf(acc1, 0.0f) : acc1);
                if (!hd.up) hd.y[p] = v;
                else {
                    const int ox = (int)(p % hd.w_);
                    const long t = p / hd.w_;
                    const int oy = (int)(t % hd.h);
                    const long n = t / hd.h;
                    float *o = hd.y + ((n * 2 * hd.h + 2 * oy) * 2 * (long)hd.w_ + 2 * ox);
                    o[0] = v; o[1] = v; o[2 * hd.w_] = v; o[2 * hd.w_ + 1] = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = lane + 64 * i;                         // 128 chunks of 8 elements: row = id / 4, chunk = id % 4
        const int row = id >> 2, c8 = (id & 3) * 8;
        const long grow = m0 + wm * 32 + row;
        const int gcol = n0 + wn * 32 + c8;
        if (grow < M && gcol < N) {
            if (gcol + 8 <= N) {
                *reinterpret_cast<fd_u16x8 *>(out + grow * N + gcol) = *reinterpret_cast<const fd_u16x8 *>(tile + row * 40 + c8);
            } else {
                for (int j = 0; j < 8 && gcol + j < N; ++j) out[grow * N + gcol + j] = tile[row * 40 + c8 + j];
            }
        }
    }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256)
fd_pw_gemm_h16(const T *__restrict__ A, const T *__restrict__ Wt, const float *__restrict__ bias, T *__restrict__ out,
               int M, int N, int K, int K64, int m_tiles, int n_tiles)
{
    fd_pw_gemm_h16_body<T, ACT, 0>(A, Wt, bias, out, M, N, K, K64, m_tiles, n_tiles, fd_pw_head{});
}
// the same GEMM with the network head on its output tile (N <= 32): the pointwise output itself is not written
template <typename T, int ACT>
__global__ void __launch_bounds__(256)
fd_pw_gemm_head_h16(const T *__restrict__ A, const T *__restrict__ Wt, const float *__restrict__ bias,
                    int M, int N, int K, int K64, int m_tiles, int n_tiles, const fd_pw_head hd)
{
    fd_pw_gemm_h16_body<T, ACT, 1>(A, Wt, bias, (T *)nullptr, M, N, K, K64, m_tiles, n_tiles, hd);
}
