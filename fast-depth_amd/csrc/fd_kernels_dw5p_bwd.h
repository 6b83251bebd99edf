// fd_kernels_dw5p_bwd.h -- backward of the depthwise 5x5 units on  up2(a_low) + a_skip  (decode_conv3 / 4 / 5 .0; autograd of reference models.py:61-68
// behind models.py:723-729) for the bf16 train plans: the row-walking pixel-pair form of fd_kernels_dw5p.h (round 6).
//
// Replaces the paired launch fd_dw_bwd<T, 5, 1, 2, ...> on these units (profiles/r05: 150 / 82 / 45 us per step, 143 VALU lane-instructions per output
// against 25 packed FMAs: LDS-tiled 8 x 16 tiles that stage 1.9 x their outputs, fp32 patches converted on every read).  One launch, two roles dealt by
// workgroup number; in both a WAVE walks down a band of rows for 2 adjacent channels x 4 adjacent columns per lane, everything it re-uses in registers:
//   role D (backward-data):  dz = A((G - c1) - (z - mu) c2) is formed once per loaded element, rounded to the storage type and kept as pixel pairs in a
//     5-row window; d_in = the correlation of that window with the FLIPPED 16-bit tap pairs on v_dot2 (15 per value, fp32 accumulation); the skip
//     gradient = d_in leaves at full resolution (the gradient of the ACTIVATED skip tensor), the producer's gradient = mask(y_low) * (2 x 2 sum of d_in) at low resolution together
//     with its BatchNorm-backward sums (sum G, sum G * xhat), which reach the producer's statistics rows once per workgroup;
//   role W (backward-weights): the forward input  relu(z_low s1 + t1) ^2 + relu6(z_skip s2 + t2)  is re-created on load, rounded, and kept as pixel pairs
//     in a 5-row window together with the pairs shifted by one pixel (v_perm of neighbouring pairs, formed once per row); dW[ky][kx] accumulates
//     dot2(dz pair, input pair) -- 2 per tap, row and lane, 12.5 per value -- in 50 registers for the whole band; a workgroup's four waves meet in LDS
//     at the end and write ONE partial row (fd_reduce_weights_batch_f32 adds the rows).
// Memory access as in fd_dw5_rows: raw-buffer loads / stores, lane offsets fixed for the band, the horizontal zero padding by the range check (plus a
// select where the padded quantity is not 0 at zero input: dz).  No LDS and no barrier until the end-of-kernel reductions.
// Numerics (layer-local test, tests/harness.py): dz and the re-created input are rounded to the storage type (as the 8-channel LDS form did); role D
// rounds the taps to the storage type as well (fd_train_plan_lds_rounding bit 3); sums and gradients accumulate in fp32.
#pragma once
#include "fd_kernels_dw5p.h"
#include "fd_kernels_bwd.h"

template <typename T> struct fd_dw5_bwd_args {
    const T *G, *Z, *Zin, *Zskip;      // this unit's dL/dy (after its consumer's mask) and raw output; the raw outputs of the low-resolution producer and of the skip source
    T *Gin, *SGout;                    // gradient handed to the producer (low resolution) / to the skip source (full resolution)
    const float *coef, *w, *st_in, *st_skip;   // BN-backward coefficients [4][C] of this unit, live taps [C][25], tables [4][C] of the producer / the skip source
    fd_stat_rows sr;                   // BatchNorm-backward statistics rows of the PRODUCER
    float *wpart;                      // weight-gradient partial rows: row = image * wgs_w + workgroup, [25][C]
    int H, W, C, groups_x;             // full-resolution map, channels, strip pairs per row (W / 8, rounded up)
    int bh_d, bh_w, wgs_d, wgs_w;      // rows per band and workgroups per image and channel block of the two roles
    fd_bn_bwd_fin fin;                 // rows != null: this unit's BatchNorm backward is finalised in the kernel's prologue (few statistics rows: fd_bstat_table_block)
};

#ifndef FD_DW5B_WAVES
#define FD_DW5B_WAVES 2               // waves per SIMD the register allocation is held to (<= 256 VGPRs: both roles keep ~200 live)
#endif
#ifdef FD_EMU
#define FD_DW5B_ATTR
#else
#define FD_DW5B_ATTR __attribute__((amdgpu_waves_per_eu(FD_DW5B_WAVES, FD_DW5B_WAVES)))
#endif

// a lane's BatchNorm-backward coefficients of channel c + ch: from the table a finalisation launch wrote, or (fin.rows != null) from the block the kernel's own
// prologue derived into LDS (s_cf: [4][CBF] for the workgroup's channel block, local channel 2 l + ch)
#define FD_DW_ROWS_COEF(ch)                                                                                                                         \
    do {                                                                                                                                            \
        if (a.fin.rows) { cA[ch] = s_cf[FD_CF_A * CBF + 2 * l + ch]; c1[ch] = s_cf[FD_CF_C1 * CBF + 2 * l + ch]; cM[ch] = s_cf[FD_CF_MU * CBF + 2 * l + ch]; c2[ch] = s_cf[FD_CF_C2 * CBF + 2 * l + ch]; } \
        else { cA[ch] = a.coef[FD_CF_A * C + c + ch]; c1[ch] = a.coef[FD_CF_C1 * C + c + ch]; cM[ch] = a.coef[FD_CF_MU * C + c + ch]; c2[ch] = a.coef[FD_CF_C2 * C + c + ch]; }                     \
    } while (0)

// 16-bit storage -> fp32 of the two channels of a loaded word
__device__ __forceinline__ float fd_w16_lo(fd_bf16, unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float fd_w16_hi(fd_bf16, unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ __forceinline__ float fd_w16_lo(fd_half, unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)v); }
__device__ __forceinline__ float fd_w16_hi(fd_half, unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); }

// ---- role D: backward-data ------------------------------------------------------------------------------------------------------------------------
template <typename T, int ACT1, int ACT2>
__device__ __forceinline__ void
fd_dw5_dgrad_rows_body(const fd_dw5_bwd_args<T> &a, float *red, const float *s_cf, const int wg, const int c0, const int n, const long stat_blk)
{
    constexpr int CBF = 64;
    const int H = a.H, W = a.W, C = a.C, Hs = H >> 1, Ws = W >> 1;
    const int wave = FD_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int item = wg * 4 + wave;
    const int band = item / a.groups_x, sg = item - band * a.groups_x;
    const int cend = c0 + 64 < C ? c0 + 64 : C;
    const int y0 = band * a.bh_d, y1 = y0 + a.bh_d < H ? y0 + a.bh_d : H;
    const int l = lane & 31, xs = 4 * (2 * sg + (lane >> 5)), c = c0 + 2 * l;
    const bool live = y0 < H && c < cend && xs < W;
    float sg0 = 0.f, sg1 = 0.f, sx0 = 0.f, sx1 = 0.f;       // this lane's sums of the producer's gradient and gradient * xhat, channels c / c + 1
    if (live) {
        // flipped taps as 16-bit pairs: d_in[x] = sum_k dz[x + 2 - k] w[k] = sum_k' dz[x - 2 + k'] w[4 - k']
        unsigned w[5][6][2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float *wc = a.w + (long)(c + ch) * 25;
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const float f0 = wc[(4 - ky) * 5 + 4], f1 = wc[(4 - ky) * 5 + 3], f2 = wc[(4 - ky) * 5 + 2], f3 = wc[(4 - ky) * 5 + 1], f4 = wc[(4 - ky) * 5 + 0];
                w[ky][0][ch] = fd_pack2(T{}, f0, f1); w[ky][1][ch] = fd_pack2(T{}, f2, f3); w[ky][2][ch] = fd_pack2(T{}, f4, 0.f);
                w[ky][3][ch] = fd_pack2(T{}, 0.f, f0); w[ky][4][ch] = fd_pack2(T{}, f1, f2); w[ky][5][ch] = fd_pack2(T{}, f3, f4);
            }
        }
        float cA[2], c1[2], cM[2], c2[2], s1[2], t1[2], m1[2], i1[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            FD_DW_ROWS_COEF(ch);
            s1[ch] = a.st_in[FD_ST_SCALE * C + c + ch]; t1[ch] = a.st_in[FD_ST_SHIFT * C + c + ch]; m1[ch] = a.st_in[FD_ST_MEAN * C + c + ch]; i1[ch] = a.st_in[FD_ST_INVSTD * C + c + ch];
        }
        unsigned so[4];                                      // byte offsets of the strip's four pixel pairs in a full-resolution row (out of range: outside the image)
        bool pin[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int x = xs - 2 + 2 * p;
            pin[p] = x >= 0 && x < W;
            so[p] = pin[p] ? (fd_mul24((unsigned)x, (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
        }
        const unsigned lo0 = (fd_mul24((unsigned)(xs >> 1), (unsigned)C) + (unsigned)c) * 2u;      // the strip's two low-resolution pixels
        const unsigned rowb = fd_mul24((unsigned)W, (unsigned)C) * 2u, pxb = (unsigned)C * 2u, rowl = rowb >> 1;
        const fd_bufrsrc r_g = fd_make_rsrc(a.G + (long)n * H * W * C, (unsigned)H * rowb), r_z = fd_make_rsrc(a.Z + (long)n * H * W * C, (unsigned)H * rowb);
        const fd_bufrsrc r_lo = fd_make_rsrc(a.Zin + (long)n * Hs * Ws * C, (unsigned)Hs * rowl);
        const fd_bufrsrc r_so = fd_make_rsrc(a.SGout + (long)n * H * W * C, (unsigned)H * rowb), r_gl = fd_make_rsrc(a.Gin + (long)n * Hs * Ws * C, (unsigned)Hs * rowl);

        unsigned ng[2][8], nz[2][8], nlo[2];                 // in flight: G / z of the next step's two dz rows, z_low under the next step's two OUTPUT rows
        bool nv = false;
        // step `it`: dz rows r = y0 - 2 + 2 it and r + 1 enter the window, d_in rows r - 2 and r - 1 (it >= 2) leave
        auto issue = [&](int it) FD_INLINE_LAMBDA {
            const int r = y0 - 2 + 2 * it;
            nv = r >= 0 && r < H;
            if (nv) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const unsigned ro = (unsigned)(r + rr) * rowb;
                        ng[rr][2 * p] = fd_buf_ld32(r_g, so[p], ro); ng[rr][2 * p + 1] = fd_buf_ld32(r_g, so[p], ro + pxb);
                        nz[rr][2 * p] = fd_buf_ld32(r_z, so[p], ro); nz[rr][2 * p + 1] = fd_buf_ld32(r_z, so[p], ro + pxb);
                    }
            }
            if (it >= 2) {
                const int y = r - 2;                         // (even; rows y, y + 1 are inside the band)
                nlo[0] = fd_buf_ld32(r_lo, lo0, (unsigned)(y >> 1) * rowl); nlo[1] = fd_buf_ld32(r_lo, lo0, (unsigned)(y >> 1) * rowl + pxb);
            }
        };
        unsigned win[5][4][2];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p) { win[q][p][0] = 0u; win[q][p][1] = 0u; }
        auto convert = [&](unsigned (&dst)[4][2], const unsigned (&g)[8], const unsigned (&z)[8], bool rv) FD_INLINE_LAMBDA {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float d00 = fd_dz(fd_w16_lo(T{}, g[2 * p]), fd_w16_lo(T{}, z[2 * p]), cA[0], c1[0], cM[0], c2[0]);
                const float d01 = fd_dz(fd_w16_lo(T{}, g[2 * p + 1]), fd_w16_lo(T{}, z[2 * p + 1]), cA[0], c1[0], cM[0], c2[0]);
                const float d10 = fd_dz(fd_w16_hi(T{}, g[2 * p]), fd_w16_hi(T{}, z[2 * p]), cA[1], c1[1], cM[1], c2[1]);
                const float d11 = fd_dz(fd_w16_hi(T{}, g[2 * p + 1]), fd_w16_hi(T{}, z[2 * p + 1]), cA[1], c1[1], cM[1], c2[1]);
                const bool v = rv && pin[p];                 // (dz of a pixel outside the image is 0, not dz(0, 0))
                dst[p][0] = v ? fd_pack2(T{}, d00, d01) : 0u; dst[p][1] = v ? fd_pack2(T{}, d10, d11) : 0u;
            }
        };
        float lacc[2][2];                                    // 2 x 2 sums of d_in: [low pixel of the strip][channel]
        unsigned lo_now[2];                                  // z_low under this step's two output rows (the registers it landed in carry the next step's already)
        auto out_row = [&](auto SB, int y, int rr) FD_INLINE_LAMBDA {
            constexpr int sb = decltype(SB)::value;
            float acc[4][2];
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const unsigned (&R)[4][2] = win[(sb + ky) % 5];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        if (ky == 0 && k == 0) {
                            acc[0][ch] = fd_dot2_first(T{}, R[k][ch], w[ky][k][ch], 0.0f);
                            acc[1][ch] = fd_dot2_first(T{}, R[k][ch], w[ky][3 + k][ch], 0.0f);
                            acc[2][ch] = fd_dot2_first(T{}, R[k + 1][ch], w[ky][k][ch], 0.0f);
                            acc[3][ch] = fd_dot2_first(T{}, R[k + 1][ch], w[ky][3 + k][ch], 0.0f);
                        } else {
                            fd_dot2_acc(T{}, R[k][ch], w[ky][k][ch], acc[0][ch]);
                            fd_dot2_acc(T{}, R[k][ch], w[ky][3 + k][ch], acc[1][ch]);
                            fd_dot2_acc(T{}, R[k + 1][ch], w[ky][k][ch], acc[2][ch]);
                            fd_dot2_acc(T{}, R[k + 1][ch], w[ky][3 + k][ch], acc[3][ch]);
                        }
                    }
                }
            }
            fd_dot2_done(acc);
            // the gradient of the ACTIVATED skip tensor at full resolution (its source's activation mask is applied by the source's own consumer, which
            // adds this buffer: ADD_SG); the producer's gradient collects its 2 x 2 block
#pragma unroll
            for (int j = 0; j < 4; ++j) fd_buf_st32(r_so, so[1 + (j >> 1)], (unsigned)y * rowb + (unsigned)(j & 1) * pxb, fd_pack2(T{}, acc[j][0], acc[j][1]));
#pragma unroll
            for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const float s = acc[2 * jl][ch] + acc[2 * jl + 1][ch];
                    lacc[jl][ch] = rr == 0 ? s : lacc[jl][ch] + s;
                }
            if (rr == 1) {
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) {
                    const float z0 = fd_w16_lo(T{}, lo_now[jl]), z1 = fd_w16_hi(T{}, lo_now[jl]);
                    const unsigned packed = fd_pack2(T{}, lacc[jl][0] * fd_actmask<ACT1>(fmaf(z0, s1[0], t1[0])), lacc[jl][1] * fd_actmask<ACT1>(fmaf(z1, s1[1], t1[1])));
                    fd_buf_st32(r_gl, lo0, (unsigned)(y >> 1) * rowl + (unsigned)jl * pxb, packed);
                    const float g0 = fd_w16_lo(T{}, packed), g1 = fd_w16_hi(T{}, packed);      // statistics of the stored (rounded) gradient
                    sg0 += g0; sg1 += g1;
                    sx0 = fmaf(g0, (z0 - m1[0]) * i1[0], sx0); sx1 = fmaf(g1, (z1 - m1[1]) * i1[1], sx1);
                }
            }
        };
        const int n_it = (y1 - y0 + 4) >> 1;
        issue(0);
        auto step = [&](auto PH, int it) FD_INLINE_LAMBDA {
            constexpr int ph = decltype(PH)::value;
            const bool rv = nv;
            unsigned hold[4][2];                             // row r + 1 takes the slot of row r - 4, which output row r - 2 still reads: its pairs wait here
            convert(win[(2 * ph) % 5], ng[0], nz[0], rv);
            convert(hold, ng[1], nz[1], rv);
            lo_now[0] = nlo[0]; lo_now[1] = nlo[1];
            if (it + 1 < n_it) issue(it + 1);                // everything the next step reads flies under this step's 240 dot2
            if (it >= 2) out_row(fd_int<(2 * ph + 1) % 5>{}, y0 - 4 + 2 * it, 0);
#pragma unroll
            for (int p = 0; p < 4; ++p) { win[(2 * ph + 1) % 5][p][0] = hold[p][0]; win[(2 * ph + 1) % 5][p][1] = hold[p][1]; }
            if (it >= 2) out_row(fd_int<(2 * ph + 2) % 5>{}, y0 - 3 + 2 * it, 1);
        };
        for (int it0 = 0; it0 < n_it; it0 += 5) {
            step(fd_int<0>{}, it0);
            if (it0 + 1 < n_it) step(fd_int<1>{}, it0 + 1);
            if (it0 + 2 < n_it) step(fd_int<2>{}, it0 + 2);
            if (it0 + 3 < n_it) step(fd_int<3>{}, it0 + 3);
            if (it0 + 4 < n_it) step(fd_int<4>{}, it0 + 4);
        }
    }
    // ---- the workgroup's BatchNorm-backward sums: the two strips of a wave (lanes l, l + 32), then the four waves through LDS; consecutive lanes add
    // consecutive channels to the producer's statistics rows
    sg0 += __shfl_xor(sg0, 32); sg1 += __shfl_xor(sg1, 32); sx0 += __shfl_xor(sx0, 32); sx1 += __shfl_xor(sx1, 32);
    if (lane < 32) {
        red[(wave * 2 + 0) * 64 + 2 * l] = sg0; red[(wave * 2 + 0) * 64 + 2 * l + 1] = sg1;
        red[(wave * 2 + 1) * 64 + 2 * l] = sx0; red[(wave * 2 + 1) * 64 + 2 * l + 1] = sx1;
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        if (c0 + ch < cend) {
            const float v = (red[(0 * 2 + which) * 64 + ch] + red[(1 * 2 + which) * 64 + ch]) + (red[(2 * 2 + which) * 64 + ch] + red[(3 * 2 + which) * 64 + ch]);
            fd_stat_add<FD_STAT_BWD>(a.sr, stat_blk, C, which, c0 + ch, v);
        }
    }
}

// ---- role W: backward-weights ---------------------------------------------------------------------------------------------------------------------
#define FD_PERM_SHIFT1 0x05040302u    /* fd_perm(next, cur, FD_PERM_SHIFT1) = (high half of cur, low half of next): the pixel pair one pixel to the right */
template <typename T, int ACT1, int ACT2>
__device__ __forceinline__ void
fd_dw5_wgrad_rows_body(const fd_dw5_bwd_args<T> &a, float *red, const float *s_cf, const int wg, const int c0, const int n, const long row_blk)
{
    constexpr int CBF = 64;
    const int H = a.H, W = a.W, C = a.C, Hs = H >> 1, Ws = W >> 1;
    const int wave = FD_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int item = wg * 4 + wave;
    const int band = item / a.groups_x, sg = item - band * a.groups_x;
    const int cend = c0 + 64 < C ? c0 + 64 : C;
    const int y0 = band * a.bh_w, y1 = y0 + a.bh_w < H ? y0 + a.bh_w : H;
    const int l = lane & 31, xs = 4 * (2 * sg + (lane >> 5)), c = c0 + 2 * l;
    const bool live = y0 < H && c < cend && xs < W;
    float acc[25][2];
#pragma unroll
    for (int t = 0; t < 25; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; }
    if (live) {
        float cA[2], c1[2], cM[2], c2[2], s1[2], t1[2], s2[2], t2[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            FD_DW_ROWS_COEF(ch);
            s1[ch] = a.st_in[FD_ST_SCALE * C + c + ch]; t1[ch] = a.st_in[FD_ST_SHIFT * C + c + ch];
            s2[ch] = a.st_skip[FD_ST_SCALE * C + c + ch]; t2[ch] = a.st_skip[FD_ST_SHIFT * C + c + ch];
        }
        unsigned so[4], lo_[4];
        bool pin[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int x = xs - 2 + 2 * p;
            pin[p] = x >= 0 && x < W;
            so[p] = pin[p] ? (fd_mul24((unsigned)x, (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
            lo_[p] = pin[p] ? (fd_mul24((unsigned)(x >> 1), (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
        }
        const unsigned rowb = fd_mul24((unsigned)W, (unsigned)C) * 2u, pxb = (unsigned)C * 2u, rowl = rowb >> 1;
        const fd_bufrsrc r_g = fd_make_rsrc(a.G + (long)n * H * W * C, (unsigned)H * rowb), r_z = fd_make_rsrc(a.Z + (long)n * H * W * C, (unsigned)H * rowb);
        const fd_bufrsrc r_sk = fd_make_rsrc(a.Zskip + (long)n * H * W * C, (unsigned)H * rowb), r_lo = fd_make_rsrc(a.Zin + (long)n * Hs * Ws * C, (unsigned)Hs * rowl);

        unsigned nsk[2][8], nlo[4], ng[2][4], nz[2][4];      // in flight: the next step's two input rows (z_skip, 8 pixels; their parents) and G / z of its two dz rows (4 pixels)
        bool nv = false;
        // step `it`: input rows r = y0 - 2 + 2 it and r + 1 enter the window; dz rows r - 2 and r - 1 (it >= 2) are multiplied with the window
        auto issue = [&](int it) FD_INLINE_LAMBDA {
            const int r = y0 - 2 + 2 * it;
            nv = r >= 0 && r < H;
            if (nv) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const unsigned ro = (unsigned)(r + rr) * rowb;
                        nsk[rr][2 * p] = fd_buf_ld32(r_sk, so[p], ro); nsk[rr][2 * p + 1] = fd_buf_ld32(r_sk, so[p], ro + pxb);
                    }
#pragma unroll
                for (int p = 0; p < 4; ++p) nlo[p] = fd_buf_ld32(r_lo, lo_[p], (unsigned)(r >> 1) * rowl);
            }
            if (it >= 2) {
                const int y = r - 2;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned ro = (unsigned)(y + rr) * rowb + (unsigned)(j & 1) * pxb;
                        ng[rr][j] = fd_buf_ld32(r_g, so[1 + (j >> 1)], ro); nz[rr][j] = fd_buf_ld32(r_z, so[1 + (j >> 1)], ro);
                    }
            }
        };
        unsigned win[5][7][2];                               // input row (y0 - 2 + r) in win[r % 5]: pairs 0..3 as loaded, 4..6 = the pairs shifted right by one pixel
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int p = 0; p < 7; ++p) { win[q][p][0] = 0u; win[q][p][1] = 0u; }
        auto convert = [&](unsigned (&dst)[7][2], const unsigned (&sk)[8], bool rv) FD_INLINE_LAMBDA {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float l0 = fd_act<ACT1>(fmaf(fd_w16_lo(T{}, nlo[p]), s1[0], t1[0])), l1 = fd_act<ACT1>(fmaf(fd_w16_hi(T{}, nlo[p]), s1[1], t1[1]));
                const float e0 = fd_act<ACT2>(fmaf(fd_w16_lo(T{}, sk[2 * p]), s2[0], t2[0])) + l0, o0 = fd_act<ACT2>(fmaf(fd_w16_lo(T{}, sk[2 * p + 1]), s2[0], t2[0])) + l0;
                const float e1 = fd_act<ACT2>(fmaf(fd_w16_hi(T{}, sk[2 * p]), s2[1], t2[1])) + l1, o1 = fd_act<ACT2>(fmaf(fd_w16_hi(T{}, sk[2 * p + 1]), s2[1], t2[1])) + l1;
                const bool v = rv && pin[p];                 // (zero padding of the ACTIVATED, summed input)
                dst[p][0] = v ? fd_pack2(T{}, e0, o0) : 0u; dst[p][1] = v ? fd_pack2(T{}, e1, o1) : 0u;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                dst[4 + p][0] = fd_perm(dst[p + 1][0], dst[p][0], FD_PERM_SHIFT1);
                dst[4 + p][1] = fd_perm(dst[p + 1][1], dst[p][1], FD_PERM_SHIFT1);
            }
        };
        // the dz pairs of the strip's 4 pixels of one row
        auto make_dz = [&](unsigned (&dz)[2][2], const unsigned (&g)[4], const unsigned (&z)[4]) FD_INLINE_LAMBDA {
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const float d00 = fd_dz(fd_w16_lo(T{}, g[2 * j2]), fd_w16_lo(T{}, z[2 * j2]), cA[0], c1[0], cM[0], c2[0]);
                const float d01 = fd_dz(fd_w16_lo(T{}, g[2 * j2 + 1]), fd_w16_lo(T{}, z[2 * j2 + 1]), cA[0], c1[0], cM[0], c2[0]);
                const float d10 = fd_dz(fd_w16_hi(T{}, g[2 * j2]), fd_w16_hi(T{}, z[2 * j2]), cA[1], c1[1], cM[1], c2[1]);
                const float d11 = fd_dz(fd_w16_hi(T{}, g[2 * j2 + 1]), fd_w16_hi(T{}, z[2 * j2 + 1]), cA[1], c1[1], cM[1], c2[1]);
                dz[j2][0] = fd_pack2(T{}, d00, d01); dz[j2][1] = fd_pack2(T{}, d10, d11);
            }
        };
        // dz row y (pairs of the strip's 4 pixels) against the window: SB = slot of input row y - 2
        auto taps = [&](auto SB, const unsigned (&dz)[2][2]) FD_INLINE_LAMBDA {
            constexpr int sb = decltype(SB)::value;
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const unsigned (&R)[7][2] = win[(sb + ky) % 5];
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    // input pixels (x + kx - 2, x + kx - 1) of dz pixels (x, x + 1): even kx -> the loaded pair kx / 2 (+ 1 for the strip's second dz pair),
                    // odd kx -> the shifted pair (kx - 1) / 2 (+ 1)
                    const int p0 = (kx & 1) ? 4 + (kx >> 1) : (kx >> 1);
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        fd_dot2_acc(T{}, dz[0][ch], R[p0][ch], acc[ky * 5 + kx][ch]);
                        fd_dot2_acc(T{}, dz[1][ch], R[p0 + 1][ch], acc[ky * 5 + kx][ch]);
                    }
                }
            }
        };
        const int n_it = (y1 - y0 + 4) >> 1;
        issue(0);
        auto step = [&](auto PH, int it) FD_INLINE_LAMBDA {
            constexpr int ph = decltype(PH)::value;
            const bool rv = nv;
            unsigned hold[7][2], dza[2][2], dzb[2][2];
            convert(win[(2 * ph) % 5], nsk[0], rv);
            convert(hold, nsk[1], rv);                       // (row r + 1 takes the slot of row r - 4, which dz row r - 2 still reads)
            if (it >= 2) { make_dz(dza, ng[0], nz[0]); make_dz(dzb, ng[1], nz[1]); }
            if (it + 1 < n_it) issue(it + 1);                // everything the next step reads flies under this step's 200 dot2
            if (it >= 2) taps(fd_int<(2 * ph + 1) % 5>{}, dza);      // dz row r - 2: input rows r - 4 .. r
#pragma unroll
            for (int p = 0; p < 7; ++p) { win[(2 * ph + 1) % 5][p][0] = hold[p][0]; win[(2 * ph + 1) % 5][p][1] = hold[p][1]; }
            if (it >= 2) taps(fd_int<(2 * ph + 2) % 5>{}, dzb);      // dz row r - 1: input rows r - 3 .. r + 1
        };
        for (int it0 = 0; it0 < n_it; it0 += 5) {
            step(fd_int<0>{}, it0);
            if (it0 + 1 < n_it) step(fd_int<1>{}, it0 + 1);
            if (it0 + 2 < n_it) step(fd_int<2>{}, it0 + 2);
            if (it0 + 3 < n_it) step(fd_int<3>{}, it0 + 3);
            if (it0 + 4 < n_it) step(fd_int<4>{}, it0 + 4);
        }
    }
    // ---- the workgroup's partial row: the two strips of a wave, then the four waves through LDS [wave][25][64 channels]
    fd_dot2_done(acc);
#pragma unroll
    for (int t = 0; t < 25; ++t) { acc[t][0] += __shfl_xor(acc[t][0], 32); acc[t][1] += __shfl_xor(acc[t][1], 32); }
    if (lane < 32) {
#pragma unroll
        for (int t = 0; t < 25; ++t) { red[(wave * 25 + t) * 64 + 2 * l] = acc[t][0]; red[(wave * 25 + t) * 64 + 2 * l + 1] = acc[t][1]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 25 * 64; i += 256) {
        const int t = i >> 6, ch = i & 63;
        if (c0 + ch < cend)
            a.wpart[(row_blk * 25 + t) * C + c0 + ch] = (red[(0 * 25 + t) * 64 + ch] + red[(1 * 25 + t) * 64 + ch]) + (red[(2 * 25 + t) * 64 + ch] + red[(3 * 25 + t) * 64 + ch]);
    }
}

// grid (wgs_d + wgs_w, channel blocks of 64, images) through fd_xcd_image_map; block 256; W % 4 == 0, H even, C % 8 == 0, image bytes < 2^31
template <typename T, int ACT1, int ACT2>
__global__ void __launch_bounds__(256) FD_DW5B_ATTR
fd_dw5_bwd_rows(const fd_dw5_bwd_args<T> a)
{
    __shared__ float red[4 * 25 * 64];
    __shared__ double sh[512];
    __shared__ float s_cf[4 * 64];
    const fd_blk3 blk = fd_xcd_image_map();
    const int c0 = blk.y * 64, n = blk.z;
    if (a.fin.rows) fd_bstat_table_block(a.fin, sh, s_cf, c0, 64, a.C, (int)threadIdx.x, blk.x == 0 && n == 0);
    if (blk.x < a.wgs_d) fd_dw5_dgrad_rows_body<T, ACT1, ACT2>(a, red, s_cf, blk.x, c0, n, (long)n * a.wgs_d + blk.x);
    else fd_dw5_wgrad_rows_body<T, ACT1, ACT2>(a, red, s_cf, blk.x - a.wgs_d, c0, n, (long)n * a.wgs_w + (blk.x - a.wgs_d));
}

// ---- train-mode FORWARD of the same units (replaces fd_dwconv_train<T, 5, 1, 2, ...>: 67 / 43 / 27 us per bf16 step) ---------------------------------
//   z_out = conv5x5( relu(z_low s1 + t1)^2 + relu6(z_skip s2 + t2) ),  raw (no BatchNorm folded: batch statistics), rounded to the storage type,
//   + the workgroup's per-channel sums of the ROUNDED values, added to the unit's statistics rows; fin.rows != null: the low-resolution producer's
//   BatchNorm is finalised here (fd_stat_table_block in the prologue, all four waves; workgroup (0, *, 0) is the writer).
// fd_dw5_rows with the two BatchNorm + activation transforms on load (once per loaded element; the pair is built by the rounding conversion itself) and
// the live fp32 taps rounded to 16-bit pairs in the prologue (fd_train_plan_lds_rounding bits 0 and 2).
// grid (wgs, channel blocks of 64, images) through fd_xcd_image_map; block 256 = 4 independent waves until the end-of-kernel statistics reduction.
template <typename T, int ACT1, int ACT2>
__global__ void __launch_bounds__(256) FD_DW5B_ATTR          // (2 waves per SIMD: at 3 the transforms' temporaries spill -- 168 VGPRs + 120 bytes of scratch ran 2.5x longer)
fd_dw5_rows_train(const T *__restrict__ zin, const float *__restrict__ st1, const T *__restrict__ zskip, const float *__restrict__ st2,
                  const float *__restrict__ wgt, T *__restrict__ zout, fd_stat_rows sr, int H, int W, int C, int groups_x, int bh, const fd_bn_fin fin)
{
    __shared__ double sh[512];
    __shared__ float s_st[2 * 64];
    __shared__ float red[4 * 2 * 64];
    const fd_blk3 blk = fd_xcd_image_map();
    const int c0 = blk.y * 64, cend = c0 + 64 < C ? c0 + 64 : C, n = blk.z;
    const int wave = FD_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int item = blk.x * 4 + wave;
    const int band = item / groups_x, sg = item - band * groups_x;
    const int y0 = band * bh, y1 = y0 + bh < H ? y0 + bh : H;
    const int l = lane & 31, xs = 4 * (2 * sg + (lane >> 5)), c = c0 + 2 * l;
    const bool live = y0 < H && c < cend && xs < W;
    const int Hs = H >> 1, Ws = W >> 1;
    if (fin.rows) fd_stat_table_block(fin, sh, s_st, c0, 64, C, (int)threadIdx.x, blk.x == 0 && blk.z == 0);
    float ssum0 = 0.f, ssum1 = 0.f, ssq0 = 0.f, ssq1 = 0.f;
    if (live) {
        unsigned w[5][6][2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float *wc = wgt + (long)(c + ch) * 25;
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const float f0 = wc[ky * 5 + 0], f1 = wc[ky * 5 + 1], f2 = wc[ky * 5 + 2], f3 = wc[ky * 5 + 3], f4 = wc[ky * 5 + 4];
                w[ky][0][ch] = fd_pack2(T{}, f0, f1); w[ky][1][ch] = fd_pack2(T{}, f2, f3); w[ky][2][ch] = fd_pack2(T{}, f4, 0.f);
                w[ky][3][ch] = fd_pack2(T{}, 0.f, f0); w[ky][4][ch] = fd_pack2(T{}, f1, f2); w[ky][5][ch] = fd_pack2(T{}, f3, f4);
            }
        }
        float s1[2], t1[2], s2[2], t2[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (fin.rows) { s1[ch] = s_st[2 * l + ch]; t1[ch] = s_st[64 + 2 * l + ch]; }
            else { s1[ch] = st1[FD_ST_SCALE * C + c + ch]; t1[ch] = st1[FD_ST_SHIFT * C + c + ch]; }
            s2[ch] = st2[FD_ST_SCALE * C + c + ch]; t2[ch] = st2[FD_ST_SHIFT * C + c + ch];
        }
        unsigned so[4], lo_[4];
        bool pin[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int x = xs - 2 + 2 * p;
            pin[p] = x >= 0 && x < W;
            so[p] = pin[p] ? (fd_mul24((unsigned)x, (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
            lo_[p] = pin[p] ? (fd_mul24((unsigned)(x >> 1), (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
        }
        const unsigned rowb = fd_mul24((unsigned)W, (unsigned)C) * 2u, pxb = (unsigned)C * 2u, rowl = rowb >> 1;
        const fd_bufrsrc r_sk = fd_make_rsrc(zskip + (long)n * H * W * C, (unsigned)H * rowb), r_lo = fd_make_rsrc(zin + (long)n * Hs * Ws * C, (unsigned)Hs * rowl);
        const fd_bufrsrc r_out = fd_make_rsrc(zout + (long)n * H * W * C, (unsigned)H * rowb);
        const unsigned oo = (fd_mul24((unsigned)xs, (unsigned)C) + (unsigned)c) * 2u;

        unsigned ns[2][8], nl[4];
        bool nv = false;
        auto issue = [&](int it) FD_INLINE_LAMBDA {
            const int r = y0 - 2 + 2 * it;
            nv = r >= 0 && r < H;
            if (!nv) return;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const unsigned ro = (unsigned)(r + rr) * rowb;
                    ns[rr][2 * p] = fd_buf_ld32(r_sk, so[p], ro); ns[rr][2 * p + 1] = fd_buf_ld32(r_sk, so[p], ro + pxb);
                }
#pragma unroll
            for (int p = 0; p < 4; ++p) nl[p] = fd_buf_ld32(r_lo, lo_[p], (unsigned)(r >> 1) * rowl);
        };
        unsigned win[5][4][2];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p) { win[q][p][0] = 0u; win[q][p][1] = 0u; }
        auto convert = [&](unsigned (&dst)[4][2], const unsigned (&sk)[8], bool rv) FD_INLINE_LAMBDA {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float l0 = fd_act<ACT1>(fmaf(fd_w16_lo(T{}, nl[p]), s1[0], t1[0])), l1 = fd_act<ACT1>(fmaf(fd_w16_hi(T{}, nl[p]), s1[1], t1[1]));
                const float e0 = fd_act<ACT2>(fmaf(fd_w16_lo(T{}, sk[2 * p]), s2[0], t2[0])) + l0, o0 = fd_act<ACT2>(fmaf(fd_w16_lo(T{}, sk[2 * p + 1]), s2[0], t2[0])) + l0;
                const float e1 = fd_act<ACT2>(fmaf(fd_w16_hi(T{}, sk[2 * p]), s2[1], t2[1])) + l1, o1 = fd_act<ACT2>(fmaf(fd_w16_hi(T{}, sk[2 * p + 1]), s2[1], t2[1])) + l1;
                const bool v = rv && pin[p];                 // (zero padding of the ACTIVATED, summed input)
                dst[p][0] = v ? fd_pack2(T{}, e0, o0) : 0u; dst[p][1] = v ? fd_pack2(T{}, e1, o1) : 0u;
            }
        };
        auto out_row = [&](auto SB, int y) FD_INLINE_LAMBDA {
            constexpr int sb = decltype(SB)::value;
            float acc[4][2];
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const unsigned (&R)[4][2] = win[(sb + ky) % 5];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        if (ky == 0 && k == 0) {
                            acc[0][ch] = fd_dot2_first(T{}, R[k][ch], w[ky][k][ch], 0.0f);
                            acc[1][ch] = fd_dot2_first(T{}, R[k][ch], w[ky][3 + k][ch], 0.0f);
                            acc[2][ch] = fd_dot2_first(T{}, R[k + 1][ch], w[ky][k][ch], 0.0f);
                            acc[3][ch] = fd_dot2_first(T{}, R[k + 1][ch], w[ky][3 + k][ch], 0.0f);
                        } else {
                            fd_dot2_acc(T{}, R[k][ch], w[ky][k][ch], acc[0][ch]);
                            fd_dot2_acc(T{}, R[k][ch], w[ky][3 + k][ch], acc[1][ch]);
                            fd_dot2_acc(T{}, R[k + 1][ch], w[ky][k][ch], acc[2][ch]);
                            fd_dot2_acc(T{}, R[k + 1][ch], w[ky][3 + k][ch], acc[3][ch]);
                        }
                    }
                }
            }
            fd_dot2_done(acc);
            if (y < y1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned packed = fd_pack2(T{}, acc[j][0], acc[j][1]);
                    fd_buf_st32(r_out, oo, (unsigned)y * rowb + (unsigned)j * pxb, packed);
                    const float g0 = fd_w16_lo(T{}, packed), g1 = fd_w16_hi(T{}, packed);      // statistics of the stored (rounded) output
                    ssum0 += g0; ssum1 += g1; ssq0 = fmaf(g0, g0, ssq0); ssq1 = fmaf(g1, g1, ssq1);
                }
            }
        };
        const int n_it = (y1 - y0 + 4) >> 1;
        issue(0);
        auto step = [&](auto PH, int it) FD_INLINE_LAMBDA {
            constexpr int ph = decltype(PH)::value;
            const bool rv = nv;
            unsigned hold[4][2];
            convert(win[(2 * ph) % 5], ns[0], rv);
            convert(hold, ns[1], rv);
            if (it + 1 < n_it) issue(it + 1);
            if (it >= 2) out_row(fd_int<(2 * ph + 1) % 5>{}, y0 - 4 + 2 * it);
#pragma unroll
            for (int p = 0; p < 4; ++p) { win[(2 * ph + 1) % 5][p][0] = hold[p][0]; win[(2 * ph + 1) % 5][p][1] = hold[p][1]; }
            if (it >= 2) out_row(fd_int<(2 * ph + 2) % 5>{}, y0 - 3 + 2 * it);
        };
        for (int it0 = 0; it0 < n_it; it0 += 5) {
            step(fd_int<0>{}, it0);
            if (it0 + 1 < n_it) step(fd_int<1>{}, it0 + 1);
            if (it0 + 2 < n_it) step(fd_int<2>{}, it0 + 2);
            if (it0 + 3 < n_it) step(fd_int<3>{}, it0 + 3);
            if (it0 + 4 < n_it) step(fd_int<4>{}, it0 + 4);
        }
    }
    ssum0 += __shfl_xor(ssum0, 32); ssum1 += __shfl_xor(ssum1, 32); ssq0 += __shfl_xor(ssq0, 32); ssq1 += __shfl_xor(ssq1, 32);
    if (lane < 32) {
        red[(wave * 2 + 0) * 64 + 2 * l] = ssum0; red[(wave * 2 + 0) * 64 + 2 * l + 1] = ssum1;
        red[(wave * 2 + 1) * 64 + 2 * l] = ssq0; red[(wave * 2 + 1) * 64 + 2 * l + 1] = ssq1;
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        if (c0 + ch < cend) {
            const float v = (red[(0 * 2 + which) * 64 + ch] + red[(1 * 2 + which) * 64 + ch]) + (red[(2 * 2 + which) * 64 + ch] + red[(3 * 2 + which) * 64 + ch]);
            fd_stat_add<FD_STAT_FWD>(sr, (long)n * gridDim.x + blk.x, C, which, c0 + ch, v);
        }
    }
}

// ======================================================================================================================================================
// Backward of the depthwise 3x3 stride-1 units on plain inputs (conv1.0 / conv3.0 / conv5.0 ...; autograd of reference imagenet/mobilenet.py:31-33) for the
// 16-bit train plans: the same row-walking wave, fp32 window.  Replaces the paired launch fd_dw_bwd<T, 3, 1, 0, ...> there (47 / 51 / 31 us per bf16 step).
// With 9 taps the arithmetic is cheaper as plain fp32 FMA (2 issue cycles each on gfx950: 18 cycles per value) than as dot2 on pixel pairs (6 x 4 cycles
// plus the pair packing), and a 3-row window of fp32 values is only 36 registers for a lane's 2 channels x 6 columns -- so nothing is rounded here: dz,
// the re-created input and the taps stay fp32 (fd_train_plan_lds_rounding reports 0 for these units).
//   role D: dz rows (G, z: 6 columns x 2 channels per lane) -> d_in = correlation with the flipped taps -> G_in = mask(y_in) * d_in, stored rounded, and its
//           BatchNorm-backward sums; role W: the input relu6(z_in s + t) re-created on load (6 columns), dz of the lane's 4 columns, 9 x 2 accumulators.
// One output row per step, the walk unrolled over the 3-row window's period.  Any even W (columns beyond the row: out-of-range loads, predicated stores).
// ======================================================================================================================================================
template <typename T> struct fd_dw3_bwd_args {
    const T *G, *Z, *Zin;              // this unit's dL/dy and raw output; the raw output of its producer
    T *Gin;                            // gradient handed to the producer
    const float *coef, *w, *st_in;     // BN-backward coefficients [4][C] of this unit, live taps [C][9], table [4][C] of the producer
    fd_stat_rows sr;                   // BatchNorm-backward statistics rows of the producer
    float *wpart;                      // weight-gradient partial rows: row = image * wgs_w + workgroup, [9][C]
    int H, W, C, groups_x;             // map, channels, strip groups per row (ceil(W / (4 * strips per wave)))
    int bh_d, bh_w, wgs_d, wgs_w;      // rows per band and workgroups per image and channel block of the two roles
    fd_bn_bwd_fin fin;                 // rows != null: finalised in the prologue
};

template <typename T, int ACT1, int CL>      // CL = channel lanes per strip: 32 (a wave = 2 strips x 64 channels) or 16 (4 strips x 32 channels: conv1.0)
__device__ __forceinline__ void
fd_dw3_dgrad_rows_body(const fd_dw3_bwd_args<T> &a, float *red, const float *s_cf, const int wg, const int c0, const int n, const long stat_blk)
{
    constexpr int CBF = 2 * CL;
    const int H = a.H, W = a.W, C = a.C;
    const int wave = FD_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int item = wg * 4 + wave;
    const int band = item / a.groups_x, sg = item - band * a.groups_x;
    constexpr int CB = 2 * CL, SPW = 64 / CL;
    const int cend = c0 + CB < C ? c0 + CB : C;
    const int y0 = band * a.bh_d, y1 = y0 + a.bh_d < H ? y0 + a.bh_d : H;
    const int l = lane % CL, xs = 4 * (SPW * sg + lane / CL), c = c0 + 2 * l;
    const bool live = y0 < H && c < cend && xs < W;
    float sg0 = 0.f, sg1 = 0.f, sx0 = 0.f, sx1 = 0.f;
    if (live) {
        float wf[3][3][2];                                   // flipped taps: d_in[y][x] = sum dz[y - 1 + ky][x - 1 + kx] * w[2 - ky][2 - kx]
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int t = 0; t < 9; ++t) wf[t / 3][t % 3][ch] = a.w[(long)(c + ch) * 9 + (8 - t)];
        float cA[2], c1[2], cM[2], c2[2], s1[2], t1[2], m1[2], i1[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            FD_DW_ROWS_COEF(ch);
            s1[ch] = a.st_in[FD_ST_SCALE * C + c + ch]; t1[ch] = a.st_in[FD_ST_SHIFT * C + c + ch]; m1[ch] = a.st_in[FD_ST_MEAN * C + c + ch]; i1[ch] = a.st_in[FD_ST_INVSTD * C + c + ch];
        }
        unsigned so[6];                                      // byte offsets of columns xs - 1 ... xs + 4 in a row (out of range: outside the image)
        bool pin[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int x = xs - 1 + i;
            pin[i] = x >= 0 && x < W;
            so[i] = pin[i] ? (fd_mul24((unsigned)x, (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
        }
        const unsigned rowb = fd_mul24((unsigned)W, (unsigned)C) * 2u;
        const fd_bufrsrc r_g = fd_make_rsrc(a.G + (long)n * H * W * C, (unsigned)H * rowb), r_z = fd_make_rsrc(a.Z + (long)n * H * W * C, (unsigned)H * rowb);
        const fd_bufrsrc r_zi = fd_make_rsrc(a.Zin + (long)n * H * W * C, (unsigned)H * rowb), r_gi = fd_make_rsrc(a.Gin + (long)n * H * W * C, (unsigned)H * rowb);
        unsigned ng[6], nz[6], nzi[4];                        // in flight: G / z of the next step's dz row, z_in under the next step's output row
        bool nv = false;
        // step `it`: dz row r = y0 - 1 + it enters the window, d_in row r - 1 (it >= 2) leaves
        auto issue = [&](int it) FD_INLINE_LAMBDA {
            const int r = y0 - 1 + it;
            nv = r >= 0 && r < H;
            if (nv) {
#pragma unroll
                for (int i = 0; i < 6; ++i) { ng[i] = fd_buf_ld32(r_g, so[i], (unsigned)r * rowb); nz[i] = fd_buf_ld32(r_z, so[i], (unsigned)r * rowb); }
            }
            if (it >= 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nzi[j] = fd_buf_ld32(r_zi, so[1 + j], (unsigned)(r - 1) * rowb);
            }
        };
        float win[3][6][2];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 6; ++i) { win[q][i][0] = 0.f; win[q][i][1] = 0.f; }
        const int n_it = (y1 - y0) + 2;
        issue(0);
        auto step = [&](auto PH, int it) FD_INLINE_LAMBDA {
            constexpr int ph = decltype(PH)::value;                 // row r lives in win[it % 3]
            const bool rv = nv;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool v = rv && pin[i];                 // (dz of a pixel outside the image is 0, not dz(0, 0))
                const float d0 = fd_dz(fd_w16_lo(T{}, ng[i]), fd_w16_lo(T{}, nz[i]), cA[0], c1[0], cM[0], c2[0]);
                const float d1 = fd_dz(fd_w16_hi(T{}, ng[i]), fd_w16_hi(T{}, nz[i]), cA[1], c1[1], cM[1], c2[1]);
                win[ph][i][0] = v ? d0 : 0.f; win[ph][i][1] = v ? d1 : 0.f;
            }
            unsigned zi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) zi[j] = nzi[j];
            if (it + 1 < n_it) issue(it + 1);
            if (it >= 2) {
                const int y = y0 + it - 2;                   // rows y - 1, y, y + 1 = slots (ph + 1) % 3, (ph + 2) % 3, ph
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float d0 = 0.f, d1 = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            d0 = fmaf(win[(ph + 1 + ky) % 3][j + kx][0], wf[ky][kx][0], d0);
                            d1 = fmaf(win[(ph + 1 + ky) % 3][j + kx][1], wf[ky][kx][1], d1);
                        }
                    const float z0 = fd_w16_lo(T{}, zi[j]), z1 = fd_w16_hi(T{}, zi[j]);
                    const unsigned packed = fd_pack2(T{}, d0 * fd_actmask<ACT1>(fmaf(z0, s1[0], t1[0])), d1 * fd_actmask<ACT1>(fmaf(z1, s1[1], t1[1])));
                    if (xs + j < W) {
                        fd_buf_st32(r_gi, so[1 + j], (unsigned)y * rowb, packed);
                        const float g0 = fd_w16_lo(T{}, packed), g1 = fd_w16_hi(T{}, packed);
                        sg0 += g0; sg1 += g1;
                        sx0 = fmaf(g0, (z0 - m1[0]) * i1[0], sx0); sx1 = fmaf(g1, (z1 - m1[1]) * i1[1], sx1);
                    }
                }
            }
        };
        for (int it0 = 0; it0 < n_it; it0 += 3) {
            step(fd_int<0>{}, it0);
            if (it0 + 1 < n_it) step(fd_int<1>{}, it0 + 1);
            if (it0 + 2 < n_it) step(fd_int<2>{}, it0 + 2);
        }
    }
#pragma unroll
    for (int m = CL; m < 64; m <<= 1) { sg0 += __shfl_xor(sg0, m); sg1 += __shfl_xor(sg1, m); sx0 += __shfl_xor(sx0, m); sx1 += __shfl_xor(sx1, m); }
    if (lane < CL) {
        red[(wave * 2 + 0) * CB + 2 * l] = sg0; red[(wave * 2 + 0) * CB + 2 * l + 1] = sg1;
        red[(wave * 2 + 1) * CB + 2 * l] = sx0; red[(wave * 2 + 1) * CB + 2 * l + 1] = sx1;
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 2 * CB) {
        const int which = tid / CB, ch = tid % CB;
        if (c0 + ch < cend) {
            const float v = (red[(0 * 2 + which) * CB + ch] + red[(1 * 2 + which) * CB + ch]) + (red[(2 * 2 + which) * CB + ch] + red[(3 * 2 + which) * CB + ch]);
            fd_stat_add<FD_STAT_BWD>(a.sr, stat_blk, C, which, c0 + ch, v);
        }
    }
}

template <typename T, int ACT1, int CL>
__device__ __forceinline__ void
fd_dw3_wgrad_rows_body(const fd_dw3_bwd_args<T> &a, float *red, const float *s_cf, const int wg, const int c0, const int n, const long row_blk)
{
    constexpr int CBF = 2 * CL;
    const int H = a.H, W = a.W, C = a.C;
    const int wave = FD_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int item = wg * 4 + wave;
    const int band = item / a.groups_x, sg = item - band * a.groups_x;
    constexpr int CB = 2 * CL, SPW = 64 / CL;
    const int cend = c0 + CB < C ? c0 + CB : C;
    const int y0 = band * a.bh_w, y1 = y0 + a.bh_w < H ? y0 + a.bh_w : H;
    const int l = lane % CL, xs = 4 * (SPW * sg + lane / CL), c = c0 + 2 * l;
    const bool live = y0 < H && c < cend && xs < W;
    float acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; }
    if (live) {
        float cA[2], c1[2], cM[2], c2[2], s1[2], t1[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            FD_DW_ROWS_COEF(ch);
            s1[ch] = a.st_in[FD_ST_SCALE * C + c + ch]; t1[ch] = a.st_in[FD_ST_SHIFT * C + c + ch];
        }
        unsigned so[6];
        bool pin[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int x = xs - 1 + i;
            pin[i] = x >= 0 && x < W;
            so[i] = pin[i] ? (fd_mul24((unsigned)x, (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
        }
        const unsigned rowb = fd_mul24((unsigned)W, (unsigned)C) * 2u;
        const fd_bufrsrc r_g = fd_make_rsrc(a.G + (long)n * H * W * C, (unsigned)H * rowb), r_z = fd_make_rsrc(a.Z + (long)n * H * W * C, (unsigned)H * rowb);
        const fd_bufrsrc r_zi = fd_make_rsrc(a.Zin + (long)n * H * W * C, (unsigned)H * rowb);
        unsigned nzi[6], ng[4], nz[4];                        // in flight: the next step's input row (6 columns), G / z of its dz row (4 columns)
        bool nv = false;
        // step `it`: input row r = y0 - 1 + it enters the window; dz row r - 1 (it >= 2) is multiplied with rows r - 2 ... r
        auto issue = [&](int it) FD_INLINE_LAMBDA {
            const int r = y0 - 1 + it;
            nv = r >= 0 && r < H;
            if (nv) {
#pragma unroll
                for (int i = 0; i < 6; ++i) nzi[i] = fd_buf_ld32(r_zi, so[i], (unsigned)r * rowb);
            }
            if (it >= 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { ng[j] = fd_buf_ld32(r_g, so[1 + j], (unsigned)(r - 1) * rowb); nz[j] = fd_buf_ld32(r_z, so[1 + j], (unsigned)(r - 1) * rowb); }
            }
        };
        float win[3][6][2];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < 6; ++i) { win[q][i][0] = 0.f; win[q][i][1] = 0.f; }
        const int n_it = (y1 - y0) + 2;
        issue(0);
        auto step = [&](auto PH, int it) FD_INLINE_LAMBDA {
            constexpr int ph = decltype(PH)::value;
            const bool rv = nv;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool v = rv && pin[i];                 // (zero padding of the ACTIVATED input)
                const float a0 = fd_act<ACT1>(fmaf(fd_w16_lo(T{}, nzi[i]), s1[0], t1[0])), a1 = fd_act<ACT1>(fmaf(fd_w16_hi(T{}, nzi[i]), s1[1], t1[1]));
                win[ph][i][0] = v ? a0 : 0.f; win[ph][i][1] = v ? a1 : 0.f;
            }
            float dz[4][2];
            if (it >= 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool v = xs + j < W;
                    const float d0 = fd_dz(fd_w16_lo(T{}, ng[j]), fd_w16_lo(T{}, nz[j]), cA[0], c1[0], cM[0], c2[0]);
                    const float d1 = fd_dz(fd_w16_hi(T{}, ng[j]), fd_w16_hi(T{}, nz[j]), cA[1], c1[1], cM[1], c2[1]);
                    dz[j][0] = v ? d0 : 0.f; dz[j][1] = v ? d1 : 0.f;
                }
            }
            if (it + 1 < n_it) issue(it + 1);
            if (it >= 2) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[ky * 3 + kx][0] = fmaf(dz[j][0], win[(ph + 1 + ky) % 3][j + kx][0], acc[ky * 3 + kx][0]);
                            acc[ky * 3 + kx][1] = fmaf(dz[j][1], win[(ph + 1 + ky) % 3][j + kx][1], acc[ky * 3 + kx][1]);
                        }
            }
        };
        for (int it0 = 0; it0 < n_it; it0 += 3) {
            step(fd_int<0>{}, it0);
            if (it0 + 1 < n_it) step(fd_int<1>{}, it0 + 1);
            if (it0 + 2 < n_it) step(fd_int<2>{}, it0 + 2);
        }
    }
#pragma unroll
    for (int m = CL; m < 64; m <<= 1)
#pragma unroll
        for (int t = 0; t < 9; ++t) { acc[t][0] += __shfl_xor(acc[t][0], m); acc[t][1] += __shfl_xor(acc[t][1], m); }
    if (lane < CL) {
#pragma unroll
        for (int t = 0; t < 9; ++t) { red[(wave * 9 + t) * CB + 2 * l] = acc[t][0]; red[(wave * 9 + t) * CB + 2 * l + 1] = acc[t][1]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * CB; i += 256) {
        const int t = i / CB, ch = i % CB;
        if (c0 + ch < cend)
            a.wpart[(row_blk * 9 + t) * C + c0 + ch] = (red[(0 * 9 + t) * CB + ch] + red[(1 * 9 + t) * CB + ch]) + (red[(2 * 9 + t) * CB + ch] + red[(3 * 9 + t) * CB + ch]);
    }
}

// grid (wgs_d + wgs_w, channel blocks of 2 * CL, images) through fd_xcd_image_map; block 256
template <typename T, int ACT1, int CL>
__global__ void __launch_bounds__(256)
fd_dw3_bwd_rows(const fd_dw3_bwd_args<T> a)
{
    __shared__ float red[4 * 9 * 2 * CL];
    __shared__ double sh[512];
    __shared__ float s_cf[4 * 2 * CL];
    const fd_blk3 blk = fd_xcd_image_map();
    const int c0 = blk.y * 2 * CL, n = blk.z;
    if (a.fin.rows) fd_bstat_table_block(a.fin, sh, s_cf, c0, 2 * CL, a.C, (int)threadIdx.x, blk.x == 0 && n == 0);
    if (blk.x < a.wgs_d) fd_dw3_dgrad_rows_body<T, ACT1, CL>(a, red, s_cf, blk.x, c0, n, (long)n * a.wgs_d + blk.x);
    else fd_dw3_wgrad_rows_body<T, ACT1, CL>(a, red, s_cf, blk.x - a.wgs_d, c0, n, (long)n * a.wgs_w + (blk.x - a.wgs_d));
}

// ======================================================================================================================================================
// Backward of the depthwise 3x3 STRIDE-2 units (conv2.0 / conv4.0 / conv6.0 / conv12.0; autograd of reference imagenet/mobilenet.py:31-33) for the 16-bit train
// plans: ONE role -- a wave produces BOTH gradients from one pass over its operands (fd_dw3s2_bwd_rows).  These units move bytes, not flops (2.25 taps per input
// value for the data gradient, 9 per OUTPUT value for the weight gradient): the register-window pair fd_dw3s2_dgrad_rows + fd_dw3_wgrad_rows read the saved input
// z_in twice and G / z twice (49 + 17 us on conv2.0 for ~180 MB).  Here a lane owns 2 channels x 4 OUTPUT columns (8 input columns + 1 halo column) and walks down
// output rows: per step it takes in input rows 2oy, 2oy + 1 (activated once, fp32, kept with row 2oy - 1 in a 3-row window), the dz row oy + 1 (5 columns, kept
// with row oy), the skip gradient of the two input rows, and emits
//   d_in(2oy, x), d_in(2oy + 1, x)  for its 8 input columns (x even: one tap column, x odd: two; row 2oy: filter row 1, row 2oy + 1: filter rows 2 and 0),
//   G_in = mask(y_in) * (d_in + skip gradient), rounded, + its BatchNorm-backward sums,   dW[ky][kx] += dz(oy, ox) * a_in(2oy - 1 + ky, 2ox - 1 + kx).
// Nothing is rounded before the stores (fp32 window, fp32 taps).  End-of-kernel reductions as in fd_dw3_bwd_rows.
// ======================================================================================================================================================
template <typename T> struct fd_dw3s2_bwd_args {
    const T *G, *Z, *Zin, *SG;         // this unit's dL/dy and raw output (OUTPUT resolution); the producer's raw output and (ADD_SG) the decoder's skip gradient (INPUT resolution)
    T *Gin;
    const float *coef, *w, *st_in;
    fd_stat_rows sr;
    float *wpart;                      // [9][C] per workgroup
    int Ho, Wo, C, groups_x;           // OUTPUT map (the input map is 2 Ho x 2 Wo), channels, strip pairs per output row (ceil(Wo / 8))
    int bh, wgs;                       // output rows per band, workgroups per image and channel block
    fd_bn_bwd_fin fin;                 // rows != null: finalised in the prologue
};

template <typename T, int ACT1, int ADD_SG>
__global__ void __launch_bounds__(256) FD_DW5B_ATTR
fd_dw3s2_bwd_rows(const fd_dw3s2_bwd_args<T> a)
{
    __shared__ float red[4 * 9 * 64];
    __shared__ double sh[512];
    __shared__ float s_cf[4 * 64];
    constexpr int CBF = 64;
    const fd_blk3 blk = fd_xcd_image_map();
    const int c0 = blk.y * 64, n = blk.z;
    const int Ho = a.Ho, Wo = a.Wo, C = a.C, H = 2 * Ho, W = 2 * Wo;
    if (a.fin.rows) fd_bstat_table_block(a.fin, sh, s_cf, c0, 64, C, (int)threadIdx.x, blk.x == 0 && n == 0);
    const int wave = FD_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int item = blk.x * 4 + wave;
    const int band = item / a.groups_x, sg = item - band * a.groups_x;
    const int cend = c0 + 64 < C ? c0 + 64 : C;
    const int oy0 = band * a.bh, oy1 = oy0 + a.bh < Ho ? oy0 + a.bh : Ho;
    const int l = lane & 31, ox0 = 4 * (2 * sg + (lane >> 5)), c = c0 + 2 * l;
    const bool live = oy0 < Ho && c < cend && ox0 < Wo;
    float acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; }
    float sg0 = 0.f, sg1 = 0.f, sx0 = 0.f, sx1 = 0.f;
    if (live) {
        float wt[3][3][2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int t = 0; t < 9; ++t) wt[t / 3][t % 3][ch] = a.w[(long)(c + ch) * 9 + t];
        float cA[2], c1[2], cM[2], c2[2], s1[2], t1[2], m1[2], i1[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            FD_DW_ROWS_COEF(ch);
            s1[ch] = a.st_in[FD_ST_SCALE * C + c + ch]; t1[ch] = a.st_in[FD_ST_SHIFT * C + c + ch]; m1[ch] = a.st_in[FD_ST_MEAN * C + c + ch]; i1[ch] = a.st_in[FD_ST_INVSTD * C + c + ch];
        }
        unsigned si[9], sd[5];                               // byte offsets: input columns 2 ox0 - 1 ... 2 ox0 + 7 in an input row; dz columns ox0 ... ox0 + 4 in an output row
        bool pi[9], pd[5];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int x = 2 * ox0 - 1 + i;
            pi[i] = x >= 0 && x < W;
            si[i] = pi[i] ? (fd_mul24((unsigned)x, (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int x = ox0 + i;
            pd[i] = x < Wo;
            sd[i] = pd[i] ? (fd_mul24((unsigned)x, (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
        }
        const unsigned rowi = fd_mul24((unsigned)W, (unsigned)C) * 2u, rowo = fd_mul24((unsigned)Wo, (unsigned)C) * 2u;
        const fd_bufrsrc r_g = fd_make_rsrc(a.G + (long)n * Ho * Wo * C, (unsigned)Ho * rowo), r_z = fd_make_rsrc(a.Z + (long)n * Ho * Wo * C, (unsigned)Ho * rowo);
        const fd_bufrsrc r_zi = fd_make_rsrc(a.Zin + (long)n * H * W * C, (unsigned)H * rowi), r_gi = fd_make_rsrc(a.Gin + (long)n * H * W * C, (unsigned)H * rowi);
        const fd_bufrsrc r_sg = fd_make_rsrc(ADD_SG ? a.SG + (long)n * H * W * C : a.Zin, (unsigned)H * rowi);

        unsigned nzi[2][9], nsg[2][8], ng[5], nz[5];          // in flight for the next step: input rows 2oy, 2oy + 1 (9 columns), their skip gradient (8), G / z of dz row oy + 1 (5)
        bool nvd = false;                                    // dz row oy + 1 exists
        auto issue = [&](int oy) FD_INLINE_LAMBDA {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const unsigned ro = (unsigned)(2 * oy + rr) * rowi;
#pragma unroll
                for (int i = 0; i < 9; ++i) nzi[rr][i] = fd_buf_ld32(r_zi, si[i], ro);
                if (ADD_SG) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) nsg[rr][i] = fd_buf_ld32(r_sg, si[1 + i], ro);
                }
            }
            nvd = oy + 1 < Ho;
            if (nvd) {
#pragma unroll
                for (int i = 0; i < 5; ++i) { ng[i] = fd_buf_ld32(r_g, sd[i], (unsigned)(oy + 1) * rowo); nz[i] = fd_buf_ld32(r_z, sd[i], (unsigned)(oy + 1) * rowo); }
            }
        };
        auto make_dz = [&](float (&dst)[5][2], const unsigned (&g)[5], const unsigned (&z)[5], bool rv) FD_INLINE_LAMBDA {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const bool v = rv && pd[i];
                const float d0 = fd_dz(fd_w16_lo(T{}, g[i]), fd_w16_lo(T{}, z[i]), cA[0], c1[0], cM[0], c2[0]);
                const float d1 = fd_dz(fd_w16_hi(T{}, g[i]), fd_w16_hi(T{}, z[i]), cA[1], c1[1], cM[1], c2[1]);
                dst[i][0] = v ? d0 : 0.f; dst[i][1] = v ? d1 : 0.f;
            }
        };
        float inw[3][9][2];                                  // activated input rows 2oy - 1, 2oy, 2oy + 1: slot 0 = the row kept from the previous step
        float dzw[2][5][2];                                  // dz rows oy (slot 0) and oy + 1 (slot 1)
        // prologue: input row 2 oy0 - 1 (zero above the image) and dz row oy0
        {
            const int r = 2 * oy0 - 1;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                unsigned v = 0u;
                if (r >= 0) v = fd_buf_ld32(r_zi, si[i], (unsigned)r * rowi);
                const bool ok = r >= 0 && pi[i];
                inw[0][i][0] = ok ? fd_act<ACT1>(fmaf(fd_w16_lo(T{}, v), s1[0], t1[0])) : 0.f;
                inw[0][i][1] = ok ? fd_act<ACT1>(fmaf(fd_w16_hi(T{}, v), s1[1], t1[1])) : 0.f;
            }
            unsigned g0[5], z0[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) { g0[i] = fd_buf_ld32(r_g, sd[i], (unsigned)oy0 * rowo); z0[i] = fd_buf_ld32(r_z, sd[i], (unsigned)oy0 * rowo); }
            make_dz(dzw[0], g0, z0, true);
        }
        issue(oy0);
        for (int oy = oy0; oy < oy1; ++oy) {
            // this step's operands leave the in-flight registers: the two new input rows (activated; the raw words stay for xhat), dz row oy + 1, the skip gradient
            unsigned zraw[2][8], sgr[2][8];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    inw[1 + rr][i][0] = pi[i] ? fd_act<ACT1>(fmaf(fd_w16_lo(T{}, nzi[rr][i]), s1[0], t1[0])) : 0.f;
                    inw[1 + rr][i][1] = pi[i] ? fd_act<ACT1>(fmaf(fd_w16_hi(T{}, nzi[rr][i]), s1[1], t1[1])) : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) { zraw[rr][i] = nzi[rr][1 + i]; sgr[rr][i] = ADD_SG ? nsg[rr][i] : 0u; }
            }
            make_dz(dzw[1], ng, nz, nvd);
            if (oy + 1 < oy1) issue(oy + 1);                   // everything the next step reads flies under this step's arithmetic
            // ---- weight gradient: dz(oy, ox0 + j) * a_in(2oy - 1 + ky, 2(ox0 + j) - 1 + kx): window column 2 j + kx
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[ky * 3 + kx][0] = fmaf(dzw[0][j][0], inw[ky][2 * j + kx][0], acc[ky * 3 + kx][0]);
                        acc[ky * 3 + kx][1] = fmaf(dzw[0][j][1], inw[ky][2 * j + kx][1], acc[ky * 3 + kx][1]);
                    }
            // ---- data gradient of input rows 2oy (filter row 1 of dz row oy) and 2oy + 1 (filter row 2 of dz row oy + filter row 0 of dz row oy + 1);
            // input column 2 m (window column 1 + 2 (m - ox0)): tap column 1 of dz column m; column 2 m + 1: tap column 2 of dz column m + tap column 0 of dz column m + 1
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = i >> 1;
                    float d[2];
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        if ((i & 1) == 0) {
                            d[ch] = rr == 0 ? dzw[0][m][ch] * wt[1][1][ch] : fmaf(dzw[0][m][ch], wt[2][1][ch], dzw[1][m][ch] * wt[0][1][ch]);
                        } else {
                            const float e = rr == 0 ? fmaf(dzw[0][m][ch], wt[1][2][ch], dzw[0][m + 1][ch] * wt[1][0][ch])
                                                    : fmaf(dzw[0][m][ch], wt[2][2][ch], fmaf(dzw[0][m + 1][ch], wt[2][0][ch], fmaf(dzw[1][m][ch], wt[0][2][ch], dzw[1][m + 1][ch] * wt[0][0][ch])));
                            d[ch] = e;
                        }
                    }
                    if (ADD_SG) { d[0] += fd_w16_lo(T{}, sgr[rr][i]); d[1] += fd_w16_hi(T{}, sgr[rr][i]); }
                    const float z0 = fd_w16_lo(T{}, zraw[rr][i]), z1 = fd_w16_hi(T{}, zraw[rr][i]);
                    const unsigned packed = fd_pack2(T{}, d[0] * fd_actmask<ACT1>(fmaf(z0, s1[0], t1[0])), d[1] * fd_actmask<ACT1>(fmaf(z1, s1[1], t1[1])));
                    if (ox0 + m < Wo) {
                        fd_buf_st32(r_gi, si[1 + i], (unsigned)(2 * oy + rr) * rowi, packed);
                        const float g0 = fd_w16_lo(T{}, packed), g1 = fd_w16_hi(T{}, packed);
                        sg0 += g0; sg1 += g1;
                        sx0 = fmaf(g0, (z0 - m1[0]) * i1[0], sx0); sx1 = fmaf(g1, (z1 - m1[1]) * i1[1], sx1);
                    }
                }
            // ---- the windows move on: input row 2oy + 1 becomes the next step's row 2(oy + 1) - 1, dz row oy + 1 its row oy
#pragma unroll
            for (int i = 0; i < 9; ++i) { inw[0][i][0] = inw[2][i][0]; inw[0][i][1] = inw[2][i][1]; }
#pragma unroll
            for (int i = 0; i < 5; ++i) { dzw[0][i][0] = dzw[1][i][0]; dzw[0][i][1] = dzw[1][i][1]; }
        }
    }
    // ---- end of kernel: the workgroup's BatchNorm-backward sums and its weight-gradient partial row
    sg0 += __shfl_xor(sg0, 32); sg1 += __shfl_xor(sg1, 32); sx0 += __shfl_xor(sx0, 32); sx1 += __shfl_xor(sx1, 32);
    if (lane < 32) {
        red[(wave * 2 + 0) * 64 + 2 * l] = sg0; red[(wave * 2 + 0) * 64 + 2 * l + 1] = sg1;
        red[(wave * 2 + 1) * 64 + 2 * l] = sx0; red[(wave * 2 + 1) * 64 + 2 * l + 1] = sx1;
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 128) {
        const int which = tid >> 6, ch = tid & 63;
        if (c0 + ch < cend) {
            const float v = (red[(0 * 2 + which) * 64 + ch] + red[(1 * 2 + which) * 64 + ch]) + (red[(2 * 2 + which) * 64 + ch] + red[(3 * 2 + which) * 64 + ch]);
            fd_stat_add<FD_STAT_BWD>(a.sr, (long)n * a.wgs + blk.x, C, which, c0 + ch, v);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t][0] += __shfl_xor(acc[t][0], 32); acc[t][1] += __shfl_xor(acc[t][1], 32); }
    if (lane < 32) {
#pragma unroll
        for (int t = 0; t < 9; ++t) { red[(wave * 9 + t) * 64 + 2 * l] = acc[t][0]; red[(wave * 9 + t) * 64 + 2 * l + 1] = acc[t][1]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
        const int t = i >> 6, ch = i & 63;
        if (c0 + ch < cend)
            a.wpart[(((long)n * a.wgs + blk.x) * 9 + t) * C + c0 + ch] = (red[(0 * 9 + t) * 64 + ch] + red[(1 * 9 + t) * 64 + ch]) + (red[(2 * 9 + t) * 64 + ch] + red[(3 * 9 + t) * 64 + ch]);
    }
}

// ======================================================================================================================================================
// Train-mode FORWARD of the depthwise 3x3 units on plain inputs (every encoder unit; reference imagenet/mobilenet.py:31-33) for the 16-bit train plans:
//   z_out = conv3x3_stride_S( act(z_in s + t) ),  raw, rounded to the storage type, + the workgroup's per-channel sums of the ROUNDED values added to the
//   unit's statistics rows; fin.rows != null: the producer's BatchNorm is finalised in the prologue (fd_stat_table_block; workgroup (0, *, 0) writes).
// The row-walking wave of fd_dw3_bwd_rows / fd_dw3s2_bwd_rows: a lane owns 2 adjacent channels x 4 adjacent OUTPUT columns, the activated input rows live
// in an fp32 register window (nothing but the stored output is rounded), the taps are fp32 (v_fmac_f32: 2 issue cycles), no LDS staging, no barrier before
// the end-of-kernel reduction.  S = 1: 6 input columns, one input row in / one output row out per step (3-row window).  S = 2: 9 input columns, two input
// rows in per output row; the lower one is carried to the next step as its top row.
// Replaces fd_dw3_rows_train (4 channels x 1 column per work-item, 3 x 3 loads per output) on the large maps and the LDS-tiled fd_dwconv_train on the small.
// grid (wgs, channel blocks of 2 CL, images) through fd_xcd_image_map; block 256 = 4 independent waves; C % 8 == 0; S = 2: H, W even.
// ======================================================================================================================================================
template <typename T, int S, int ACT1, int CL>
__global__ void __launch_bounds__(256)
fd_dw3_rows_fwd(const T *__restrict__ zin, const float *__restrict__ st1, const float *__restrict__ wgt, T *__restrict__ zout, fd_stat_rows sr,
                int H, int W, int Ho, int Wo, int C, int groups_x, int bh, const fd_bn_fin fin)
{
    constexpr int CB = 2 * CL, SPW = 64 / CL, NI = S == 1 ? 6 : 9;
    __shared__ double sh[512];
    __shared__ float s_st[2 * CB];
    __shared__ float red[4 * 2 * CB];
    const fd_blk3 blk = fd_xcd_image_map();
    const int c0 = blk.y * CB, cend = c0 + CB < C ? c0 + CB : C, n = blk.z;
    const int wave = FD_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int item = blk.x * 4 + wave;
    const int band = item / groups_x, sg = item - band * groups_x;
    const int y0 = band * bh, y1 = y0 + bh < Ho ? y0 + bh : Ho;
    const int l = lane % CL, xs = 4 * (SPW * sg + lane / CL), c = c0 + 2 * l;
    const bool live = y0 < Ho && c < cend && xs < Wo;
    if (fin.rows) fd_stat_table_block(fin, sh, s_st, c0, CB, C, (int)threadIdx.x, blk.x == 0 && blk.z == 0);
    float ssum0 = 0.f, ssum1 = 0.f, ssq0 = 0.f, ssq1 = 0.f;
    if (live) {
        float w[3][3][2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t / 3][t % 3][ch] = wgt[(long)(c + ch) * 9 + t];
        float s1[2], t1[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (fin.rows) { s1[ch] = s_st[2 * l + ch]; t1[ch] = s_st[CB + 2 * l + ch]; }
            else { s1[ch] = st1[FD_ST_SCALE * C + c + ch]; t1[ch] = st1[FD_ST_SHIFT * C + c + ch]; }
        }
        unsigned so[NI];                                     // byte offsets of the lane's input columns S xs - 1 ... in a row (outside the image: out of range)
        bool pin[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int x = S * xs - 1 + i;
            pin[i] = x >= 0 && x < W;
            so[i] = pin[i] ? (fd_mul24((unsigned)x, (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
        }
        const unsigned rowb = fd_mul24((unsigned)W, (unsigned)C) * 2u, rowo = fd_mul24((unsigned)Wo, (unsigned)C) * 2u, pxb = (unsigned)C * 2u;
        const fd_bufrsrc r_in = fd_make_rsrc(zin + (long)n * H * W * C, (unsigned)H * rowb), r_out = fd_make_rsrc(zout + (long)n * Ho * Wo * C, (unsigned)Ho * rowo);
        const unsigned oo = (fd_mul24((unsigned)xs, (unsigned)C) + (unsigned)c) * 2u;
        // a loaded row -> activated fp32 values (zero padding of the ACTIVATED input: act(t) != 0 in general)
        auto convert = [&](float (&dst)[NI][2], const unsigned (&src)[NI], bool rv) FD_INLINE_LAMBDA {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const bool v = rv && pin[i];
                const float a0 = fd_act<ACT1>(fmaf(fd_w16_lo(T{}, src[i]), s1[0], t1[0])), a1 = fd_act<ACT1>(fmaf(fd_w16_hi(T{}, src[i]), s1[1], t1[1]));
                dst[i][0] = v ? a0 : 0.f; dst[i][1] = v ? a1 : 0.f;
            }
        };
        auto emit = [&](const float (&acc)[4][2], int y) FD_INLINE_LAMBDA {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned packed = fd_pack2(T{}, acc[j][0], acc[j][1]);
                if (xs + j < Wo) {
                    fd_buf_st32(r_out, oo, (unsigned)y * rowo + (unsigned)j * pxb, packed);
                    const float g0 = fd_w16_lo(T{}, packed), g1 = fd_w16_hi(T{}, packed);          // statistics of the stored (rounded) output
                    ssum0 += g0; ssum1 += g1; ssq0 = fmaf(g0, g0, ssq0); ssq1 = fmaf(g1, g1, ssq1);
                }
            }
        };
        if constexpr (S == 1) {
            // two rows in flight: at one or two waves per SIMD (the small maps) a step's load latency is not covered by the other waves
            unsigned nx[2][NI];
            bool nv[2] = {false, false};
            auto issue = [&](auto B, int it) FD_INLINE_LAMBDA {  // step `it`: input row r = y0 - 1 + it enters, output row r - 1 (it >= 2) leaves
                constexpr int b = decltype(B)::value;
                const int r = y0 - 1 + it;
                nv[b] = r >= 0 && r < H;
                if (nv[b]) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) nx[b][i] = fd_buf_ld32(r_in, so[i], (unsigned)r * rowb);
                }
            };
            float win[3][NI][2];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < NI; ++i) { win[q][i][0] = 0.f; win[q][i][1] = 0.f; }
            const int n_it = (y1 - y0) + 2;                      // >= 3
            issue(fd_int<0>{}, 0);
            issue(fd_int<1>{}, 1);
            auto step = [&](auto PH, int it) FD_INLINE_LAMBDA {
                constexpr int ph = decltype(PH)::value % 3, b = decltype(PH)::value % 2;
                convert(win[ph], nx[b], nv[b]);
                if (it + 2 < n_it) issue(fd_int<b>{}, it + 2);
                if (it >= 2) {                                   // rows y - 1, y, y + 1 = slots (ph + 1) % 3, (ph + 2) % 3, ph
                    float acc[4][2];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float d0 = 0.f, d1 = 0.f;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                d0 = fmaf(win[(ph + 1 + ky) % 3][j + kx][0], w[ky][kx][0], d0);
                                d1 = fmaf(win[(ph + 1 + ky) % 3][j + kx][1], w[ky][kx][1], d1);
                            }
                        acc[j][0] = d0; acc[j][1] = d1;
                    }
                    emit(acc, y0 + it - 2);
                }
            };
            for (int it0 = 0; it0 < n_it; it0 += 6) {
                step(fd_int<0>{}, it0);
                if (it0 + 1 < n_it) step(fd_int<1>{}, it0 + 1);
                if (it0 + 2 < n_it) step(fd_int<2>{}, it0 + 2);
                if (it0 + 3 < n_it) step(fd_int<3>{}, it0 + 3);
                if (it0 + 4 < n_it) step(fd_int<4>{}, it0 + 4);
                if (it0 + 5 < n_it) step(fd_int<5>{}, it0 + 5);
            }
        } else {
            unsigned nb[2][NI], nc[2][NI];                       // in flight: input rows 2 oy and 2 oy + 1 of the next TWO steps
            auto issue = [&](auto B, int oy) FD_INLINE_LAMBDA {
                constexpr int b = decltype(B)::value;
#pragma unroll
                for (int i = 0; i < NI; ++i) { nb[b][i] = fd_buf_ld32(r_in, so[i], (unsigned)(2 * oy) * rowb); nc[b][i] = fd_buf_ld32(r_in, so[i], (unsigned)(2 * oy + 1) * rowb); }
            };
            float top[2][NI][2], mid[NI][2];                     // top[ph]: row 2 oy - 1 of this step; the step's row 2 oy + 1 lands in top[1 - ph]
            {
                const bool tv = y0 > 0;
                unsigned tx[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) tx[i] = tv ? fd_buf_ld32(r_in, so[i], (unsigned)(2 * y0 - 1) * rowb) : 0u;
                issue(fd_int<0>{}, y0);
                if (y0 + 1 < y1) issue(fd_int<1>{}, y0 + 1);
                convert(top[0], tx, tv);
            }
            auto step = [&](auto PH, int oy) FD_INLINE_LAMBDA {
                constexpr int ph = decltype(PH)::value;
                convert(mid, nb[ph], true);
                convert(top[1 - ph], nc[ph], true);
                if (oy + 2 < y1) issue(fd_int<ph>{}, oy + 2);
                float acc[4][2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float d0 = 0.f, d1 = 0.f;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        d0 = fmaf(top[ph][2 * j + kx][0], w[0][kx][0], d0); d1 = fmaf(top[ph][2 * j + kx][1], w[0][kx][1], d1);
                        d0 = fmaf(mid[2 * j + kx][0], w[1][kx][0], d0); d1 = fmaf(mid[2 * j + kx][1], w[1][kx][1], d1);
                        d0 = fmaf(top[1 - ph][2 * j + kx][0], w[2][kx][0], d0); d1 = fmaf(top[1 - ph][2 * j + kx][1], w[2][kx][1], d1);
                    }
                    acc[j][0] = d0; acc[j][1] = d1;
                }
                emit(acc, oy);
            };
            for (int oy = y0; oy < y1; oy += 2) {
                step(fd_int<0>{}, oy);
                if (oy + 1 < y1) step(fd_int<1>{}, oy + 1);
            }
        }
    }
#pragma unroll
    for (int m = CL; m < 64; m <<= 1) { ssum0 += __shfl_xor(ssum0, m); ssum1 += __shfl_xor(ssum1, m); ssq0 += __shfl_xor(ssq0, m); ssq1 += __shfl_xor(ssq1, m); }
    if (lane < CL) {
        red[(wave * 2 + 0) * CB + 2 * l] = ssum0; red[(wave * 2 + 0) * CB + 2 * l + 1] = ssum1;
        red[(wave * 2 + 1) * CB + 2 * l] = ssq0; red[(wave * 2 + 1) * CB + 2 * l + 1] = ssq1;
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 2 * CB) {
        const int which = tid / CB, ch = tid % CB;
        if (c0 + ch < cend) {
            const float v = (red[(0 * 2 + which) * CB + ch] + red[(1 * 2 + which) * CB + ch]) + (red[(2 * 2 + which) * CB + ch] + red[(3 * 2 + which) * CB + ch]);
            fd_stat_add<FD_STAT_FWD>(sr, (long)n * gridDim.x + blk.x, C, which, c0 + ch, v);
        }
    }
}

// ======================================================================================================================================================
// Stem backward-weights for the 16-bit train plans (conv_bn(3, 32, 2): imagenet/mobilenet.py:22-27; autograd of the dense 3x3 stride-2 convolution):
//   dW[co][c][ky][kx] = sum over output pixels of dz[px][co] * x[c][2 oy - 1 + ky][2 ox - 1 + kx],   dz = A((G - c1) - (z - mu) c2) formed on load.
// 864 MACs per output pixel, 0.35 GMAC per step at B = 32: nothing for the VALU (4 us of fp32 FMA issue) -- the MFMA kernel it replaces (fd_stem_wgrad: 256-pixel
// blocks, 27 scalar stride-2 gathers per work-item into LDS, dz staged in LDS, two barriers per block) ran 55 us for 70 MB.  Here the row-walking wave again:
// a lane owns 4 adjacent output channels (two 16-bit words of G / z) x 4 adjacent output columns and walks down a band of output rows; per row and input plane
// it loads the 3 x 9 input values its 4 pixels touch (two aligned 16-byte loads + one scalar per input row; the 8 channel lanes of a column group read the same
// addresses -- with 2 channels per lane and 16 lanes per group the texture path, not the VALU, was the bound: 32 us) and adds 4 x 9 x 4 products into its 27 x 4 accumulators.  No LDS, no barrier until the end-of-kernel reduction (column groups
// by shuffle, waves through LDS); one partial row [Cout][27] per workgroup, reduced by fd_reduce_weights_batch_f32.
// grid (wgs, images); block 256 = 4 independent waves (16 channel lanes x 4 column groups); Cout == 32, Wo % 16 == 0 is NOT required (Wo % 4 == 0 is), H = 2 Ho.
// ======================================================================================================================================================
// (200 VGPRs, two waves per SIMD: the compiler keeps all three planes' loads of a row in flight.  Held to three waves (168) it spills 140 bytes;
// with 4 channels per lane -- half the redundant input loads -- the 108 accumulators leave one wave per SIMD: both measured slower or not at all)
template <typename T, int CL, int CPL>  // CPL channels per lane (2 or 4), CL = Cout / CPL channel lanes; a wave = CL channel lanes x 64 / CL column groups of 4 output columns
__global__ void __launch_bounds__(256)
fd_stem_wgrad_rows(const float *__restrict__ x, const T *__restrict__ G, const T *__restrict__ Z, const float *__restrict__ coef, float *__restrict__ wpart,
                   int H, int W, int groups_x, int bh)
{
    constexpr int C = CPL * CL, GPW = 64 / CL, NA = 27 * CPL, WPL = CPL / 2;
    __shared__ float red[4 * NA * CL];
    const int Ho = H >> 1, Wo = W >> 1;
    const int n = blockIdx.y;
    const int wave = FD_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    const int band = item / groups_x, sg = item - band * groups_x;
    const int y0 = band * bh, y1 = y0 + bh < Ho ? y0 + bh : Ho;
    const int l = lane % CL, ox0 = 4 * (GPW * sg + lane / CL), c = CPL * l;
    const bool live = y0 < Ho && ox0 < Wo;
    float acc[27][CPL];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int ch = 0; ch < CPL; ++ch) acc[t][ch] = 0.f;
    if (live) {
        float cA[CPL], c1[CPL], cM[CPL], c2[CPL];
#pragma unroll
        for (int ch = 0; ch < CPL; ++ch) { cA[ch] = coef[FD_CF_A * C + c + ch]; c1[ch] = coef[FD_CF_C1 * C + c + ch]; cM[ch] = coef[FD_CF_MU * C + c + ch]; c2[ch] = coef[FD_CF_C2 * C + c + ch]; }
        const float *xn = x + (long)n * 3 * H * W + 2 * ox0;                  // column 2 ox0 of plane 0, row 0 (16-byte aligned: ox0 % 4 == 0)
        const unsigned *gw = reinterpret_cast<const unsigned *>(G) + (((long)n * Ho * Wo + ox0) * C + c) / 2;
        const unsigned *zw = reinterpret_cast<const unsigned *>(Z) + (((long)n * Ho * Wo + ox0) * C + c) / 2;
        const bool left = ox0 > 0;
        for (int oy = y0; oy < y1; ++oy) {
            float dz[4][CPL];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long o = ((long)oy * Wo + j) * (C / 2);
#pragma unroll
                for (int q = 0; q < WPL; ++q) {
                    const unsigned g = gw[o + q], z = zw[o + q];
                    dz[j][2 * q] = fd_dz(fd_w16_lo(T{}, g), fd_w16_lo(T{}, z), cA[2 * q], c1[2 * q], cM[2 * q], c2[2 * q]);
                    dz[j][2 * q + 1] = fd_dz(fd_w16_hi(T{}, g), fd_w16_hi(T{}, z), cA[2 * q + 1], c1[2 * q + 1], cM[2 * q + 1], c2[2 * q + 1]);
                }
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                float xw[3][9];                                               // input rows 2 oy - 1 ... 2 oy + 1, columns 2 ox0 - 1 ... 2 ox0 + 7 of plane p
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = 2 * oy - 1 + ky;
                    const bool rv = iy >= 0;                                  // (iy <= 2 Ho - 1 = H - 1 always)
                    const float *row = xn + ((long)p * H + (rv ? iy : 0)) * W;
                    const fd_f32x4 a = fd_ld4(row), b = fd_ld4(row + 4);
                    const float e = left ? row[-1] : 0.0f;
                    xw[ky][0] = rv ? e : 0.f;
                    xw[ky][1] = rv ? a.x : 0.f; xw[ky][2] = rv ? a.y : 0.f; xw[ky][3] = rv ? a.z : 0.f; xw[ky][4] = rv ? a.w : 0.f;
                    xw[ky][5] = rv ? b.x : 0.f; xw[ky][6] = rv ? b.y : 0.f; xw[ky][7] = rv ? b.z : 0.f; xw[ky][8] = rv ? b.w : 0.f;
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int ch = 0; ch < CPL; ++ch)
                                acc[(p * 3 + ky) * 3 + kx][ch] = fmaf(dz[j][ch], xw[ky][2 * j + kx], acc[(p * 3 + ky) * 3 + kx][ch]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int ch = 0; ch < CPL; ++ch)
#pragma unroll
            for (int m = CL; m < 64; m <<= 1) acc[t][ch] += __shfl_xor(acc[t][ch], m);
    if (lane < CL) {
#pragma unroll
        for (int t = 0; t < 27; ++t)
#pragma unroll
            for (int ch = 0; ch < CPL; ++ch) red[(wave * NA + CPL * t + ch) * CL + l] = acc[t][ch];
    }
    __syncthreads();
    float *o = wpart + ((long)n * gridDim.x + blockIdx.x) * (C * 27);
    for (int i = threadIdx.x; i < NA * CL; i += 256) {
        const int q = i / CL, ll = i % CL, t = q / CPL, ch = q % CPL;         // q = CPL t + ch
        o[(CPL * ll + ch) * 27 + t] = (red[(0 * NA + q) * CL + ll] + red[(1 * NA + q) * CL + ll]) + (red[(2 * NA + q) * CL + ll] + red[(3 * NA + q) * CL + ll]);
    }
}
