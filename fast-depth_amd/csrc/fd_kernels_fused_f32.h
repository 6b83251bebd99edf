// fd_kernels_fused_f32.h -- fused depthwise-separable unit for gfx950:  out = act2(W_pw * act1(dw_KxK(in) + b_dw) + b_pw)
// (reference: one conv_dw block, imagenet/mobilenet.py:29-38, or decode_conv1 = depthwise(5) + pointwise, models.py:683-685),
// stride 1, input read as stored (no upsample / skip), BatchNorm folded (inference).
//
// Why: after the unfused kernels were tuned, the depthwise kernels of the 14x14 / 7x7 layers were pure launch + latency
// (8-12 us for 6-13 MB) and every depthwise output made an HBM round trip only to be re-read as the GEMM's A operand.  Here the
// depthwise result never leaves the CU: it is produced straight into the GEMM's A tile in LDS.
//
// Workgroup = 64 output pixels x 64 output channels, K loop over 32-channel chunks:
//   stage (LDS-DMA, double buffered, one chunk ahead):  the INPUT PATCH  [NP pixels][32 ch]  that the 64 pixels' KxK windows touch,
//                                                        the pointwise weight tile [64 n][32 k] (XOR-swizzled for the fragment reads),
//                                                        the K*K depthwise tap rows [K*K][32 ch]
//   dw phase (VALU):  work-item = (pixel r and r+32, 4 channels): K*K taps from the LDS patch -> bias, activation -> A tile
//                     (written with the GEMM swizzle)
//   mma phase:        16 x v_mfma_f32_32x32x2_f32 per wave on the A tile x weight tile (same fragment scheme as fd_pw_gemm_f32)
// Two pixel->tile mappings share all code through (patch base, row stride RS):
//   flat (small maps, W <= 28): the 64 pixels are consecutive in the flattened (n, y, x) order, the patch is the contiguous pixel
//        range [p0 - P*(W+1), p0 + 63 + P*(W+1)] (it may straddle rows and images: validity masks kill the out-of-image taps), RS = W;
//   2-D  (W, H multiples of 8): the 64 pixels are an 8 x 8 tile, the patch is its (8+2P)^2 halo box, RS = 8 + 2P.
// Zero padding is a per-pixel K*K-bit mask computed once (the pixel of a work-item does not change over the K loop).
#pragma once
#include "fd_device.h"

template <int K, int ACT1, int ACT2, int GPW>
__global__ void __launch_bounds__(256)
fd_sep_unit_f32(const float *__restrict__ in, const float *__restrict__ wdw, const float *__restrict__ bdw,
                const float *__restrict__ Wt, const float *__restrict__ bias, float *__restrict__ out,
                int Bimg, int H, int W, int C, int C32, int N, int NP, int flat, int m_tiles, int n_tiles)
{
    constexpr int P = K / 2, KK = K * K, BK = 32;
    constexpr int TAPROWS = (KK + 7) / 8 * 8;                 // tap rows padded to a whole LDS-DMA group
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int PATCH = NP * BK, WTILE = 64 * BK, TAPS = TAPROWS * BK;
    const int STAGE = PATCH + WTILE + TAPS;                    // floats per stage (patch | weight tile | tap rows)
    float *atile = smem + 3 * STAGE;                           // [64][32], GEMM-swizzled; followed by a 1 KiB dump row group for padding DMAs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % n_tiles, mt = (slot / n_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const int n0 = nt * 64;
    const long Mtot = (long)Bimg * H * W;
    const int RS = flat ? W : 8 + 2 * P;

    // ---- tile geometry ----
    long p0 = 0;                                                // flat: first output pixel
    int tn = 0, ty0 = 0, tx0 = 0;                               // 2-D: image, tile origin
    if (flat) p0 = (long)mt * 64;
    else { const int tpr = W >> 3, tpi = tpr * (H >> 3); tn = mt / tpi; const int tt = mt - tn * tpi; ty0 = (tt / tpr) * 8; tx0 = (tt - (tt / tpr) * tpr) * 8; }
    auto out_pixel = [&](int r, int &n, int &y, int &x, long &pix) {
        if (flat) { pix = p0 + r; long q = pix < Mtot ? pix : Mtot - 1; x = (int)(q % W); q /= W; y = (int)(q % H); n = (int)(q / H); }
        else { n = tn; y = ty0 + (r >> 3); x = tx0 + (r & 7); pix = ((long)n * H + y) * W + x; }
    };

    // ---- LDS-DMA sources: row groups of 8 rows = [patch rows | 64 weight rows | TAPROWS tap rows], dealt round-robin to the waves ----
    // Every wave issues exactly GPW LDS-DMA instructions per chunk (GPW = ceil(groups / 4), a template constant) so that a
    // counted s_waitcnt vmcnt(GPW) leaves precisely the next chunk's loads in flight; surplus slots copy a valid row into a dump area.
    const int n_groups = NP / 8 + 8 + TAPROWS / 8;
    constexpr int MAXG = GPW;
    const float *src[MAXG];
    int dst_off[MAXG], src_kind[MAXG];                          // kind 0: activation row (k offset applies), 1: weight row, 2: tap row, 3: padding
    int my_groups = 0;
    for (int g = wave; my_groups < MAXG; g += 4) {
        if (g >= n_groups) {                                    // padding slot
            src[my_groups] = in + (lane & 7) * 4; src_kind[my_groups] = 3; dst_off[my_groups] = -1; ++my_groups; continue;
        }
        const int rr = g * 8 + (lane >> 3);
        const int c = lane & 7;
        int kind, off; const float *sp;
        if (rr < NP) {
            long pi;
            if (flat) { pi = p0 - (long)P * (W + 1) + rr; pi = pi < 0 ? 0 : (pi > Mtot - 1 ? Mtot - 1 : pi); }
            else {
                int py = rr / RS, px = rr - py * RS;
                int gy = ty0 - P + py, gx = tx0 - P + px;
                gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy); gx = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx);
                pi = ((long)tn * H + gy) * W + gx;
            }
            sp = in + pi * C + c * 4; kind = 0; off = rr * BK + c * 4;
        } else if (rr < NP + 64) {
            const int wr = rr - NP;
            int row = n0 + wr; if (row > N - 1) row = N - 1;
            sp = Wt + (long)row * C32 + ((c ^ ((wr >> 1) & 7)) << 2); kind = 1; off = PATCH + wr * BK + c * 4;
        } else {
            int tr = rr - NP - 64; const int trc = tr < KK ? tr : KK - 1;
            sp = wdw + (long)trc * C + c * 4; kind = 2; off = PATCH + WTILE + tr * BK + c * 4;
        }
        src[my_groups] = sp; src_kind[my_groups] = kind; dst_off[my_groups] = off - (lane & 63) * 4 + 0;   // wave-uniform base = off of lane 0
        ++my_groups;
    }
    float *dump = atile + 64 * BK;
    auto issue = [&](int t) {
        float *base = smem + (t % 3) * STAGE;
#pragma unroll
        for (int i = 0; i < MAXG; ++i) {
            int k = t * BK;
            if (src_kind[i] == 3) { fd_glds16(src[i], dump); continue; }
            if (src_kind[i] != 1 && k + (lane & 7) * 4 >= C) k = 0;         // ragged channel tail: finite data, zero pointwise weights
            fd_glds16(src[i] + k, base + dst_off[i]);
        }
    };

    // ---- depthwise work assignment: pixels r0 = tid>>3 and r0+32, channel group c4 = tid&7 ----
    const int c4 = tid & 7, r0 = tid >> 3;
    int pbase[2]; unsigned vmask[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = r0 + 32 * j;
        int n, y, x; long pix;
        out_pixel(r, n, y, x, pix);
        pbase[j] = flat ? r : ((r >> 3) * RS + (r & 7));      // patch index of tap (0,0)-P,-P... see below
        unsigned m = 0;
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int yy = y + ky - P, xx = x + kx - P;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W && pix < Mtot) m |= 1u << (ky * K + kx);
            }
        vmask[j] = m;
    }
    // patch index of tap (ky, kx) for pixel j: pbase[j] + ky*RS + kx   (flat: p - pstart = r + P*(W+1), minus P*(W+1) for the tap origin)

    fd_f32x16 acc;
    {
        const int col = n0 + wn * 32 + (lane & 31);
        const float bv = col < N ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bv;
    }
    const int h = lane >> 5;
    int a_off[4], b_off[4];
    {
        const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a_off[g] = ra * BK + (((2 * g + h) ^ ((ra >> 1) & 7)) << 2);
            b_off[g] = rb * BK + (((2 * g + h) ^ ((rb >> 1) & 7)) << 2);
        }
    }
    const int T = C32 / BK;
    issue(0);
    if (T > 1) issue(1);
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) fd_wait_vmcnt<GPW>(); else fd_wait_vmcnt<0>();
        fd_block_barrier();                                     // chunk t landed for every wave; previous MMA phase is over
        if (t + 2 < T) issue(t + 2);
        const float *stage = smem + (t % 3) * STAGE;
        const float *patch = stage, *wtile = stage + PATCH, *taps = stage + PATCH + WTILE;
        // ---- depthwise phase ----
        {
            const int cg = t * BK + c4 * 4;
            fd_f32x4 b4 = fd_zero4();
            if (cg < C) b4 = fd_ld4(bdw + cg);
            fd_f32x4 a0 = b4, a1 = b4;
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const fd_f32x4 wv = fd_ld4(taps + (ky * K + kx) * BK + c4 * 4);
                    const fd_f32x4 v0 = fd_ld4(patch + (pbase[0] + ky * RS + kx) * BK + c4 * 4);
                    const fd_f32x4 v1 = fd_ld4(patch + (pbase[1] + ky * RS + kx) * BK + c4 * 4);
                    if (vmask[0] >> (ky * K + kx) & 1) a0 += v0 * wv;
                    if (vmask[1] >> (ky * K + kx) & 1) a1 += v1 * wv;
                }
            if (cg >= C) { a0 = fd_zero4(); a1 = fd_zero4(); }
            else { a0 = fd_act4<ACT1>(a0); a1 = fd_act4<ACT1>(a1); }
            fd_st4(atile + r0 * BK + ((c4 ^ ((r0 >> 1) & 7)) << 2), a0);
            fd_st4(atile + (r0 + 32) * BK + ((c4 ^ (((r0 + 32) >> 1) & 7)) << 2), a1);
        }
        fd_block_barrier_lds();                                 // A tile complete (raw barrier: chunk t+1's LDS-DMA stays in flight)
        // ---- MMA phase ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const fd_f32x4 a = fd_ld4(atile + a_off[g]);
            const fd_f32x4 b = fd_ld4(wtile + b_off[g]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
        }
    }
    // ---- epilogue ----
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            int n, y, x; long pix;
            out_pixel(row, n, y, x, pix);
            if (pix < Mtot) out[pix * N + col] = fd_act<ACT2>(acc[r]);
        }
    }
}
