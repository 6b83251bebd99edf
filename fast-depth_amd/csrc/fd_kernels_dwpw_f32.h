// fd_kernels_dwpw_f32.h -- one depthwise-separable unit of the LARGE feature maps as ONE kernel (gfx950, fp32, inference):
//     out = act(W_pw * act(dw_KxK(input) + b_dw) + b_pw)
// (reference: one conv_dw block, imagenet/mobilenet.py:29-38; a decoder stage depthwise(5)+pointwise after the nearest-2x upsample
// and the additive skip, models.py:683-732).  input = the stored tensor (MODE 0) or up2(low) + skip (MODE 2).
//
// Why: on the 112x112 / 56x56 maps both halves of the unit are HBM streams -- the depthwise output is written (up to 103 MB at batch 32)
// only to be read back as the GEMM's A operand by the next launch.  Here it never leaves the CU.  Unlike the small maps (where the
// pointwise GEMM is the cost and its EPILOGUE evaluates the next depthwise layer, fd_kernels_gemm16_f32.h) the channel counts are small
// (C, N <= 128..256): a workgroup owns a pixel tile with ALL N output channels and keeps the whole weight matrix in LDS.
//
// Persistent, wave-specialised workgroup (512 work-items, one per CU; measured: with ordinary workgroups all resident workgroups of a CU
// run load -> depthwise -> MFMA -> store in lock-step and nothing overlaps):
//   waves 0-3  PRODUCERS  stage the next 32-channel chunk of the input patch [(TH-1)S+K][(TW-1)S+K][32] (global -> registers -> LDS;
//                         up2 / skip add / zero padding applied on the way; two register sets: the loads of items i+1 and i+2 are in
//                         flight while item i is computed), then the depthwise taps: work-item = (strip of 4 pixels along x, 4 channels) -> GEMM A tile
//                         [BM][32] (double buffered, 16-byte-chunk XOR swizzle of fd_pw_gemm_f32);
//   waves 4-7  CONSUMERS  wave (wm, wn) owns rows [32 wm, +32) x column tiles wn*NT .. +NT of the tile: 16*NT v_mfma_f32_32x32x2_f32 per
//                         chunk on the A tile the producers finished one step earlier; a finished tile is parked in registers and its
//                         stores (bias was the accumulators' start value: activation, one 128-byte row segment per D register) go out a
//                         quarter at a time between the next item's MFMAs.
// An ITEM is (pixel tile, channel chunk); step i of a workgroup: producers commit + convolve item i while consumers multiply item i-1;
// two workgroup barriers per step (patch committed / A tile complete), raw s_barrier so that the prefetch stays in flight.
// Tiles are dealt so that all tiles of an image run on one XCD (workgroup b -> XCD b % 8 -> images n = b (mod 8)): patch halos hit L2.
// The patch image's row pitch PSTR comes from the host (pick_patch_pitch, fd_api.hip): the smallest pitch for which the strips' ds_read_b128
// are bank-conflict free.
#pragma once
#include "fd_device.h"

#ifdef FD_DWPW_PROBE   // tools/microbench/dwpw.hip: shader-clock totals per wave of one workgroup (segments of a pipeline step)
__device__ unsigned long long fd_dwpw_probe[8][6];
#define FD_DWPW_T(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pr[k] += t_ - t0; t0 = t_; } while (0)
#else
#define FD_DWPW_T(k) do { } while (0)
#endif

// HEAD = 1 (N == 32): the unit's only reader is the network's 32 -> 1 pointwise head (decode_conv6, models.py:698,731), which is evaluated
// on the accumulators -- a 32-lane sum per pixel -- and written 2x2 (a 1x1 conv + BN + ReLU commutes with the nearest upsampling): the
// unit's own 51 MB output is neither written nor read back.
struct fd_dwpw_head { const float *w, *b; float *y; int act, up; };
template <int KS, int S, int MODE, int ACT, int WM, int NT, int NLD, int HEAD = 0, int ABL = 0>   // ABL: ablations of tools/microbench/dwpw.hip (0 in the product)
__global__ void __launch_bounds__(512)
fd_dwpw_f32(const float *__restrict__ in, const float *__restrict__ skip, const float *__restrict__ wdw, const float *__restrict__ bdw,
            const float *__restrict__ Wt, const float *__restrict__ bias, float *__restrict__ out,
            int B, int Hin, int Win, int Ho, int Wo, int C, int K32, int N, int TH, int tw_shift, int tiles_x, int tiles_per_img, int xcd_mode, int PSTR,
            const fd_dwpw_head hd)
{
    static_assert(!HEAD || (NT == 1 && WM == 4), "the head reads all 32 columns of a pixel from one wave");
    constexpr int SW = S == 2 ? 2 : 4, SWS = S == 2 ? 1 : 2;   // output pixels per depthwise strip: the stride-2 units' 64-pixel tile gives every producer work-item a 2-pixel strip
    constexpr int P = KS / 2, NIN = (SW - 1) * S + KS, WN = 4 / WM, BM = 32 * WM, KK = KS * KS;
    static_assert(WM * WN == 4, "four consumer waves");
    FD_DYN_SMEM(smem_raw);
    const int nchunks = C >> 5, cshift = 31 - __builtin_clz(nchunks);      // C / 32 is a power of two (checked by the plan)
    float *s_in = reinterpret_cast<float *>(smem_raw);       // [NLD*32][PSTR]        input patch of the current item
    float *s_a = s_in + NLD * 32 * PSTR;                     // [2][BM][32]           GEMM A tiles (swizzled)
    float *s_wb = s_a + 2 * BM * 32;                         // [nchunks][N][32]      pointwise weights, chunk-major (swizzled)
    float *s_t = s_wb + N * C;                               // [nchunks][KK][32]     depthwise taps
    float *s_bd = s_t + KK * C;                              // [C]                   depthwise bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TW = 1 << tw_shift;
    const int TH_in = (TH - 1) * S + KS, TW_in = (TW - 1) * S + KS, npx = TH_in * TW_in;

    // ---- once per workgroup: weights, taps and depthwise bias become LDS-resident ----
    {
        const int q4 = C >> 2, qs = cshift + 3;               // 16-byte pieces per row
        for (int idx = tid; idx < N * q4; idx += 512) {
            const int row = idx >> qs, rem = idx & (q4 - 1), chunk = rem >> 3, ch = rem & 7;
            fd_st4(s_wb + chunk * N * 32 + row * 32 + ((ch ^ ((row >> 1) & 7)) << 2), fd_ld4(Wt + (long)row * K32 + rem * 4));
        }
        for (int idx = tid; idx < KK * q4; idx += 512) {
            const int t = idx >> qs, rem = idx & (q4 - 1), chunk = rem >> 3, ch = rem & 7;
            fd_st4(s_t + chunk * KK * 32 + t * 32 + ch * 4, fd_ld4(wdw + (long)t * C + rem * 4));
        }
        for (int idx = tid; idx < q4; idx += 512) fd_st4(s_bd + idx * 4, fd_ld4(bdw + idx * 4));
    }
    __syncthreads();

    // ---- this workgroup's tiles: q = first, first + step, ... < ntl ----
    int first, step, ntl, img0, img_step;
    if (xcd_mode) {
        const int xcd = blockIdx.x & 7;
        first = blockIdx.x >> 3; step = gridDim.x >> 3; img0 = xcd; img_step = 8;
        ntl = ((B - xcd + 7) >> 3) * tiles_per_img;
    } else { first = blockIdx.x; step = gridDim.x; img0 = 0; img_step = 1; ntl = B * tiles_per_img; }
    const int my_tiles = first < ntl ? (ntl - first + step - 1) / step : 0;
    const int n_items = my_tiles << cshift;
    const float rcp_tpi = 1.0f / (float)tiles_per_img, rcp_tx = 1.0f / (float)tiles_x;
    auto tile_origin = [&](int k, int &n, int &oy0, int &ox0) __attribute__((always_inline)) {
        const int q = FD_UNIFORM(first + k * step);
        // exact for the small integers involved (q < 2^20): (q + 0.5) / d is never within rounding distance of an integer
        const int im = (int)(((float)q + 0.5f) * rcp_tpi), t = q - im * tiles_per_img;
        const int ty = (int)(((float)t + 0.5f) * rcp_tx), tx = t - ty * tiles_x;
        n = img0 + im * img_step; oy0 = ty * TH; ox0 = tx << tw_shift;
    };

    const bool producer = wave < 4;
    // ---- producer state: NSET register sets of patch loads in flight (items i .. i+NSET-1) ----
    constexpr int NSET = 2;
    const int c4 = tid & 7, pt = (tid >> 3) & 31;
    int piy[NLD], pix[NLD], off_a[NLD], off_b[NLD];
    unsigned okbits = 0, okb[NSET];
    fd_f32x4 pv[NSET][NLD], ps[NSET][NLD];
    {
        const float rcp = 1.0f / (float)TW_in;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int px = pt + 32 * u;
            piy[u] = (int)(((float)px + 0.5f) * rcp); pix[u] = px - piy[u] * TW_in;
            if (px >= npx) piy[u] = 1 << 20;                // never inside an image
            off_a[u] = 0; off_b[u] = 0;
#pragma unroll
            for (int st = 0; st < NSET; ++st) { pv[st][u] = fd_zero4(); ps[st][u] = fd_zero4(); }
        }
#pragma unroll
        for (int st = 0; st < NSET; ++st) okb[st] = 0;
    }
    auto setup_tile = [&](int k) __attribute__((always_inline)) {
        int n, oy0, ox0;
        tile_origin(k, n, oy0, ox0);
        const int iy0 = oy0 * S - P, ix0 = ox0 * S - P;
        okbits = 0;
        // branch-free staging: every lane loads unconditionally -- pixels outside the image read the image's first pixel instead (any
        // valid address) and are replaced by the zero padding at commit
        const int img_a = MODE == 0 ? n * Hin * Win * C + c4 * 4 : n * (Hin >> 1) * (Win >> 1) * C + c4 * 4;
        const int img_b = n * Hin * Win * C + c4 * 4;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int gy = iy0 + piy[u], gx = ix0 + pix[u];
            const bool ok = (unsigned)gy < (unsigned)Hin && (unsigned)gx < (unsigned)Win;
            if (ok) okbits |= 1u << u;
            if (MODE == 0) {
                off_a[u] = img_a + (ok ? (gy * Win + gx) * C : 0);
            } else {
                off_a[u] = img_a + (ok ? ((gy >> 1) * (Win >> 1) + (gx >> 1)) * C : 0);
                off_b[u] = img_b + (ok ? (gy * Win + gx) * C : 0);
            }
        }
    };
    int issued = 0;                                           // next item whose loads go out (items are issued in order)
    auto issue_next = [&](auto SET) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
        const int cn = issued & (nchunks - 1);
        if (cn == 0) setup_tile(issued >> cshift);
        okb[st] = okbits;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            if (ABL == 1) continue;
            pv[st][u] = fd_ld4(in + off_a[u] + cn * 32);
            if (MODE == 2) ps[st][u] = fd_ld4(skip + off_b[u] + cn * 32);
        }
        ++issued;
    };
    auto commit = [&](auto SET) __attribute__((always_inline)) {
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            fd_f32x4 v = pv[st][u];
            if (MODE == 2) v += ps[st][u];
            if (!((okb[st] >> u) & 1u)) v = fd_zero4();
            fd_st4(s_in + (pt + 32 * u) * PSTR + c4 * 4, v);
        }
    };
    const int TWS = TW >> SWS, nstrips = TH * TWS;
    const int soy = pt >> (tw_shift - SWS), sox = (pt & (TWS - 1)) * SW;
    // 3x3 units with a single channel chunk (C = 32) keep their nine taps and the bias in registers for the workgroup's life
    constexpr bool TAPREG = KS == 3;
    fd_f32x4 tapr[TAPREG ? KK : 1], biasr = fd_zero4();
    const bool taps_in_regs = TAPREG && nchunks == 1;
    if (taps_in_regs) {
#pragma unroll
        for (int t = 0; t < (TAPREG ? KK : 1); ++t) tapr[t] = fd_ld4(s_t + t * 32 + c4 * 4);
        biasr = fd_ld4(s_bd + c4 * 4);
    }
    auto depthwise = [&](int item) __attribute__((always_inline)) {
        if (pt >= nstrips) return;
        const int chunk = item & (nchunks - 1);
        const float *taps = s_t + chunk * KK * 32 + c4 * 4;
        if (taps_in_regs) {
            fd_f32x4 d[SW];
#pragma unroll
            for (int j = 0; j < SW; ++j) d[j] = biasr;
#pragma unroll
            for (int ky = 0; ky < (TAPREG ? KS : 0); ++ky) {
                const float *row = s_in + ((soy * S + ky) * TW_in + sox * S) * PSTR + c4 * 4;
                fd_f32x4 r[NIN];
#pragma unroll
                for (int i = 0; i < NIN; ++i) r[i] = fd_ld4(row + i * PSTR);
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int j = 0; j < SW; ++j) d[j] += r[j * S + kx] * tapr[TAPREG ? ky * KS + kx : 0];
            }
            float *A = s_a + (item & 1) * BM * 32;
#pragma unroll
            for (int j = 0; j < SW; ++j) {
                const int arow = (soy << tw_shift) + sox + j;
                fd_st4(A + arow * 32 + ((c4 ^ ((arow >> 1) & 7)) << 2), fd_act4<ACT>(d[j]));
            }
            return;
        }
        const fd_f32x4 b4 = fd_ld4(s_bd + chunk * 32 + c4 * 4);
        fd_f32x4 d[SW];
#pragma unroll
        for (int j = 0; j < SW; ++j) d[j] = b4;
#pragma unroll 1
        for (int ky = 0; ky < (ABL == 2 ? 0 : KS); ++ky) {
            const float *row = s_in + ((soy * S + ky) * TW_in + sox * S) * PSTR + c4 * 4;
            fd_f32x4 r[NIN];
#pragma unroll
            for (int i = 0; i < NIN; ++i) r[i] = fd_ld4(row + i * PSTR);
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const fd_f32x4 w = fd_ld4(taps + (ky * KS + kx) * 32);
#pragma unroll
                for (int j = 0; j < SW; ++j) d[j] += r[j * S + kx] * w;
            }
        }
        float *A = s_a + (item & 1) * BM * 32;
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            const int arow = (soy << tw_shift) + sox + j;
            fd_st4(A + arow * 32 + ((c4 ^ ((arow >> 1) & 7)) << 2), fd_act4<ACT>(d[j]));
        }
    };

    // ---- consumer state ----
    const int cw = wave & 3, wm = cw / WN, wn = cw - wm * WN;
    const int l31 = lane & 31, h = lane >> 5, swz = (l31 >> 1) & 7;
    const int a_row = (wm * 32 + l31) * 32, b_row = (wn * NT * 32 + l31) * 32;
    fd_f32x16 acc[NT];
    float bcol[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = (wn * NT + j) * 32 + l31;
        bcol[j] = col < N ? bias[col] : 0.0f;              // accumulators start at the folded-BN bias of their column
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = bcol[j];
    }
    // the finished tile is parked in `hold` and its stores go out a quarter at a time between the MFMAs of the NEXT item (pushed all at
    // once they fill the CU's write queue and the wave stalls instead of multiplying: measured 4.7 kclk per tile).  D register r of lane
    // l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31 of the wave's 32x32 tile: the four registers of a group (r>>2) are four
    // consecutive pixels of one tile row (TW is a multiple of 4), so a group is ONE byte offset + three immediates (N*4 bytes apart).
    fd_f32x16 hold[NT];
    unsigned h_off[4];                                       // byte offset of pixel (group g, register 0), column wn*NT*32 + l31
    bool pending = false, h_full = false;
    int h_n = 0, h_oy0 = 0, h_ox0 = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) h_off[g] = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) hold[j] = acc[j];
    float head_w = 0.0f, head_b = 0.0f, head_v = 0.0f;
    unsigned head_off = 0;
    bool head_ok = false;
    if (HEAD) { head_w = hd.w[l31]; head_b = hd.b[0]; }
    auto store_part = [&](auto PART) __attribute__((always_inline)) {
        constexpr int g = decltype(PART)::value;
        if (!pending) return;
        if (HEAD) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v = fd_act<ACT>(hold[0][g * 4 + rr]) * head_w;
#ifdef FD_EMU
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
#else
                // 32-lane sum on the VALU's data-parallel-primitive path (no LDS crossbar traffic next to the producers' tap reads): quad
                // swaps, half-row and row mirrors give every lane its 16-lane row sum; row 1 / 3 then add lane 15 of row 0 / 2
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1, 3
#endif
                if (l31 == 16 + g * 4 + rr) head_v = v;      // lane 16 + r of each 32-lane half (rows 1 / 3 hold the totals) keeps the pixel of D register r
            }
            if (g == 3) {
                // lane (16 + r, half) holds row (r&3) + 8*(r>>2) + 4*half; lane i < 32 fetches row i, so that 16 lanes cover one tile row segment
                // (128 contiguous bytes per output row instead of 8-byte fragments)
                const int src = 16 + ((lane & 3) | (((lane >> 3) & 3) << 2)) + (((lane >> 2) & 1) << 5);
                head_v = __shfl(head_v, src);
            }
            if (g == 3 && head_ok) {
                float v = head_v + head_b;
                if (hd.act >= 1) v = fmaxf(v, 0.0f);
                if (hd.act == 2) v = fminf(v, 6.0f);
                if (hd.up) {
                    const fd_f32x2 vv = {v, v};
                    *reinterpret_cast<fd_f32x2 *>(hd.y + head_off) = vv;
                    *reinterpret_cast<fd_f32x2 *>(hd.y + head_off + 2 * Wo) = vv;
                } else hd.y[head_off] = v;
            }
            return;
        }
        char *base = reinterpret_cast<char *>(out);
        if (h_full) {                                         // whole tile inside the map (workgroup-uniform): no predicates
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float v = fd_act<ACT>(hold[j][g * 4 + rr]);
                    if (ABL != 4 || v == 123.456f) *reinterpret_cast<float *>(base + h_off[g] + (unsigned)(rr * N + j * 32) * 4u) = v;
                }
        } else {
            const int row = wm * 32 + 8 * g + 4 * h;
            const int oy = row >> tw_shift, ox = row & (TW - 1);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (oy < TH && h_oy0 + oy < Ho && h_ox0 + ox + rr < Wo && (wn * NT + j) * 32 + l31 < N)
                        *reinterpret_cast<float *>(base + h_off[g] + (unsigned)(rr * N + j * 32) * 4u) = fd_act<ACT>(hold[j][g * 4 + rr]);
        }
    };
    auto mma_half = [&](int item, auto HALF) __attribute__((always_inline)) {
        constexpr int half = decltype(HALF)::value;
        const float *A = s_a + (item & 1) * BM * 32 + a_row;
        const float *Wc = s_wb + (item & (nchunks - 1)) * N * 32 + b_row;
        // both 16-byte fragment groups of this half are requested up front: the second group's LDS latency hides under the first group's MFMAs
        fd_f32x4 a[2], b[2][NT];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int og = ((2 * (2 * half + gg) + h) ^ swz) << 2;
            a[gg] = fd_ld4(A + og);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[gg][j] = fd_ld4(Wc + j * 1024 + og);
        }
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            if (ABL != 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gg][q], b[gg][j][q], acc[j], 0, 0, 0);
            }
            if (gg == 0) store_part(fd_int<2 * half>{}); else store_part(fd_int<2 * half + 1>{});
        }
        if (half == 1) pending = false;
    };
    auto retire = [&](int k) __attribute__((always_inline)) {   // the tile's last chunk has been multiplied
        tile_origin(k, h_n, h_oy0, h_ox0);
        h_full = h_oy0 + TH <= Ho && h_ox0 + TW <= Wo && (TH << tw_shift) == BM;
        if (HEAD) {                                           // lane i (< 32) writes the pixel of tile row i of this wave
            const int row = wm * 32 + l31;
            const int oy = row >> tw_shift, ox = row & (TW - 1), gy = h_oy0 + oy, gx = h_ox0 + ox;
            head_ok = lane < 32 && oy < TH && gy < Ho && gx < Wo;
            head_off = hd.up ? (unsigned)((h_n * 2 * Ho + 2 * gy) * 2 * Wo + 2 * gx) : (unsigned)((h_n * Ho + gy) * Wo + gx);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = wm * 32 + 8 * g + 4 * h;
            int gy = h_oy0 + (row >> tw_shift), gx = h_ox0 + (row & (TW - 1));
            gy = gy < Ho ? gy : Ho - 1; gx = gx < Wo ? gx : Wo - 1;          // (clamped: invalid pixels are predicated off)
            h_off[g] = (unsigned)(((h_n * Ho + gy) * Wo + gx) * N + wn * NT * 32 + l31) * 4u;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            hold[j] = acc[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = bcol[j];
        }
        pending = true;
    };

    // ---- the pipeline: step i = producers commit + convolve item i | consumers multiply item i-1.  Two loops with the same barrier
    // sequence (one per role: the register allocator then sees the two roles' state as never live together) ----
#ifdef FD_DWPW_PROBE
    unsigned long long pr[6] = {0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime();
#endif
    if (producer) {
        if (issued < n_items) issue_next(fd_int<0>{});
        if (NSET == 2 && issued < n_items) issue_next(fd_int<NSET - 1>{});
        auto pipe_step = [&](auto SET, int i) __attribute__((always_inline)) {
            if (i < n_items) {
                commit(SET);
                FD_DWPW_T(4);
                if (issued < n_items) issue_next(SET);       // item i + NSET: in flight during the next NSET steps
            }
            FD_DWPW_T(0);
            fd_block_barrier_lds();                           // patch of item i committed
            FD_DWPW_T(1);
            if (i < n_items) depthwise(i);
            FD_DWPW_T(2);
            fd_block_barrier_lds();                           // A tile of item i complete; patch buffer free
            FD_DWPW_T(3);
        };
        for (int i = 0; i <= n_items; i += 2) {
            pipe_step(fd_int<0>{}, i);
            if (i + 1 <= n_items) pipe_step(fd_int<NSET - 1>{}, i + 1);
        }
    } else {
        for (int i = 0; i <= n_items; ++i) {
            if (i > 0) mma_half(i - 1, fd_int<0>{});
            FD_DWPW_T(0);
            fd_block_barrier_lds();
            FD_DWPW_T(1);
            if (i > 0) {
                mma_half(i - 1, fd_int<1>{});
                FD_DWPW_T(4);
                if (((i - 1) & (nchunks - 1)) == nchunks - 1) retire((i - 1) >> cshift);
            }
            FD_DWPW_T(2);
            fd_block_barrier_lds();                           // A tile of item i-1 free
            FD_DWPW_T(3);
        }
        store_part(fd_int<0>{}); store_part(fd_int<1>{}); store_part(fd_int<2>{}); store_part(fd_int<3>{});   // the last tile
    }
#ifdef FD_DWPW_PROBE
    if (blockIdx.x == 8 && lane == 0)
        for (int k = 0; k < 6; ++k) fd_dwpw_probe[wave][k] = pr[k];
#endif
}
