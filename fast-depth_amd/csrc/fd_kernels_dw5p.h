// fd_kernels_dw5p.h -- depthwise 5x5 (stride 1) + BN + activation on  up2(low) + skip  for the 16-bit plans: the row-walking pixel-pair kernel (round 6).
//
// Replaces fd_dwconv<T, 5, 1, 2, ACT, {4, 8}> on decode_conv3 / 4 / 5 (reference models.py:61-68 behind the nearest x2 of models.py:723 and the skip
// additions of models.py:724-729).  What the LDS-tiled kernel paid for was its instruction count, not its bytes (PMC: 49 VALU lane-instructions per
// output against 12.5 packed tap FMAs): an 8 x 16 tile stages 1.875 x its outputs, every staged element is converted again on each of its
// 5 LDS reads, and the taps run on converted fp32 values.  Here
//   * every WAVE is on its own -- no LDS, no barrier.  A lane owns 2 adjacent channels x 4 adjacent output columns and walks down a band of rows with
//     the 5-row input window AND the 30 tap words of its channels in registers; an input row is loaded, summed and packed once per band and lane;
//   * the window holds PIXEL PAIRS: one 32-bit word = the same channel of pixels (2j, 2j+1) in the storage type.  Both pixels of a pair have the same
//     low-resolution parent (the parent is fetched once per pair and row pair);
//   * the taps run on v_dot2_f32_{f16,bf16}: output pixel x, filter row ky is 3 dot2 on the pairs covering pixels x-2 .. x+2 against the 16-bit tap pairs
//     (w0,w1)(w2,w3)(w4,0) for even x and (0,w0)(w1,w2)(w3,w4) for odd x -- 15 instructions per output and channel, fp32 accumulation from the folded
//     bias, no conversions.  The dot2 are issued in source order (8 independent chains; fd_dot2_acc);
//   * memory access is raw-buffer (fd_buf_ld32 / fd_buf_st32): 9 lane offsets for the whole band, row / pixel-parity offsets scalar, the horizontal zero
//     padding done by the hardware's range check; the next two rows' 20 loads are in flight under the 240 dot2 of the current two.
// Measured (MI355X, B = 32, fp16; tools/microbench/dw5pairs.hip, profiles/r06/dw5_microbench_*.txt): decode_conv5.0 42.6 -> 33 us, decode_conv4.0 26.4 -> 20,
// decode_conv3.0 15.3 -> 13; two structurally different forms of the same arithmetic (a 256-thread workgroup with VALU-staged pairs in LDS; waves fed by
// LDS-DMA: tools/microbench/dw5_variants.h) land within 3 % of it.  What bounds all three: on gfx950 every VALU instruction except plain fp32
// FMA / ADD / MUL issues in 4 cycles per wave (v_dot2, v_pk_fma_f32, v_perm, conversions, v_max: profiles/r06/valu_issue_rates_*.txt), so 15 dot2 + ~6
// others per output and channel is ~85 cycles -- 24 us of pure issue time for decode_conv5.0 against 21 us of HBM time for its bytes, two near-critical
// resources that overlap imperfectly.
// Numerics: the summed input is rounded to the storage type (as fd_dwconv's 8-channel form did) and the BatchNorm-folded taps are rounded to the storage
// type (the 16-bit plans' pointwise weights already are); accumulation, bias and activation are fp32.
#pragma once
#include "fd_device.h"

#ifdef FD_EMU
#define FD_INLINE_LAMBDA
#else
#define FD_INLINE_LAMBDA __attribute__((always_inline))     // (the tap lambdas hold volatile asm: the inliner's cost model would leave them as calls, with the register windows in scratch)
#endif

// folded fp32 taps [25][C] (fd_pack_fold, tap-major) -> tap pairs [5 filter rows][6][C] in the storage type: (w0,w1) (w2,w3) (w4,0) (0,w0) (w1,w2) (w3,w4), low half first
template <typename T>
__global__ void __launch_bounds__(256)
fd_pack_dw5_pairs(const float *__restrict__ wf, unsigned *__restrict__ wpk, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 30 * C) return;
    const int c = i % C, q = i / C, ky = q / 6, k = q - ky * 6;
    const float *w = wf + (long)(ky * 5) * C + c;
    const float w0 = w[0], w1 = w[C], w2 = w[2 * C], w3 = w[3 * C], w4 = w[4 * C];
    float a, b;
    switch (k) {
    case 0: a = w0; b = w1; break;
    case 1: a = w2; b = w3; break;
    case 2: a = w4; b = 0.f; break;
    case 3: a = 0.f; b = w0; break;
    case 4: a = w1; b = w2; break;
    default: a = w3; b = w4; break;
    }
    wpk[i] = fd_pack2(T{}, a, b);
}

typedef unsigned fd_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned fd_u32x2 __attribute__((ext_vector_type(2)));

// One input row of a strip as pixel pairs.  raw[i] = (channel 2l, channel 2l + 1) of pixel i of the strip's 8, par[p] = the same two channels of the
// parent of pixels (2p, 2p + 1); pair[p][ch] = (pixel 2p, pixel 2p + 1) of channel ch, each  skip + parent  rounded to the storage type.
//   fp16: two v_perm_b32 build the skip pair and the parent pair, one v_pk_add_f16 adds them (the fp16 sum of two fp16 values is the correctly rounded
//         fp32 sum);  bf16 (no packed add): both operands widen to fp32 with one shift / mask each, two fp32 additions, and v_cvt_pk_bf16_f32 rounds AND
//         packs (even pixel, odd pixel) -- the conversion builds the pair, no permute at all.
__device__ __forceinline__ void fd_dw5_make_pairs(fd_half, const unsigned (&raw)[8], const unsigned (&par)[4], unsigned (&pair)[4][2])
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        pair[p][0] = fd_pair_sum(fd_half{}, fd_perm(raw[2 * p + 1], raw[2 * p], FD_PERM_LO), fd_perm(par[p], par[p], FD_PERM_LO));
        pair[p][1] = fd_pair_sum(fd_half{}, fd_perm(raw[2 * p + 1], raw[2 * p], FD_PERM_HI), fd_perm(par[p], par[p], FD_PERM_HI));
    }
}
__device__ __forceinline__ void fd_dw5_make_pairs(fd_bf16, const unsigned (&raw)[8], const unsigned (&par)[4], unsigned (&pair)[4][2])
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float l0 = __builtin_bit_cast(float, par[p] << 16), l1 = __builtin_bit_cast(float, par[p] & 0xffff0000u);
        const unsigned e = raw[2 * p], o = raw[2 * p + 1];
        pair[p][0] = fd_f32x2_to_bf16x2(__builtin_bit_cast(float, e << 16) + l0, __builtin_bit_cast(float, o << 16) + l0);
        pair[p][1] = fd_f32x2_to_bf16x2(__builtin_bit_cast(float, e & 0xffff0000u) + l1, __builtin_bit_cast(float, o & 0xffff0000u) + l1);
    }
}
// two activated fp32 outputs (adjacent channels) -> one word of the storage type.  ReLU commutes with the rounding and is one packed signed-integer
// maximum on the packed word afterwards (sign-magnitude: a negative value is a negative int16, +-0 and positives are unchanged): v_pk_max_i16 for two
// values instead of a 4-cycle v_max_f32 for each.
template <typename T, int ACT>
__device__ __forceinline__ unsigned fd_dw5_pack_act(float a, float b)
{
    if (ACT == FD_ACT_RELU_) {
        typedef short fd_s16x2 __attribute__((ext_vector_type(2)));
        const fd_s16x2 v = __builtin_bit_cast(fd_s16x2, fd_pack2(T{}, a, b)), z = {0, 0};
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, z));
    }
    return fd_pack2(T{}, fd_act_raw<ACT>(a), fd_act_raw<ACT>(b));
}

#ifndef FD_DW5R_BLOCK
#define FD_DW5R_BLOCK 256        // a workgroup is FD_DW5R_BLOCK / 64 consecutive (band, strip group) items: its waves share nothing (64 ... 256 measured equal)
#endif
#ifndef FD_DW5R_WAVES
#define FD_DW5R_WAVES 3          // waves per SIMD the register allocation is held to (<= 168 VGPRs; 4 spills and runs 2x longer)
#endif
#ifdef FD_EMU
#define FD_DW5R_ATTR
#else
#define FD_DW5R_ATTR __attribute__((amdgpu_waves_per_eu(FD_DW5R_WAVES, FD_DW5R_WAVES)))
#endif

//   low  [B][H/2][W/2][C], skip [B][H][W][C], out [B][H][W][C] (NHWC, storage type T); wpk: fd_pack_dw5_pairs; bias [C] fp32 (folded)
//   CL = channel lanes per strip: 32 (a wave = 2 strips x 64 channels) or 64 (1 strip x 128 channels: maps whose strip count is odd)
//   cbs: channels per block (multiple of 8, <= 2 * CL); groups_x = ceil(W / 4 / (64 / CL)); bh: output rows per band (even)
// grid (ceil(groups_x * bands / (FD_DW5R_BLOCK / 64)), channel blocks, images) through fd_xcd_image_map; W % 4 == 0, H even, C % 8 == 0, image bytes < 2^31.
template <typename T, int ACT, int CL>
__global__ void __launch_bounds__(FD_DW5R_BLOCK) FD_DW5R_ATTR
fd_dw5_rows(const T *__restrict__ low, const T *__restrict__ skip, const unsigned *__restrict__ wpk, const float *__restrict__ bias,
            T *__restrict__ out, int H, int W, int C, int cbs, int groups_x, int bh)
{
    constexpr int SPW = 64 / CL;                            // strips per wave
    const fd_blk3 blk = fd_xcd_image_map();                  // all waves of an image on one XCD: the x-overlap of neighbouring strips and the band halos are L2 hits
    const int item = blk.x * (FD_DW5R_BLOCK / 64) + FD_UNIFORM((int)(threadIdx.x >> 6));
    const int band = item / groups_x, sg = item - band * groups_x;
    if (band * bh >= H) return;
    const int c0 = blk.y * cbs, cend = c0 + cbs < C ? c0 + cbs : C, n = blk.z;
    const int y0 = band * bh, y1 = y0 + bh < H ? y0 + bh : H;
    const int lane = threadIdx.x & 63, l = lane % CL, xs = 4 * (sg * SPW + lane / CL);
    const int c = c0 + 2 * l;
    if (!(c < cend && xs < W)) return;                      // (no collective operation below: idle lanes simply leave)
    const int Hs = H >> 1, Ws = W >> 1;

    unsigned w[5][6][2];
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const fd_u32x2 v = *reinterpret_cast<const fd_u32x2 *>(wpk + (long)(ky * 6 + k) * C + c);
            w[ky][k][0] = v.x; w[ky][k][1] = v.y;
        }
    const float b0 = bias[c], b1 = bias[c + 1];

    // per-lane BYTE offsets inside an image row; everything that changes per row or pixel parity is a scalar offset of the buffer access.  A pixel
    // pair left or right of the image gets the out-of-range offset: its loads return 0 (skip and parent alike, so the staged sum is the zero padding)
    unsigned so[4], lo_[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int x = xs - 2 + 2 * p;
        const bool in = x >= 0 && x < W;
        so[p] = in ? (fd_mul24((unsigned)x, (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
        lo_[p] = in ? (fd_mul24((unsigned)(x >> 1), (unsigned)C) + (unsigned)c) * 2u : FD_BUF_OOB;
    }
    const unsigned rowb = fd_mul24((unsigned)W, (unsigned)C) * 2u, pxb = (unsigned)C * 2u;   // bytes per image row / per pixel
    const fd_bufrsrc r_skip = fd_make_rsrc(skip + (long)n * H * W * C, (unsigned)H * rowb);
    const fd_bufrsrc r_low = fd_make_rsrc(low + (long)n * Hs * Ws * C, (unsigned)Hs * (rowb >> 1));
    const fd_bufrsrc r_out = fd_make_rsrc(out + (long)n * H * W * C, (unsigned)H * rowb);
    const unsigned oo = (fd_mul24((unsigned)xs, (unsigned)C) + (unsigned)c) * 2u;

    unsigned ns[2][8], nl[4];                               // loads in flight: the next step's two skip rows (8 pixels each) and its parent row (4 pixels)
    bool nv = false, nv_b = false;                          // (wave-uniform) whether those rows are inside the image
    // step `it` consumes rows r = y0 - 2 + 2 * it (even) and r + 1, which share the parent row r / 2 (r even, H even: both are inside the image or
    // neither).  Its loads are issued in two halves, each as soon as the registers it lands in have been consumed: the first row right after step
    // it - 1 has converted ITS first row, the second row and the parent row after it has converted its second row
    auto issue_a = [&](int it) FD_INLINE_LAMBDA {
        const int r = y0 - 2 + 2 * it;
        nv = r >= 0 && r < H;
        if (!nv) return;
        const unsigned ro = (unsigned)r * rowb;
#pragma unroll
        for (int p = 0; p < 4; ++p) { ns[0][2 * p] = fd_buf_ld32(r_skip, so[p], ro); ns[0][2 * p + 1] = fd_buf_ld32(r_skip, so[p], ro + pxb); }
    };
    auto issue_b = [&](int it) FD_INLINE_LAMBDA {
        const int r = y0 - 2 + 2 * it;
        nv_b = r >= 0 && r < H;
        if (!nv_b) return;
        const unsigned ro = (unsigned)(r + 1) * rowb, rl = (unsigned)(r >> 1) * (rowb >> 1);
#pragma unroll
        for (int p = 0; p < 4; ++p) { ns[1][2 * p] = fd_buf_ld32(r_skip, so[p], ro); ns[1][2 * p + 1] = fd_buf_ld32(r_skip, so[p], ro + pxb); nl[p] = fd_buf_ld32(r_low, lo_[p], rl); }
    };

    unsigned win[5][4][2];                                  // input row (y0 - 2 + r) lives in win[r % 5]: [pair of the strip][channel]
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int p = 0; p < 4; ++p) { win[a][p][0] = 0u; win[a][p][1] = 0u; }

    auto convert = [&](auto SLOT, const unsigned (&raw)[8], bool rv) FD_INLINE_LAMBDA {
        constexpr int slot = decltype(SLOT)::value;
        if (rv) fd_dw5_make_pairs(T{}, raw, nl, win[slot]);
        else {
#pragma unroll
            for (int p = 0; p < 4; ++p) { win[slot][p][0] = 0u; win[slot][p][1] = 0u; }
        }
    };
    // one output row from the window; SB = slot of the row under filter row 0.  The dot2 are issued in source order: 8 independent chains interleaved
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
                        const float bb = ch ? b1 : b0;
                        acc[0][ch] = fd_dot2_first(T{}, R[k][ch], w[ky][k][ch], bb);
                        acc[1][ch] = fd_dot2_first(T{}, R[k][ch], w[ky][3 + k][ch], bb);
                        acc[2][ch] = fd_dot2_first(T{}, R[k + 1][ch], w[ky][k][ch], bb);
                        acc[3][ch] = fd_dot2_first(T{}, R[k + 1][ch], w[ky][3 + k][ch], bb);
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
            for (int j = 0; j < 4; ++j) fd_buf_st32(r_out, oo, (unsigned)y * rowb + (unsigned)j * pxb, fd_dw5_pack_act<T, ACT>(acc[j][0], acc[j][1]));
        }
    };

    const int n_it = (y1 - y0 + 4) >> 1;                    // steps of two input rows: rows y0 - 2 ... y1 + 1
    issue_a(0); issue_b(0);
    // PH = step number mod 5: the window slots of a step are compile-time constants (the walk is unrolled over one period of the 5-row window)
    auto step = [&](auto PH, int it) FD_INLINE_LAMBDA {
        constexpr int ph = decltype(PH)::value;
        const bool more = it + 1 < n_it;
        convert(fd_int<(2 * ph) % 5>{}, ns[0], nv);
        if (more) issue_a(it + 1);                           // the next step's loads fly under this step's 240 dot2
        if (it >= 2) out_row(fd_int<(2 * ph + 1) % 5>{}, y0 - 4 + 2 * it);
        convert(fd_int<(2 * ph + 1) % 5>{}, ns[1], nv_b);
        if (more) issue_b(it + 1);
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
