// fd_kernels_dw5p.h -- depthwise 5x5 (stride 1) + BN + activation on  up2(low) + skip  for the 16-bit plans: the "pixel-pair" kernel (round 6).
//
// Replaces fd_dwconv<T, 5, 1, 2, ACT, {4, 8}> on decode_conv3 / 4 / 5 (reference models.py:61-68 behind the nearest x2 of models.py:723 and the skip
// additions of models.py:724-729).  What the LDS-tiled kernel paid for was its instruction count, not its bytes (PMC: 49 VALU lane-instructions per
// output against 12.5 packed tap FMAs): an 8 x 16 tile stages 1.875 x its outputs, every staged element is converted again on each
// of its 5 LDS reads, and the taps themselves run on converted fp32 values.  This kernel changes all three:
//   * a workgroup owns (image, <= 64 channels, <= 28 output columns) and WALKS DOWN a band of rows.  LDS is only a pass-through that hands every
//     new input row to the work-items that need it (two rows per step, double buffered, one barrier per step); the 5-row window a work-item needs
//     lives in its registers, so an input element is staged once per band (x-halo 32/28, y-halo 4 rows per band) and read from LDS once or twice;
//   * the staged value is a PIXEL PAIR: one 32-bit word = the same channel of pixels (2j, 2j+1) in the storage type.  Both pixels of a pair have
//     the same low-resolution parent, so  up2(low) + skip  is one packed addition per word (v_pk_add_f16; bf16: through fp32) against the
//     parent's value broadcast to both halves -- the parent is loaded once per pair, not once per staged element;
//   * the taps run on v_dot2_f32_{f16,bf16}: output pixel x, filter row ky is 3 dot2 on the pairs that cover pixels x-2 .. x+2 against the
//     16-bit tap pairs (w0,w1)(w2,w3)(w4,0) for even x and (0,w0)(w1,w2)(w3,w4) for odd x -- 15 instructions per output and channel, fp32
//     accumulation, no conversions.  The 30 tap words of a channel stay in registers for the whole band.
// Work-item = 2 adjacent channels x 4 adjacent output columns; 256 work-items = 32 channel lanes x 8 strips.  Numerics: the summed input is rounded
// to the storage type (as fd_dwconv's 8-channel form did) and the BatchNorm-folded taps are rounded to the storage type (new: the 16-bit plans'
// pointwise weights already are); accumulation, bias and activation are fp32.
#pragma once
#include "fd_device.h"

#ifndef FD_DW5P_FENCE
#define FD_DW5P_FENCE() FD_SCHED_FENCE()
#endif
#ifndef FD_DW5P_ABL
#define FD_DW5P_ABL 0            // tools/microbench/dw5pairs.hip only: 1 = no global loads, 2 = no stores, 4 = no taps
#endif
#define FD_DW5P_NP 16            // pairs per staged row: (28 output columns + 4 halo pixels) / 2
#define FD_DW5P_ROW (FD_DW5P_NP * 64)   // dwords per staged row: [pair][64 channel slots]
#define FD_DW5P_LDS (2 * 2 * FD_DW5P_ROW * 4 + 4 * 1024)   // two buffers x two rows + one 1 KiB output exchange tile per wave

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

// grid (tiles_x * bands, channel blocks, images) through fd_xcd_image_map; block 256; dynamic LDS FD_DW5P_LDS.
//   low  [B][H/2][W/2][C], skip [B][H][W][C], out [B][H][W][C] (NHWC, storage type T); wpk: fd_pack_dw5_pairs; bias [C] fp32 (folded)
//   cbs: channels per block (multiple of 8, <= 64); two: output columns per tile (multiple of 4, <= 28); bh: output rows per band (even)
template <typename T, int ACT>
__global__ void __launch_bounds__(256)
fd_dw5_pairs(const T *__restrict__ low, const T *__restrict__ skip, const unsigned *__restrict__ wpk, const float *__restrict__ bias,
             T *__restrict__ out, int H, int W, int C, int cbs, int two, int tiles_x, int bh)
{
    FD_DYN_SMEM(smem_raw);
    unsigned *s_buf = reinterpret_cast<unsigned *>(smem_raw);
    const fd_blk3 blk = fd_xcd_image_map();
    const int band = blk.x / tiles_x, tx = blk.x - band * tiles_x;
    const int c0 = blk.y * cbs, cend = c0 + cbs < C ? c0 + cbs : C, n = blk.z;
    const int x0 = tx * two, y0 = band * bh, y1 = y0 + bh < H ? y0 + bh : H;
    const int xend = x0 + two < W ? x0 + two : W;
    const int tid = threadIdx.x;

    // ---- tap role: channel lane l owns the two channels whose pair words sit at dwords 2l, 2l+1 of a staged pixel pair
    // (slot order of a pair's 64 channels: [half h = (c >> 2) & 1][group g = c >> 3][k = c & 3], so that the staging role's two 16-byte stores per
    // item are contiguous over its 8 group lanes)
    const int l = tid & 31, s = tid >> 5;
    const int cl = 8 * ((l & 15) >> 1) + 4 * (l >> 4) + 2 * (l & 1);
    const int c = c0 + cl;
    const bool tap_ok = c < cend && 4 * s < two && x0 + 4 * s < W;
    unsigned w[5][6][2];
    float b0 = 0.f, b1 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            fd_u32x2 v = {0u, 0u};
            if (tap_ok) v = *reinterpret_cast<const fd_u32x2 *>(wpk + (long)(ky * 6 + k) * C + c);
            w[ky][k][0] = v.x; w[ky][k][1] = v.y;
        }
    if (tap_ok) { b0 = bias[c]; b1 = bias[c + 1]; }

    // ---- staging role: item = (row sr of the step's two rows, pair jp, 8-channel group g): two skip pixels + their low-resolution parent
    const int sr = tid >> 7, jp = (tid >> 3) & 15, g = tid & 7;
    const int spx = x0 - 2 + 2 * jp;                        // first pixel of the pair (even: W is even, so a pair is inside or outside as a whole)
    const bool st_item = 2 * jp < two + 4;
    const bool st_ok = st_item && spx >= 0 && spx < W && c0 + 8 * g < cend;
    const int qx = spx < 0 ? 0 : (spx >= W ? W - 2 : spx), qc = c0 + 8 * g < cend ? c0 + 8 * g : 0;   // clamped: every lane loads unconditionally
    const int Hs = H >> 1, Ws = W >> 1;
    const T *skip_n = skip + (long)n * H * W * C, *low_n = low + (long)n * Hs * Ws * C;
    const unsigned sk_col = fd_mul24((unsigned)qx, (unsigned)C) + (unsigned)qc, lo_col = fd_mul24((unsigned)(qx >> 1), (unsigned)C) + (unsigned)qc;
    unsigned *const st_dst = s_buf + sr * FD_DW5P_ROW + jp * 64 + g * 4;
    const unsigned *const tp_src = s_buf + (2 * s) * 64 + 2 * l;

    fd_u32x4 pa, pb, pl;                                    // the item's loads in flight: skip pixel 2j, skip pixel 2j+1, low parent
    bool pv = false;
    auto issue = [&](int it) {
        const int gy = y0 - 2 + 2 * it + sr;
        pv = st_ok && gy >= 0 && gy < H;
        const int qy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        const T *ps = skip_n + fd_mul24((unsigned)qy, fd_mul24((unsigned)W, (unsigned)C)) + sk_col;
        if (FD_DW5P_ABL & 1) { pa = fd_u32x4{(unsigned)it, 1u, 2u, 3u}; pb = pa; pl = pa; return; }
        pa = *reinterpret_cast<const fd_u32x4 *>(ps);
        pb = *reinterpret_cast<const fd_u32x4 *>(ps + C);
        pl = *reinterpret_cast<const fd_u32x4 *>(low_n + fd_mul24((unsigned)(qy >> 1), fd_mul24((unsigned)Ws, (unsigned)C)) + lo_col);
    };
    auto commit = [&](int buf) {
        if (!st_item) return;
        fd_u32x4 o0, o1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned e = fd_pair_sum(T{}, fd_perm(pb[i], pa[i], FD_PERM_LO), fd_perm(pl[i], pl[i], FD_PERM_LO));   // channel 2i
            const unsigned o = fd_pair_sum(T{}, fd_perm(pb[i], pa[i], FD_PERM_HI), fd_perm(pl[i], pl[i], FD_PERM_HI));   // channel 2i + 1
            const unsigned ev = pv ? e : 0u, ov = pv ? o : 0u;
            if (i < 2) { o0[2 * i] = ev; o0[2 * i + 1] = ov; } else { o1[2 * i - 4] = ev; o1[2 * i - 3] = ov; }
        }
        unsigned *d = st_dst + buf * (2 * FD_DW5P_ROW);
        *reinterpret_cast<fd_u32x4 *>(d) = o0;
        *reinterpret_cast<fd_u32x4 *>(d + 32) = o1;
    };

    const int n_it = (y1 - y0 + 4) >> 1;                    // steps of two input rows: rows y0 - 2 ... y1 + 1
    issue(0);
    commit(0);
    __syncthreads();

    unsigned win[5][4][2];                                  // input row (y0 - 2 + r) lives in win[r % 5]: [pair of the strip][channel]
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int p = 0; p < 4; ++p) { win[a][p][0] = 0u; win[a][p][1] = 0u; }

    // output exchange tile of this wave: [8 pixels = 2 strips x 4][32 channel pairs]
    unsigned *const xt = s_buf + 4 * FD_DW5P_ROW + (tid >> 6) * 256;
    unsigned *const xw = xt + ((s & 1) * 4) * 32 + (cl >> 1);
    const unsigned *const xr = xt + (tid & 63) * 4;
    const int fl_x = x0 + 4 * (2 * (tid >> 6) + ((tid & 63) >> 5)) + ((tid >> 3) & 3), fl_c = c0 + 8 * (tid & 7);
    const bool fl_ok = fl_x < xend && fl_c < cend;
    T *const out_n = out + (long)n * H * W * C;

    // one output row from the window into the exchange tile; SB = slot of the row under filter row 0
    auto out_row = [&](auto SB) {
        constexpr int sb = decltype(SB)::value;
        float acc[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j][0] = b0; acc[j][1] = b1; }
#pragma unroll
        for (int ky = 0; ky < ((FD_DW5P_ABL & 4) ? 1 : 5); ++ky) {
            const unsigned (&R)[4][2] = win[(sb + ky) % 5];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    acc[0][ch] = fd_dot2(T{}, R[k][ch], w[ky][k][ch], acc[0][ch]);
                    acc[1][ch] = fd_dot2(T{}, R[k][ch], w[ky][3 + k][ch], acc[1][ch]);
                    acc[2][ch] = fd_dot2(T{}, R[k + 1][ch], w[ky][k][ch], acc[2][ch]);
                    acc[3][ch] = fd_dot2(T{}, R[k + 1][ch], w[ky][3 + k][ch], acc[3][ch]);
                }
                FD_DW5P_FENCE();                             // eight independent accumulation chains stay interleaved (the scheduler would otherwise run them two at a time)
            }
        }
        // the wave's 8 pixels x 64 channels of this row go through its exchange tile, so that every lane stores 16 contiguous bytes (flush_row)
#pragma unroll
        for (int j = 0; j < 4; ++j) xw[j * 32] = fd_pack2(T{}, fd_act<ACT>(acc[j][0]), fd_act<ACT>(acc[j][1]));
    };
    // all lanes of the wave: lane L stores pixel L >> 3 of the wave's eight, channels 8 * (L & 7) .. + 7 (4-byte stores of a lane's own two channels
    // measured 18 of 33 us on decode_conv5.0: tools/microbench/dw5pairs.hip, profiles/r06)
    auto flush_row = [&](int y) {
        fd_wave_lds_fence();
        const fd_u32x4 v = *reinterpret_cast<const fd_u32x4 *>(xr);
        fd_wave_lds_fence();
        if (fl_ok && y < y1 && (!(FD_DW5P_ABL & 2) || v[0] == 0x12345u))
            *reinterpret_cast<fd_u32x4 *>(out_n + fd_mul24(fd_mul24((unsigned)y, (unsigned)W) + (unsigned)fl_x, (unsigned)C) + (unsigned)fl_c) = v;
    };

    // one step: the next step's loads are issued, this step's two rows move from LDS into the window, two output rows are produced, the next
    // step's rows are committed to the other buffer.  PH = step number mod 5 (the window slots of a step are compile-time constants)
    auto step = [&](auto PH, int it) {
        constexpr int ph = decltype(PH)::value;
        const bool more = it + 1 < n_it;
        if (more) issue(it + 1);                             // next step's loads fly under this step's taps
        // (the step's second row takes the window slot of the oldest row, which the first output row still needs: it waits in `nb`)
        fd_u32x2 nb[4] = {};
        if (tap_ok) {
            const unsigned *src = tp_src + (it & 1) * (2 * FD_DW5P_ROW);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const fd_u32x2 v = *reinterpret_cast<const fd_u32x2 *>(src + p * 64);
                win[(2 * ph) % 5][p][0] = v.x; win[(2 * ph) % 5][p][1] = v.y;
                nb[p] = *reinterpret_cast<const fd_u32x2 *>(src + FD_DW5P_ROW + p * 64);
            }
            if (it >= 2) out_row(fd_int<(2 * ph + 1) % 5>{});
        }
        if (it >= 2) flush_row(y0 - 4 + 2 * it);
        if (tap_ok) {
#pragma unroll
            for (int p = 0; p < 4; ++p) { win[(2 * ph + 1) % 5][p][0] = nb[p].x; win[(2 * ph + 1) % 5][p][1] = nb[p].y; }
            if (it >= 2) out_row(fd_int<(2 * ph + 2) % 5>{});
        }
        if (it >= 2) flush_row(y0 - 3 + 2 * it);
        if (more) commit((it + 1) & 1);
        __syncthreads();
    };
    for (int it0 = 0; it0 < n_it; it0 += 5) {
        step(fd_int<0>{}, it0);
        if (it0 + 1 < n_it) step(fd_int<1>{}, it0 + 1);
        if (it0 + 2 < n_it) step(fd_int<2>{}, it0 + 2);
        if (it0 + 3 < n_it) step(fd_int<3>{}, it0 + 3);
        if (it0 + 4 < n_it) step(fd_int<4>{}, it0 + 4);
    }
}
