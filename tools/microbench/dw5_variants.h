// tools/microbench/dw5_variants.h -- NOT product code: the two alternative forms of the round-6 pixel-pair 5x5 kernel that were built, verified against
// the CPU reference and measured against fd_dw5_rows (fast-depth_amd/csrc/fd_kernels_dw5p.h) on MI355X, and lost or tied (profiles/r06/dw5_microbench_*.txt):
//   fd_dw5_pairs: 256-thread workgroup, VALU-staged pixel pairs in LDS (double buffered, one barrier per two rows), outputs through an exchange tile
//   fd_dw5_dma:   independent waves fed by global_load_lds_dwordx4, all LDS reads of a step ahead of the next transfer
// All three land within 3 % of each other (decode_conv5.0 fp16: 33.4 / 32.3 / 33.3 us): the bound is the VALU work itself.
#pragma once
#include "../../fast-depth_amd/csrc/fd_kernels_dw5p.h"
#ifndef FD_DW5P_ABL
#define FD_DW5P_ABL 0
#endif
#define FD_DW5P_NP 16            // pairs per staged row: (28 output columns + 4 halo pixels) / 2
#define FD_DW5P_ROW (FD_DW5P_NP * 64)   // dwords per staged row: [pair][64 channel slots]
#define FD_DW5P_LDS (2 * 2 * FD_DW5P_ROW * 4 + 4 * 1024)   // two buffers x two rows + one 1 KiB output exchange tile per wave
#ifndef FD_DW5P_FENCE
#define FD_DW5P_FENCE() FD_SCHED_FENCE()
#endif

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
    auto issue = [&](int it) FD_INLINE_LAMBDA {
        const int gy = y0 - 2 + 2 * it + sr;
        pv = st_ok && gy >= 0 && gy < H;
        const int qy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        const T *ps = skip_n + fd_mul24((unsigned)qy, fd_mul24((unsigned)W, (unsigned)C)) + sk_col;
        if (FD_DW5P_ABL & 1) { pa = fd_u32x4{(unsigned)it, 1u, 2u, 3u}; pb = pa; pl = pa; return; }
        pa = *reinterpret_cast<const fd_u32x4 *>(ps);
        pb = *reinterpret_cast<const fd_u32x4 *>(ps + C);
        pl = *reinterpret_cast<const fd_u32x4 *>(low_n + fd_mul24((unsigned)(qy >> 1), fd_mul24((unsigned)Ws, (unsigned)C)) + lo_col);
    };
    auto commit = [&](int buf) FD_INLINE_LAMBDA {
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
    auto out_row = [&](auto SB) FD_INLINE_LAMBDA {
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
    auto flush_row = [&](int y) FD_INLINE_LAMBDA {
        fd_wave_lds_fence();
        const fd_u32x4 v = *reinterpret_cast<const fd_u32x4 *>(xr);
        fd_wave_lds_fence();
        if (fl_ok && y < y1 && (!(FD_DW5P_ABL & 2) || v[0] == 0x12345u))
            *reinterpret_cast<fd_u32x4 *>(out_n + fd_mul24(fd_mul24((unsigned)y, (unsigned)W) + (unsigned)fl_x, (unsigned)C) + (unsigned)fl_c) = v;
    };

    // one step: the next step's loads are issued, this step's two rows move from LDS into the window, two output rows are produced, the next
    // step's rows are committed to the other buffer.  PH = step number mod 5 (the window slots of a step are compile-time constants)
    auto step = [&](auto PH, int it) FD_INLINE_LAMBDA {
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

// ------------------------------------------------------------------------------------------------------------------------------------
// fd_dw5_dma: the row-walking wave of fd_dw5_rows fed through LDS-DMA instead of 4-byte loads.  Measured on the two kernels above (MI355X, B = 32,
// decode_conv5.0, profiles/r06/dw5_microbench.txt): both take 33.5 us although their VALU work alone takes 23.6 (no loads, no stores) -- fd_dw5_rows
// is bound by the texture-address path (a wave-wide 4-byte access costs the same 16 cycles as a 16-byte one: 28 accesses per step and wave), fd_dw5_pairs
// by its per-step workgroup barrier and the registers its in-flight loads occupy.  Here
//   * a wave is on its own (no barrier): it owns (image, <= 64 channels, 8 output columns) and walks down a band, window + taps in registers;
//   * its input arrives by global_load_lds_dwordx4: 16 bytes per lane straight into the wave's LDS staging area (4 instructions per step of two
//     rows, no VGPRs in flight, one step ahead); the tap lanes read their 8 pixels x 2 channels per row from LDS as 4-byte words (2-cycle LDS
//     reads instead of 16-cycle address-path accesses) and build the pixel pairs with v_perm_b32 as fd_dw5_rows does;
//   * outputs leave through the wave's exchange tile as 16-byte stores (one per lane and step);
//   * every LDS read of a step happens BEFORE the step issues the next DMA, so no LDS read ever waits for a transfer that was just started.
// Horizontal zero padding: staging slots of pixels outside the image are zeroed once and never written (their DMA lanes are switched off).
// grid (ceil(strip-pair groups * bands / 4), channel blocks, images) through fd_xcd_image_map; block 256 = 4 independent waves; W % 4 == 0, H even.
// ------------------------------------------------------------------------------------------------------------------------------------
#define FD_DW5D_STAGE 1024       // dwords per staging buffer: two skip rows [12 px][32 dwords] + the parent row [6 px][32 dwords] (+ pad)
template <typename T, int ACT>
__global__ void __launch_bounds__(256) FD_DW5R_ATTR
fd_dw5_dma(const T *__restrict__ low, const T *__restrict__ skip, const unsigned *__restrict__ wpk, const float *__restrict__ bias,
           T *__restrict__ out, int H, int W, int C, int cbs, int groups_x, int bh)
{
    __shared__ __attribute__((aligned(16))) unsigned s_stage[4][2][FD_DW5D_STAGE];
    __shared__ __attribute__((aligned(16))) unsigned s_xchg[4][2][256];          // [wave][first / second row of a step][8 pixels][32 channel pairs]
    const fd_blk3 blk = fd_xcd_image_map();
    const int wave = FD_UNIFORM((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int item = blk.x * 4 + wave;
    const int band = item / groups_x, sg = item - band * groups_x;
    if (band * bh >= H) return;                              // (a whole wave: the waves of a workgroup share nothing)
    const int c0 = blk.y * cbs, cend = c0 + cbs < C ? c0 + cbs : C, n = blk.z;
    const int x0 = 8 * sg, y0 = band * bh, y1 = y0 + bh < H ? y0 + bh : H;
    const int Hs = H >> 1, Ws = W >> 1;

    // ---- tap role: 2 strips x 32 channel lanes
    const int l = lane & 31, s2 = lane >> 5, c = c0 + 2 * l, xs = x0 + 4 * s2;
    const bool tap_ok = c < cend && xs < W;
    unsigned w[5][6][2];
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            fd_u32x2 v = {0u, 0u};
            if (tap_ok) v = *reinterpret_cast<const fd_u32x2 *>(wpk + (long)(ky * 6 + k) * C + c);
            w[ky][k][0] = v.x; w[ky][k][1] = v.y;
        }
    float b0 = 0.f, b1 = 0.f;
    if (tap_ok) { b0 = bias[c]; b1 = bias[c + 1]; }

    // ---- DMA role: slot q = 64 k + lane of the two skip rows [row][12 px][8 groups of 8 channels]; lanes 0..47 also fetch the parent row [6 px][8 groups]
    const T *skip_n = skip + (long)n * H * W * C, *low_n = low + (long)n * Hs * Ws * C;
    unsigned d_off[3], d_row[3], dl_off;
    bool d_ok[3], dl_ok;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = 64 * k + lane, row = q >= 96 ? 1 : 0, rem = q - 96 * row, px = rem >> 3, g = rem & 7, x = x0 - 2 + px;
        d_ok[k] = x >= 0 && x < W && c0 + 8 * g < cend;
        d_row[k] = (unsigned)row;
        d_off[k] = d_ok[k] ? fd_mul24((unsigned)x, (unsigned)C) + (unsigned)(c0 + 8 * g) : 0u;
    }
    {
        const int pxl = lane >> 3, g = lane & 7, xl = (x0 >> 1) - 1 + pxl;
        dl_ok = lane < 48 && xl >= 0 && xl < Ws && c0 + 8 * g < cend;
        dl_off = dl_ok ? fd_mul24((unsigned)xl, (unsigned)C) + (unsigned)(c0 + 8 * g) : 0u;
    }
    // the staging buffers start as zeros: slots whose DMA lane is switched off (outside the image / the channel block) are the zero padding
    for (int i = lane * 4; i < 2 * FD_DW5D_STAGE; i += 256) *reinterpret_cast<fd_u32x4 *>(&s_stage[wave][0][0] + i) = fd_u32x4{0u, 0u, 0u, 0u};
    fd_wave_fence();

    bool nv = false;                                         // (wave-uniform) whether the rows of the step in flight are inside the image
    auto issue = [&](int it) FD_INLINE_LAMBDA {
        const int r = y0 - 2 + 2 * it;                       // even: rows r, r + 1 and their parent row r / 2 are inside the image together or not at all
        nv = r >= 0 && r < H;
        if (!nv) return;
        unsigned *dst = &s_stage[wave][it & 1][0];
        const T *ps = skip_n + fd_mul24((unsigned)r, fd_mul24((unsigned)W, (unsigned)C));
        const unsigned rowe = fd_mul24((unsigned)W, (unsigned)C);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (d_ok[k]) fd_glds16(reinterpret_cast<const float *>(ps + d_off[k] + d_row[k] * rowe), reinterpret_cast<float *>(dst + 256 * k));
        if (dl_ok) fd_glds16(reinterpret_cast<const float *>(low_n + fd_mul24((unsigned)(r >> 1), fd_mul24((unsigned)Ws, (unsigned)C)) + dl_off), reinterpret_cast<float *>(dst + 768));
    };

    unsigned win[5][4][2];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int p = 0; p < 4; ++p) { win[a][p][0] = 0u; win[a][p][1] = 0u; }

    unsigned ra[8], rb[8], rl[4];                            // this step's rows as read from LDS: [pixel of the strip] = (channel 2l, channel 2l + 1)
    auto convert = [&](auto SLOT, const unsigned (&raw)[8], bool rv) FD_INLINE_LAMBDA {
        constexpr int slot = decltype(SLOT)::value;
        if (rv) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                win[slot][p][0] = fd_pair_sum(T{}, fd_perm(raw[2 * p + 1], raw[2 * p], FD_PERM_LO), fd_perm(rl[p], rl[p], FD_PERM_LO));
                win[slot][p][1] = fd_pair_sum(T{}, fd_perm(raw[2 * p + 1], raw[2 * p], FD_PERM_HI), fd_perm(rl[p], rl[p], FD_PERM_HI));
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) { win[slot][p][0] = 0u; win[slot][p][1] = 0u; }
        }
    };
    // one output row from the window into exchange tile `xr`; SB = slot of the row under filter row 0; the dot2 are issued in source order
    auto out_row = [&](auto SB, int xr) FD_INLINE_LAMBDA {
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
        unsigned *xw = &s_xchg[wave][xr][(4 * s2) * 32 + l];
#pragma unroll
        for (int j = 0; j < 4; ++j) xw[j * 32] = fd_pack2(T{}, fd_act_raw<ACT>(acc[j][0]), fd_act_raw<ACT>(acc[j][1]));
    };
    // lane L stores pixel L >> 3 of the wave's eight, channels 8 * (L & 7) .. + 7, of the two rows the previous step left in the exchange tiles
    const int fl_x = x0 + (lane >> 3), fl_c = c0 + 8 * (lane & 7);
    const bool fl_ok = fl_x < W && fl_c < cend;
    T *const out_n = out + (long)n * H * W * C;
    auto flush = [&](int y) FD_INLINE_LAMBDA {                // rows y, y + 1 (y >= y0)
        const fd_u32x4 va = *reinterpret_cast<const fd_u32x4 *>(&s_xchg[wave][0][lane * 4]);
        const fd_u32x4 vb = *reinterpret_cast<const fd_u32x4 *>(&s_xchg[wave][1][lane * 4]);
        if (fl_ok) {
            T *po = out_n + fd_mul24(fd_mul24((unsigned)y, (unsigned)W) + (unsigned)fl_x, (unsigned)C) + (unsigned)fl_c;
            if (y < y1) *reinterpret_cast<fd_u32x4 *>(po) = va;
            if (y + 1 < y1) *reinterpret_cast<fd_u32x4 *>(po + fd_mul24((unsigned)W, (unsigned)C)) = vb;
        }
    };

    const int n_it = (y1 - y0 + 4) >> 1;
    issue(0);
    auto step = [&](auto PH, int it) FD_INLINE_LAMBDA {
        constexpr int ph = decltype(PH)::value;
        const bool rv = nv;                                  // validity of the rows that issue(it) fetched
        fd_wave_dma_wait();                                  // this step's transfer has landed (it was issued one step ago)
        if (tap_ok && rv) {
            const unsigned *src = &s_stage[wave][it & 1][(4 * s2) * 32 + l];
#pragma unroll
            for (int i = 0; i < 8; ++i) { ra[i] = src[i * 32]; rb[i] = src[384 + i * 32]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) rl[j] = s_stage[wave][it & 1][768 + (2 * s2 + j) * 32 + l];
        }
        if (it >= 3) flush(y0 - 6 + 2 * it);                  // the rows of step it - 1
        fd_wave_fence();                                 // every LDS read of this step is complete before the next transfer is issued
        if (it + 1 < n_it) issue(it + 1);
        if (tap_ok) {
            convert(fd_int<(2 * ph) % 5>{}, ra, rv);
            if (it >= 2) out_row(fd_int<(2 * ph + 1) % 5>{}, 0);
            convert(fd_int<(2 * ph + 1) % 5>{}, rb, rv);
            if (it >= 2) out_row(fd_int<(2 * ph + 2) % 5>{}, 1);
        }
    };
    for (int it0 = 0; it0 < n_it; it0 += 5) {
        step(fd_int<0>{}, it0);
        if (it0 + 1 < n_it) step(fd_int<1>{}, it0 + 1);
        if (it0 + 2 < n_it) step(fd_int<2>{}, it0 + 2);
        if (it0 + 3 < n_it) step(fd_int<3>{}, it0 + 3);
        if (it0 + 4 < n_it) step(fd_int<4>{}, it0 + 4);
    }
    fd_wave_fence();
    flush(y0 - 4 + 2 * (n_it - 1));
}
