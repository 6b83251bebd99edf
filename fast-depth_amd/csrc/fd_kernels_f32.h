// fd_kernels_f32.h -- fp32 inference kernels of the FastDepth hot path for gfx950 (MI355X).
//
// Activation layout is NHWC (channel fastest): a wavefront's 64 lanes read/write 16-byte channel
// groups of neighbouring pixels, so every global access is a run of >=128 contiguous bytes.
// BatchNorm (inference form) is folded into the weights by fd_pack_fold_f32, so each kernel is
// conv + bias + activation with no separate normalisation / activation pass.
//
// Kernels (reference statement each one replaces):
//   fd_pack_fold_f32    BN(eval) algebra: gamma*(x-mean)/sqrt(var+eps)+beta  == conv(w*s) + (beta-mean*s)
//   fd_stem3x3s2        conv_bn(3,32,2) as an MFMA product   imagenet/mobilenet.py:22-27,41
//   fd_dwconv_f32       depthwise 3x3 s1/s2 (+BN+ReLU6)      imagenet/mobilenet.py:31-33
//                       depthwise 5x5 (+BN+ReLU) with the nearest-x2 upsample and the additive skip
//                       of the PREVIOUS decoder stage folded into the tile read   models.py:61-68,723-729
//   fd_pw_gemm_f32      pointwise 1x1 (+BN+ReLU/ReLU6) as an fp32 MFMA GEMM       mobilenet.py:35-37; models.py:70-75
//   fd_head_pw1_f32     decode_conv6 = pointwise(32,1) evaluated at the low resolution and replicated
//                       2x2 on write (a 1x1 conv + per-channel BN + ReLU commutes with nearest upsampling) models.py:698,723,731
#pragma once
#include "fd_device.h"

// ------------------------------------------------------------------------------------------------
// Weight packing: wp = w * scale[cout]  (optionally transposed to [inner][cout]),  bias = beta - mean*scale
// ------------------------------------------------------------------------------------------------
template <typename TW>
__global__ void __launch_bounds__(256)
fd_pack_fold(const float *__restrict__ w, const float *__restrict__ gamma, const float *__restrict__ beta,
             const float *__restrict__ mean, const float *__restrict__ var, float eps,
             TW *__restrict__ wp, float *__restrict__ bias, int cout, int inner, int transpose, int row_pitch)
{
    // transpose == 0: wp[co][row_pitch] (rows zero-padded beyond `inner`);  transpose == 1: wp[inner][cout]
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = transpose ? (long)cout * inner : (long)cout * row_pitch;
    if (idx < total) {
        const int per = transpose ? inner : row_pitch;
        const int co = (int)(idx / per), i = (int)(idx - (long)co * per);
        const float scale = gamma[co] / sqrtf(var[co] + eps);
        const float v = i < inner ? w[(long)co * inner + i] * scale : 0.0f;
        if (transpose) fd_st1(wp + (long)i * cout + co, v); else fd_st1(wp + idx, v);
    }
    if (idx < cout) {
        const float scale = gamma[idx] / sqrtf(var[idx] + eps);
        bias[idx] = beta[idx] - mean[idx] * scale;
    }
}

// ------------------------------------------------------------------------------------------------
// Stem: dense 3x3 stride-2 conv, 3 -> Cout channels (Cout <= 64), as a [pixels x 27] x [27 x Cout] matrix product on
// v_mfma_f32_32x32x2_f32.  x is NCHW-planar (as the dataloader hands it over), y is NHWC.
// A workgroup owns 256 consecutive output pixels of ONE image (row-major): they span at most ceil(256 / Wo) + 1 output rows, i.e. a
// band of input rows that is staged ONCE into LDS with coalesced 16-byte loads -- zero padded, one plane after the other.  (Round 1
// gathered the 27 taps of every pixel straight from the planes: 28 scalar 4-byte loads per lane with stride-2 addresses, quarter-width
// transactions -> 30 us for 71 MB = 0.29 of HBM.)  A wave owns 64 of the pixels (two 32-row MFMA tiles): lane l supplies
// A[pixel l%32][tap 2s + l/32] for the 14 K steps (tap 27 is zero padding) from LDS and B[tap][channel l%32] from the folded weights
// wp[27][Cout].  The D layout (lane = channel, register = pixel) goes through LDS once more so that the NHWC store is 16 bytes per
// lane: a pixel's Cout channels are one contiguous run.  Accumulators start at the folded-BN bias.
// ------------------------------------------------------------------------------------------------
template <typename T, int ACT, int CHUNK>
__global__ void __launch_bounds__(256)
fd_stem3x3s2(const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
             T *__restrict__ y, int B, int H, int W, int Cout)
{
    (void)CHUNK; (void)B;
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1;
    const fd_blk3 blk = fd_xcd_image_map2();                 // the pixel blocks of an image on one XCD: their input bands overlap
    const int n = blk.z;
    const int p0 = blk.x * 256, npix = Ho * Wo;             // pixel range of this workgroup within image n
    const int p1 = p0 + 256 < npix ? p0 + 256 : npix;
    const int oy_first = p0 / Wo, oy_last = (p1 - 1) / Wo;
    const int iy_first = 2 * oy_first - 1, nrows = 2 * (oy_last - oy_first) + 3;       // input rows iy_first .. iy_first + nrows - 1
    const int PW = W + 4;                                    // patch row: x = -1 at index 3 (so that x = 0 is 16-byte aligned), x = W at index W + 4 - ... see below
    // patch[c][r][4 + x] for x in [-1, W]: index 3 holds the left padding, index 4 + W the right padding
    const int PR = PW + 4;                                   // row pitch in floats (multiple of 4)
    float *patch = smem;                                     // [3][nrows][PR]
    const float *xn = x + (long)n * 3 * H * W;
    const int W4 = W >> 2;                                   // W % 4 == 0 (W % 32 == 0)
    // staging with memory-level parallelism: up to U 16-byte loads per work-item are requested back to back before any of them is
    // written to LDS (a load -> ds_write chain per chunk would serialise ~6 HBM round trips per workgroup)
    // (no integer division anywhere on the per-lane path: a wave takes patch rows wave, wave + 4, ..., a lane one 16-byte chunk of the row)
    constexpr int U = 8;
    for (int rbase = wave; rbase < 3 * nrows; rbase += 4 * U)
        for (int qb = lane; qb < W4; qb += 64) {
            fd_f32x4 v[U];
            int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = rbase + 4 * u;
                const int c = (rr >= nrows) + (rr >= 2 * nrows), r = rr - c * nrows;
                const int iy = iy_first + r;
                const bool in_patch = rr < 3 * nrows;
                const bool ok = in_patch && iy >= 0 && iy < H;
                const int qy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), qc = c > 2 ? 2 : c;      // clamped address, unconditional load, padding by select
                v[u] = CHUNK == 103 ? fd_zero4() : fd_ld4(xn + ((long)qc * H + qy) * W + qb * 4);   // (CHUNK > 100: ablations of tools/microbench/stem.hip)
                if (!ok) v[u] = fd_zero4();
                off[u] = in_patch ? rr * PR + 4 + qb * 4 : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (off[u] >= 0) fd_st4(patch + off[u], v[u]);
        }
    for (int rr = tid; rr < 3 * nrows; rr += 256) { patch[rr * PR + 3] = 0.0f; patch[rr * PR + 4 + W] = 0.0f; }   // left / right padding columns
    __syncthreads();
    const int pw0 = p0 + wave * 64;                          // this wave's 64 pixels
    const int oyw = FD_UNIFORM(pw0 / Wo), oxw = pw0 - oyw * Wo;
    float a[2][14];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int p = pw0 + 32 * m + i;
        const bool valid = p < p1;
        int oy = oyw, ox = oxw + 32 * m + i;                  // pixel p = row oy, column ox: walk from the wave's first pixel, no division
        while (ox >= Wo) { ox -= Wo; ++oy; }
        if (!valid) { oy = oy_first; ox = 0; }
        const float *base = patch + (2 * (oy - oy_first)) * PR + 4 + 2 * ox - 1;     // tap (ky = 0, kx = 0) of plane 0
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int t0 = 2 * s, t1 = 2 * s + 1;            // tap index 2s + h: both alternatives are compile-time, the lane half selects
            const int c = h ? t1 / 9 : t0 / 9, ky = h ? (t1 % 9) / 3 : (t0 % 9) / 3, kx = h ? t1 % 3 : t0 % 3;
            const bool ok = valid && (2 * s + h) < 27;
            const int qc = c > 2 ? 2 : c;
            const float v = base[(qc * nrows + ky) * PR + kx];
            a[m][s] = ok ? v : 0.0f;
        }
    }
    __syncthreads();                                         // the patch is consumed: its LDS becomes the output staging area
    float *otile = smem + wave * 64 * 36;                    // [64 pixels][32 channels + 4]
    for (int n0 = 0; n0 < Cout; n0 += 32) {
        const int col = n0 + i;
        const bool col_ok = col < Cout;
        float b[14];
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int t = 2 * s + h;
            b[s] = (col_ok && t < 27) ? wp[t * Cout + col] : 0.0f;
        }
        const float bv = col_ok ? bias[col] : 0.0f;
        fd_f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = bv;
        // tile 0 first, then tile 1: the LDS transposition and the stores of tile 0 are issued under the MFMAs of tile 1
        const int cw = Cout - n0 < 32 ? Cout - n0 : 32;      // channels of this chunk (multiple of 8)
        const int lsh = cw >= 32 ? 3 : (cw >= 16 ? 2 : 1), lpp = 1 << lsh;     // lanes per pixel (4 channels each): 8, 4 or 2
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int s = 0; s < 14; ++s) if (CHUNK != 102) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][s], b[s], acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) otile[(32 * m + (r & 3) + 8 * (r >> 2) + 4 * h) * 36 + i] = fd_act<ACT>(acc[m][r]);
            // (wave-private tile: the wave's own LDS writes are visible to its own reads without a workgroup barrier once they have completed)
            fd_wave_lds_fence();
            for (int q = lane; q < 32 * lpp; q += 64) {
                const int px = 32 * m + (q >> lsh), c4 = (q & (lpp - 1)) * 4;
                const int p = pw0 + px;
                if (p < p1 && (CHUNK != 101 || otile[px * 36 + c4] == 12345.f)) fd_st4(y + ((long)n * npix + p) * Cout + n0 + c4, fd_ld4(otile + px * 36 + c4));
            }
        }
        fd_wave_lds_fence();
    }
}

// ------------------------------------------------------------------------------------------------
// Depthwise K x K conv, stride S, NHWC, LDS-tiled.
//   MODE 0: input read as stored.
//   MODE 1: input is the nearest-x2 upsampling of `in` (stored at half resolution)      models.py:723
//   MODE 2: MODE 1 plus `skip` (stored at full resolution) added                          models.py:724-729
// A workgroup owns TH x TW output pixels x CB channels (CB = 4 << cbq).  Phase 1 stages the
// (TH-1)*S+K by (TW-1)*S+K input patch -- zero padded, upsampled and skip-added on the fly -- into LDS
// with 16-byte loads; phase 2 gives every work-item strips of 4 output pixels x 4 channels: per filter
// row it pulls 3*S+K input vectors from LDS once and reuses them across the K taps and 4 outputs.
// The upsampled / summed tensor is never written to HBM.
// ------------------------------------------------------------------------------------------------
// N = channels per work-item (fd_lane, fd_device.h): 4 = fp32 patches, 16 bytes per lane in LDS; 8 (16-bit T only) = patches kept in the storage
// type, 16 bytes per lane in memory and in LDS.  PSTR is the patch pitch in LDS elements (floats for N = 4, 16-bit words for N = 8).
template <typename T, int K, int S, int MODE, int ACT, int N>
__global__ void __launch_bounds__(256)
fd_dwconv(const T *__restrict__ in, const T *__restrict__ skip, const float *__restrict__ wp,
          const float *__restrict__ bias, T *__restrict__ out, int Hin, int Win, int Ho, int Wo, int C,
          int cbq, int TH, int TW, int tiles_x, int csplit, int PSTR)
{
    typedef fd_lane<T, N> LN;
    typedef typename LN::vec vec;
    typedef typename LN::raw raw;
    typedef typename LN::lds_t lds_t;
    constexpr int P = K / 2;
    constexpr int NIN = 3 * S + K;                       // input columns feeding 4 adjacent outputs
    FD_DYN_SMEM(smem_raw);
    const int lanes_c = 1 << cbq, CB = lanes_c * N;       // PSTR: bank-conflict-free row pitch of the patch image (host: pick_patch_pitch)
    const int TH_in = (TH - 1) * S + K, TW_in = (TW - 1) * S + K;
    lds_t *s_in = reinterpret_cast<lds_t *>(smem_raw);    // [TH_in*TW_in][PSTR]
    float *s_w = reinterpret_cast<float *>(smem_raw + ((size_t)TH_in * TW_in * PSTR * sizeof(lds_t) + 15) / 16 * 16);   // [K*K][CB]
    float *s_b = s_w + K * K * CB;                        // [CB]
    const fd_blk3 blk = fd_xcd_image_map();                // all tiles / channel blocks of an image on one XCD: halo re-reads hit its L2
    const int ty = blk.x / tiles_x, tx = blk.x - ty * tiles_x;
    const int c0 = blk.y * CB, n = blk.z;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - P, ix0 = ox0 * S - P;
    const int tid = threadIdx.x, c4 = tid & (lanes_c - 1), pt = tid >> cbq, npt = 256 >> cbq;
    const int cg = c0 + c4 * N;                            // first global channel of this lane
    const bool c_ok = cg < C;

    // taps and bias of this channel block: requested now, written to LDS after the patch loads have been issued -- one round trip at the head of
    // the workgroup instead of two (K*K*lanes_c <= 200 vectors: one per work-item)
    const bool w_item = tid < K * K * lanes_c;
    const int w_t = tid >> cbq, w_cc = tid & (lanes_c - 1);
    vec w_reg = LN::zero(), b_reg = LN::zero();
    if (w_item && c0 + w_cc * N < C) w_reg = LN::ldf(wp + (long)w_t * C + c0 + w_cc * N);
    if (tid < lanes_c && c0 + tid * N < C) b_reg = LN::ldf(bias + c0 + tid * N);

    // Staging with memory-level parallelism: U patch pixels per work-item are requested back to back (2*U
    // independent 16-byte loads in flight in MODE 2) before any of them is consumed; a load -> add -> ds_write
    // chain per pixel would serialise ~12 HBM round trips per workgroup.
    const int npx_in = TH_in * TW_in;
    constexpr int U = 8;
    fd_px_walk wk(pt, npt, TW_in);
    for (int base = pt; base < npx_in; base += npt * U) {
        raw v[U], sk[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = base + u * npt;
            const int iy = wk.iy, ix = wk.ix;
            wk.next();
            const int gy = iy0 + iy, gx = ix0 + ix;
            ok[u] = px < npx_in && c_ok && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
            // branch-free: the address is clamped into the image / tensor and every lane loads unconditionally (all 2*U loads go out
            // back to back; measured -30 % on the train-mode twin of this kernel); the zero padding is applied when the value is stored
            const int qy = gy < 0 ? 0 : (gy >= Hin ? Hin - 1 : gy), qx = gx < 0 ? 0 : (gx >= Win ? Win - 1 : gx);
            const int qg = c_ok ? cg : 0;
            if (MODE == 0) {
                v[u] = LN::ldraw(in + fd_nhwc(n, Hin, qy, Win, qx, C, qg));
            } else {
                const int Hs = Hin >> 1, Ws = Win >> 1;
                if (MODE == 3) {
                    // channel concatenation cat(up2(in), skip): channels [0, csplit) come from the low-resolution tensor (pitch
                    // csplit), the rest from the skip tensor (pitch C - csplit); a lane's N channels never straddle (csplit % N == 0)
                    if (qg < csplit) v[u] = LN::ldraw(in + fd_nhwc(n, Hs, (qy >> 1), Ws, (qx >> 1), csplit, qg));
                    else v[u] = LN::ldraw(skip + fd_nhwc(n, Hin, qy, Win, qx, (C - csplit), (qg - csplit)));
                } else {
                    v[u] = LN::ldraw(in + fd_nhwc(n, Hs, (qy >> 1), Ws, (qx >> 1), C, qg));
                    if (MODE == 2) sk[u] = LN::ldraw(skip + fd_nhwc(n, Hin, qy, Win, qx, C, qg));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = base + u * npt;
            if (px < npx_in) {
                lds_t *dst = s_in + px * PSTR + c4 * N;
                // (N = 8: the up2(low) + skip sum is rounded to the storage type on its way into LDS; plain inputs are copied bit for bit)
                if (MODE == 2) { if (ok[u]) LN::lds_st_sum(dst, v[u], sk[u]); else LN::lds_st(dst, LN::zero()); }
                else if (ok[u]) LN::lds_st_raw(dst, v[u]);
                else LN::lds_st(dst, LN::zero());
            }
        }
    }
    if (w_item) LN::stf(s_w + w_t * CB + w_cc * N, w_reg);
    if (tid < lanes_c) LN::stf(s_b + tid * N, b_reg);
    __syncthreads();

    const int TWS = TW >> 2, nstrips = TH * TWS;
    const vec b4 = LN::ldf(s_b + c4 * N);
    for (int s = pt; s < nstrips; s += npt) {
        const int oy = s / TWS, ox = (s - oy * TWS) * 4;
        vec acc[4] = {b4, b4, b4, b4};
#pragma unroll 1   // one filter row in flight: keeps the kernel at <=128 VGPRs (>=4 waves/SIMD hide the LDS latency)
        for (int ky = 0; ky < K; ++ky) {
            const lds_t *row = s_in + ((oy * S + ky) * TW_in + ox * S) * PSTR + c4 * N;
            vec r[NIN];
#pragma unroll
            for (int i = 0; i < NIN; ++i) r[i] = LN::lds_ld(row + i * PSTR);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const vec w = LN::ldf(s_w + (ky * K + kx) * CB + c4 * N);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += r[j * S + kx] * w;
            }
        }
        const int gy = oy0 + oy;
        if (c_ok && gy < Ho) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gx = ox0 + ox + j;
                if (gx < Wo) LN::st(out + fd_nhwc(n, Ho, gy, Wo, gx, C, cg), fd_act4<ACT>(acc[j]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3, stride S, NHWC, register-window variant (no LDS, no barriers) for the encoder layers.
// In NHWC an image row is one contiguous run of W*C floats, so a work-item is indexed by q = x*(C/4) + c4
// (one 16-byte channel group of one column) and neighbouring work-items touch neighbouring 16-byte words:
// every wave access is a dense 1 KiB run whatever C is (also for the pruned, non-power-of-two widths).
// Each work-item walks down TH output rows keeping the 3 x 3 input window (and the 9 folded taps) in
// registers: per output row it loads only the 3*S new input vectors; the horizontal neighbours it needs are
// the same words its neighbours load, so they are L1/L2 hits and HBM sees every input byte once.
// ------------------------------------------------------------------------------------------------
template <typename T, int S, int ACT>
__global__ void __launch_bounds__(256)
fd_dw3_rows(const T *__restrict__ in, const float *__restrict__ wp, const float *__restrict__ bias,
            T *__restrict__ out, int H, int W, int Ho, int Wo, int C, int TH)
{
    const int CG = C >> 2;
    const fd_blk3 blk = fd_xcd_image_map();                  // the strips of an image on one XCD: they share their boundary rows in its L2
    const int q = blk.x * 256 + threadIdx.x;                 // output column-group index within a row
    if (q >= Wo * CG) return;
    const int xo = q / CG, c4 = q - xo * CG;
    const int n = blk.z;
    const int oy0 = blk.y * TH;
    const int oy1 = (oy0 + TH < Ho) ? oy0 + TH : Ho;
    fd_f32x4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = fd_ld4(wp + (long)t * C + c4 * 4);
    const fd_f32x4 b4 = fd_ld4(bias + c4 * 4);
    const T *img = in + (long)n * H * W * C + c4 * 4;
    const int x0 = xo * S - 1;                                  // leftmost input column of the window
    const bool okl = x0 >= 0, okr = (x0 + 2) < W;               // centre column x0+1 is always valid
    // branch-free: row / column indices are clamped into the image so that the three loads are always issued (back to back, no
    // exec-mask juggling between them); the zero padding is a select on the loaded value
    const int xl = okl ? x0 : x0 + 1, xr = okr ? x0 + 2 : x0 + 1;
    auto load_row = [&](int iy, fd_f32x4 &l, fd_f32x4 &c, fd_f32x4 &r) {
        const bool oky = iy >= 0 && iy < H;
        const int qy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        const T *p = img + (long)qy * W * C;
        const fd_f32x4 vl = fd_ld4(p + (long)xl * C), vc = fd_ld4(p + (long)(x0 + 1) * C), vr = fd_ld4(p + (long)xr * C);
        l = (oky && okl) ? vl : fd_zero4();
        c = oky ? vc : fd_zero4();
        r = (oky && okr) ? vr : fd_zero4();
    };
    T *o = out + (((long)n * Ho + oy0) * Wo) * C + (long)q * 4;
    if (S == 1) {
        fd_f32x4 r0l, r0c, r0r, r1l, r1c, r1r, r2l, r2c, r2r;
        load_row(oy0 - 1, r0l, r0c, r0r);
        load_row(oy0, r1l, r1c, r1r);
        for (int oy = oy0; oy < oy1; ++oy) {
            load_row(oy + 1, r2l, r2c, r2r);
            fd_f32x4 acc = b4;
            acc += r0l * w[0]; acc += r0c * w[1]; acc += r0r * w[2];
            acc += r1l * w[3]; acc += r1c * w[4]; acc += r1r * w[5];
            acc += r2l * w[6]; acc += r2c * w[7]; acc += r2r * w[8];
            fd_st4(o, fd_act4<ACT>(acc));
            o += (long)Wo * C;
            r0l = r1l; r0c = r1c; r0r = r1r; r1l = r2l; r1c = r2c; r1r = r2r;
        }
    } else {
        fd_f32x4 r0l, r0c, r0r, r1l, r1c, r1r, r2l, r2c, r2r;
        load_row(2 * oy0 - 1, r0l, r0c, r0r);
        for (int oy = oy0; oy < oy1; ++oy) {
            load_row(2 * oy, r1l, r1c, r1r);
            load_row(2 * oy + 1, r2l, r2c, r2r);
            fd_f32x4 acc = b4;
            acc += r0l * w[0]; acc += r0c * w[1]; acc += r0r * w[2];
            acc += r1l * w[3]; acc += r1c * w[4]; acc += r1r * w[5];
            acc += r2l * w[6]; acc += r2c * w[7]; acc += r2r * w[8];
            fd_st4(o, fd_act4<ACT>(acc));
            o += (long)Wo * C;
            r0l = r2l; r0c = r2c; r0r = r2r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Pointwise 1x1 conv as GEMM:  out[M][N] = act(A[M][K] * Wt[N][K32]^T + bias[N]),  M = B*H*W pixels (NHWC
// rows), K = Cin, N = Cout, all fp32, on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
// Wt rows are zero-padded to K32 = ceil32(K) by fd_pack_fold_f32, so a ragged last K tile multiplies by exact 0.
//
// 4 waves as WGM x WGN, each wave owns TM x TN tiles of 32x32 -> block tile (WGM*TM*32) x (WGN*TN*32), BK = 32.
//  * Operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write) into a
//    3-stage ring, issued TWO K tiles ahead; counted s_waitcnt vmcnt + a raw s_barrier keep those loads in
//    flight across the one barrier per K tile.  (With the fp32 MFMA at 64 cycles the kernel is latency, not
//    bandwidth, sensitive: a one-tile-ahead register prefetch left ~20% of the time exposed.)
//  * LDS rows are 128 bytes (8 chunks of 16 B) with an XOR swizzle chunk' = chunk ^ ((row>>1)&7).  LDS-DMA writes
//    lane-linearly, so the swizzle is applied to the per-lane SOURCE address (the 8 lanes of a row still read
//    one 128-byte line) and again on the fragment reads, which makes the 16-lane ds_read_b128 groups hit 16
//    distinct 16-byte slots.
//  * K-permutation trick: one MFMA step multiplies the k held by lanes 0-31 with the k' held by lanes 32-63,
//    and any pairing is valid as long as A and B use the same one.  Every lane fetches ONE 16-byte chunk (4
//    consecutive k of its row; lanes 0-31 chunk 2g, lanes 32-63 chunk 2g+1) per ds_read_b128 and feeds 4 MFMAs.
//  * XCD-aware 1-D grid: workgroup b runs on XCD b%8 (observed dispatch order); the N tiles of one M tile get
//    consecutive slots on ONE XCD so the A panel is fetched into that XCD's L2 once; weights stay L2 resident.
// Ragged M / N: source rows are clamped (finite garbage in rows that are never stored).
// ------------------------------------------------------------------------------------------------
#ifndef FD_F32_STAGES
#define FD_F32_STAGES 3      // depth of the LDS-DMA ring of fd_pw_gemm_f32 (build switch: 2 = 33 KB per 64 x 64 workgroup)
#endif
template <int WGM, int WGN, int TM, int TN, int ACT>
__global__ void __launch_bounds__(64 * WGM * WGN)
fd_pw_gemm_f32(const float *__restrict__ A, const float *__restrict__ Wt, const float *__restrict__ bias,
               float *__restrict__ out, int M, int N, int K, int K32, int m_tiles, int n_tiles)
{
    constexpr int NW = WGM * WGN;                          // waves per workgroup (1, 2 or 4)
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32, BK = 32;
    constexpr int ROWS = BM + BN;                          // LDS rows per stage (A rows then W rows), 32 floats each
    constexpr int STAGE = ROWS * BK;                       // floats per stage
    constexpr int RG = ROWS / 8 / NW;                      // LDS-DMA instructions (8 rows = 1 KiB each) per wave per K tile
    static_assert((ROWS / 8) % NW == 0, "row groups must divide evenly over the waves");
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % n_tiles, mt = (slot / n_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    // ---- LDS-DMA source pointers: this lane's row / swizzled chunk for each of the wave's row groups ----
    const float *src[RG];
    int src_chunk[RG];
    bool src_is_a[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        const int r = (wave + NW * i) * 8 + (lane >> 3);    // row within the stage
        const int c = (lane & 7) ^ ((r >> 1) & 7);          // global chunk that lands in LDS slot (r, lane&7)
        src_chunk[i] = c * 4;
        src_is_a[i] = r < BM;
        if (r < BM) { long row = m0 + r; if (row > M - 1) row = M - 1; src[i] = A + row * K; }
        else { int row = n0 + (r - BM); if (row > N - 1) row = N - 1; src[i] = Wt + (long)row * K32; }
    }
    auto issue = [&](int t) {
        float *dst = smem + (t % FD_F32_STAGES) * STAGE + wave * 8 * BK;
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            int k = t * BK + src_chunk[i];
            if (src_is_a[i] && k >= K) k = 0;                // ragged K: any finite data; the padded weights are 0 there
            fd_glds16(src[i] + k, dst + i * NW * 8 * BK);
        }
    };

    // accumulators start at the folded-BN bias of their column (a D register holds 16 rows of ONE column), so the
    // epilogue is just activation + store and no load is pending when the stores are issued
    fd_f32x16 acc[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        const float bv = col < N ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }

    // ---- fragment read offsets (floats) within a stage: row*32 + ((2g + h) ^ swz(row))*4 ----
    const int h = lane >> 5;
    int a_off[TM][4], b_off[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) a_off[i][g] = row * BK + (((2 * g + h) ^ ((row >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = BM + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) b_off[j][g] = row * BK + (((2 * g + h) ^ ((row >> 1) & 7)) << 2);
    }

    const int T = K32 / BK;
    issue(0);
    if (FD_F32_STAGES > 2 && T > 1) issue(1);
    for (int t = 0; t < T; ++t) {
        if (FD_F32_STAGES > 2 && t + 1 < T) fd_wait_vmcnt<RG>(); else fd_wait_vmcnt<0>();   // leave only tile t+1's loads in flight
        fd_block_barrier();                                  // tile t landed for every wave; stage (t+2)%3 is free again
        const float *cur = smem + (t % FD_F32_STAGES) * STAGE;
        // software pipeline inside the K tile: the fragments of chunk pair g+1 are requested before the MFMAs of
        // pair g are issued, and the LDS-DMA for tile t+2 is issued under the first fragment reads' latency
        fd_f32x4 a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = fd_ld4(cur + a_off[i][0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = fd_ld4(cur + b_off[j][0]);
        if (t + FD_F32_STAGES - 1 < T) issue(t + FD_F32_STAGES - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(g + 1) & 1][i] = fd_ld4(cur + a_off[i][g + 1]);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(g + 1) & 1][j] = fd_ld4(cur + b_off[j][g + 1]);
            }
            FD_SCHED_FENCE();
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][q], b[g & 1][j][q], acc[i][j], 0, 0, 0);
            FD_SCHED_FENCE();
        }
    }

    // epilogue: D register r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31 of the 32x32 tile
    const bool full = m0 + BM <= M;                          // workgroup-uniform: only the last M tile can be ragged
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
        if (col >= N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long rbase = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
            float *o = out + rbase * N + col;
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * N] = fd_act<ACT>(acc[i][j][r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (rbase + (r & 3) + 8 * (r >> 2) < M) o[((r & 3) + 8 * (r >> 2)) * N] = fd_act<ACT>(acc[i][j][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Head: pointwise Cin -> 1 (+BN+ReLU).  8 lanes share one low-resolution pixel (16-byte channel groups,
// dense 128-byte reads), a 3-step wave shuffle reduces the dot product, lane 0 writes the value once
// (up == 0) or as the 2x2 block it becomes after nearest upsampling (up == 1).
// ------------------------------------------------------------------------------------------------
template <typename T, int ACT>
__global__ void __launch_bounds__(256)
fd_head_pw1(const T *__restrict__ in, const float *__restrict__ wp, const float *__restrict__ bias,
            float *__restrict__ y, long npix, int h, int w, int Cin, int up)
{
    const long g = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int l8 = threadIdx.x & 7;
    float s = 0.0f;
    if (g < npix) {
        for (int c = l8 * 4; c < Cin; c += 32) {
            const fd_f32x4 v = fd_ld4(in + g * Cin + c), q = fd_ld4(wp + c);
            s = fmaf(v.x, q.x, s); s = fmaf(v.y, q.y, s); s = fmaf(v.z, q.z, s); s = fmaf(v.w, q.w, s);
        }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (g < npix && l8 == 0) {
        const float v = fd_act<ACT>(s + bias[0]);
        if (!up) { y[g] = v; return; }
        const int ox = (int)(g % w);
        const long t = g / w;
        const int oy = (int)(t % h);
        const long n = t / h;
        float *o = y + ((n * 2 * h + 2 * oy) * 2 * (long)w + 2 * ox);
        const fd_f32x2 vv = {v, v};
        *reinterpret_cast<fd_f32x2 *>(o) = vv;
        *reinterpret_cast<fd_f32x2 *>(o + 2 * w) = vv;
    }
}
