// fd_kernels_train.h -- TRAIN-mode forward kernels of the FastDepth hot path (gfx950).  T = storage type of the saved raw
// conv outputs z (float, or fd_bf16 for the 16-bit train plan); all arithmetic, statistics and tables are fp32.
//
// Train mode changes BatchNorm to batch statistics (reference: nn.BatchNorm2d instantiated at
// imagenet/mobilenet.py:25,32,36 and models.py:66,73, module in .train()): the statistics of a unit's conv
// output are only known after the whole conv has run, so BN cannot be folded into the weights.  The design:
//   * every unit stores its RAW conv output z (this is also exactly what backward needs: the activation mask
//     and x_hat are functions of z), and its epilogue ADDS its per-channel sum(z), sum(z^2) partials into the unit's
//     statistics rows with 64-bit integer atomics (fd_stat_add, fd_device.h: exact, order-independent);
//   * the CONSUMER's workgroups turn the rows into (scale s = gamma*invstd, shift t = beta - mean*s) for their own
//     channels in their prologue (fd_stat_table_block); one designated workgroup also writes the table (+ mean, invstd) the
//     backward pass reads and updates running_mean / running_var (momentum 0.1, UNBIASED variance)
//     and num_batches_tracked.  fd_bn_finalize_rows_f32 does the same as a launch of its own where no consumer can;
//   * the CONSUMER applies  a = act(z*s + t)  while it loads its input ("normalise on read"), so the
//     normalised / activated tensor is never written to HBM.  Nearest-x2 upsampling and the additive skips stay
//     fused into the consumer's read exactly as in the inference kernels (models.py:723-729).
// Reductions are deterministic: per-workgroup partials in fixed order, then exact integer accumulation across workgroups.
#pragma once
#include "fd_device.h"
#include "fd_bn_stats.h"

// measurement aid (tools/microbench/dwtrain.hip): shader-clock timestamps of a workgroup's phases; nothing in product builds
#ifdef FD_DW_PROBE
__device__ long long fd_dw_probe[8 * 16384];
#define FD_DW_PROBE_AT(k) do { if (threadIdx.x == 0) { const unsigned b_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); if (b_ < 16384) fd_dw_probe[8 * b_ + (k)] = clock64(); } } while (0)
__device__ int fd_dw_abl;                                 // ablation bits (microbench): 1 = no output stores, 2 = no patch loads
#define FD_DW_ABL(b) ((fd_dw_abl & (b)) != 0)
#else
#define FD_DW_PROBE_AT(k) ((void)0)
#define FD_DW_ABL(b) false
#endif

// staging depth of the LDS-tiled depthwise train kernels: patch pixels whose loads a work-item has in flight before it consumes the first (build switches for
// tools/build_variant.py sweeps).  A patch of more pixels than (pixel-threads x depth) costs a second, dependent round of loads.
// Measured (round 4, B = 32, paired backward family / step): bf16 plan 575.7 us / 2.513 ms at (8; 4, 4) -> 555.0 / 2.485 at (10; 6, 8); fp32 plan 679.8 / 4.269 ->
// 678.0 / 4.268 with the deeper forward / backward-data staging alone, 714 / 4.306 with the deeper weight-gradient staging (its fp32 vectors cost registers).
#ifndef FD_DW_U3
#define FD_DW_U3 10      // forward / backward-data, 3x3 units (whole-frame 14 x 14 tiles: 9 patch pixels per work-item)
#endif
#ifndef FD_DW_WU3
#define FD_DW_WU3 6      // backward-weights of the 16-bit plans, 3x3 / 5x5 units (fp32 plans: 4)
#define FD_DW_WU5 8
#endif

// bytes of an LDS patch image of npx pixels at pitch pstr (LDS elements), at least 8 KiB (the region doubles as fp32 reduction scratch) and a
// multiple of 16 bytes (the fp32 tap table follows it)
template <typename LT> __device__ __forceinline__ size_t fd_lds_patch_bytes(int npx, int pstr)
{
    const size_t b = (size_t)npx * pstr * sizeof(LT);
    return b < 8192 ? 8192 : (b + 15) / 16 * 16;
}
template <int ACT, typename V>
__device__ __forceinline__ V fd_bn_act4(V z, V s, V t)          // V: fd_f32x4 or fd_f32x8 (fd_lane)
{
    return fd_act4<ACT>(z * s + t);
}
// ------------------------------------------------------------------------------------------------
// Stem, train mode, forward.  A workgroup owns 256 consecutive output pixels [p0, p1) of ONE
// image (grid: blocks per image x images, fd_xcd_image_map2 -- neighbouring blocks' input bands overlap, so an image stays on one XCD's L2).
// fd_stem_stage_band brings the zero-padded band of input rows under those pixels into LDS with 16-byte row loads (the first
// generation issued 27 strided 4-byte loads per pixel):  patch[c][r][PR], input column x at index 4 + x (x = -1 at index 3, so that x = 0 is
// 16-byte aligned), rows iy_first .. iy_first + nrows - 1, PR = W + 8.
// ------------------------------------------------------------------------------------------------
struct fd_stem_band { int oy_first, nrows, PR; };
__device__ __forceinline__ fd_stem_band fd_stem_stage_band(const float *__restrict__ xn, float *patch, int H, int W, int Wo, int p0, int p1, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    fd_stem_band g;
    g.oy_first = p0 / Wo;
    const int oy_last = (p1 - 1) / Wo;
    const int iy_first = 2 * g.oy_first - 1;
    g.nrows = 2 * (oy_last - g.oy_first) + 3;
    g.PR = W + 8;
    const int W4 = W >> 2;                                    // W % 32 == 0
    constexpr int U = 8;                                      // row chunks in flight per work-item before the first LDS write
    for (int rbase = wave; rbase < 3 * g.nrows; rbase += 4 * U)
        for (int qb = lane; qb < W4; qb += 64) {
            fd_f32x4 v[U];
            int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = rbase + 4 * u;                 // row of the [3 * nrows] band: plane c, band row r
                const int c = (rr >= g.nrows) + (rr >= 2 * g.nrows), r = rr - c * g.nrows;
                const int iy = iy_first + r;
                const bool in_band = rr < 3 * g.nrows;
                const int qy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), qc = c > 2 ? 2 : c;       // clamped address, unconditional load, padding by select
                v[u] = fd_ld4(xn + ((long)qc * H + qy) * W + qb * 4);
                if (!(in_band && iy >= 0 && iy < H)) v[u] = fd_zero4();
                off[u] = in_band ? rr * g.PR + 4 + qb * 4 : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (off[u] >= 0) fd_st4(patch + off[u], v[u]);
        }
    for (int rr = tid; rr < 3 * g.nrows; rr += 256) { patch[rr * g.PR + 3] = 0.0f; patch[rr * g.PR + 4 + W] = 0.0f; }   // x = -1 and x = W
    return g;
}
// offset of tap t = (c, ky, kx) relative to a pixel's tap (0, 0, 0) inside the band
__device__ __forceinline__ int fd_stem_tap_offset(int t, int nrows, int PR)
{
    const int c = t / 9, ky = (t - c * 9) / 3, kx = t - c * 9 - ky * 3;
    return (c * nrows + ky) * PR + kx;
}
__device__ __forceinline__ float fd_round1(float, float v) { return v; }
__device__ __forceinline__ float fd_round1(fd_bf16, float v) { return fd_bf16_to_f32(fd_f32_to_bf16(v)); }
__device__ __forceinline__ float fd_round1(fd_half, float v) { return (float)(_Float16)v; }

// Forward: z[p][co] = sum_t tap[p][t] * w[co][t] as 32x32x2 MFMAs (a wave = 64 pixels = two 32-row tiles, K = 27 taps padded to 28, raw torch
// weights w[Cout][27]); the accumulators are rounded to T, transposed through a wave-private LDS tile for 16-byte NHWC stores, and their
// per-channel sums over the valid pixels are added to the unit's statistics rows (fd_stat_add; blk = image * gridDim.x + block picks the row).
// LDS: max(band, 4 x [64][36] output tiles) + [4][2][32] statistics.
template <typename T>
__global__ void __launch_bounds__(256)
fd_stem_train(const float *__restrict__ x, const float *__restrict__ w, T *__restrict__ z, fd_stat_rows sr, int H, int W, int Cout, int tile_floats)
{
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    float *red = smem + tile_floats;                          // [4 waves][2][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1, npix = Ho * Wo;
    const fd_blk3 blk = fd_xcd_image_map2();
    const int n = blk.z;
    const int p0 = blk.x * 256;
    const int p1 = p0 + 256 < npix ? p0 + 256 : npix;
    const fd_stem_band g = fd_stem_stage_band(x + (long)n * 3 * H * W, smem, H, W, Wo, p0, p1, tid);
    __syncthreads();
    const int pw0 = p0 + wave * 64;                           // this wave's 64 pixels
    float a[2][14];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int p = pw0 + 32 * m + i;
        const bool valid = p < p1;
        const int pq = valid ? p : p0;
        const int oy = pq / Wo, ox = pq - oy * Wo;
        const float *base = smem + (2 * (oy - g.oy_first)) * g.PR + 4 + 2 * ox - 1;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            // tap 2s + h of this half-wave in MFMA step s (t = 27: the zero pad); both alternatives have compile-time (c, ky, kx), the lane half selects
            const int o0 = fd_stem_tap_offset(2 * s, g.nrows, g.PR), o1 = fd_stem_tap_offset(2 * s + 1 < 27 ? 2 * s + 1 : 26, g.nrows, g.PR);
            const float v = base[h ? o1 : o0];
            a[m][s] = (valid && 2 * s + h < 27) ? v : 0.0f;
        }
    }
    __syncthreads();                                          // the band is consumed: its LDS becomes the output staging area
    float *otile = smem + wave * 64 * 36;                     // [64 pixels][32 channels + 4]
    const long blk_row = (long)n * gridDim.x + blk.x;
    for (int n0 = 0; n0 < Cout; n0 += 32) {
        const int col = n0 + i;
        const bool col_ok = col < Cout;
        float b[14];
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int t = 2 * s + h;
            b[s] = (col_ok && t < 27) ? w[(long)(col_ok ? col : 0) * 27 + (t < 27 ? t : 0)] : 0.0f;
        }
        fd_f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 14; ++s) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][s], b[s], acc[m], 0, 0, 0);
        }
        const int cw = Cout - n0 < 32 ? Cout - n0 : 32;       // channels of this chunk (a multiple of 8)
        const int lsh = cw >= 32 ? 3 : (cw >= 16 ? 2 : 1), lpp = 1 << lsh;     // lanes per pixel (4 channels each): 8, 4 or 2
        float ssum = 0.0f, ssq = 0.0f;                        // column i over this lane's rows of both tiles (rounded values, valid pixels)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = fd_round1(T{}, acc[m][r]);
                otile[row * 36 + i] = v;
                if (pw0 + row < p1 && col_ok) { ssum += v; ssq = fmaf(v, v, ssq); }
            }
        }
        fd_wave_lds_fence();                                  // wave-private tile
        for (int q = lane; q < 64 * lpp; q += 64) {
            const int px = q >> lsh, c4 = (q & (lpp - 1)) * 4;
            const int p = pw0 + px;
            if (p < p1) fd_st4(z + ((long)n * npix + p) * Cout + n0 + c4, fd_ld4(otile + px * 36 + c4));
        }
        ssum += __shfl_xor(ssum, 32); ssq += __shfl_xor(ssq, 32);             // the two half-waves hold different rows of column i
        if (lane < 32) { red[(wave * 2 + 0) * 32 + lane] = ssum; red[(wave * 2 + 1) * 32 + lane] = ssq; }
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5, c = tid & 31;
            if (n0 + c < Cout)
                fd_stat_add<FD_STAT_FWD>(sr, blk_row, Cout, which, n0 + c,
                                         (red[(0 * 2 + which) * 32 + c] + red[(1 * 2 + which) * 32 + c]) + (red[(2 * 2 + which) * 32 + c] + red[(3 * 2 + which) * 32 + c]));
        }
        __syncthreads();                                      // red and the output tiles are reused by the next channel chunk
    }
}

// Sums a pair of 4-channel vectors over all work-items of a 256-item workgroup that share the channel group c4 = tid & (lanes_c - 1) (the per-tile
// BatchNorm partial sums of the depthwise kernels).  Within a wave the items of a channel group are lanes c4, c4 + lanes_c, ...: a shuffle
// butterfly adds them (fixed order), the four wave sums meet through `red` (>= 4 * lanes_c * 8 floats of LDS that no work-item still reads).
// Work-items tid < lanes_c return true and hold the totals.  (Probe, tools/microbench/dwtrain.hip: the former 32-step serial LDS walk of lanes_c
// work-items was 1.2 us of a 6.2 us workgroup life.)
template <typename V>                                     // V: fd_f32x4 or fd_f32x8 (NV floats per channel group)
__device__ __forceinline__ bool fd_wg_sum_by_channel_group(V &a, V &b, float *red, int lanes_c, int tid)
{
    constexpr int NV = sizeof(V) / sizeof(float);
    for (int m = lanes_c; m < 64; m <<= 1) {
#pragma unroll
        for (int j = 0; j < NV; ++j) { a[j] += __shfl_xor(a[j], m); b[j] += __shfl_xor(b[j], m); }
    }
    const int wave = tid >> 6, lane = tid & 63;
    __syncthreads();
    if (lane < lanes_c) {
#pragma unroll
        for (int j = 0; j < NV; ++j) { red[(wave * lanes_c + lane) * 2 * NV + j] = a[j]; red[(wave * lanes_c + lane) * 2 * NV + NV + j] = b[j]; }
    }
    __syncthreads();
    if (tid >= lanes_c) return false;
#pragma unroll
    for (int j = 0; j < NV; ++j) { a[j] = red[tid * 2 * NV + j]; b[j] = red[tid * 2 * NV + NV + j]; }
#pragma unroll
    for (int w = 1; w < 4; ++w)
#pragma unroll
        for (int j = 0; j < NV; ++j) { a[j] += red[(w * lanes_c + tid) * 2 * NV + j]; b[j] += red[(w * lanes_c + tid) * 2 * NV + NV + j]; }
    return true;
}

// The same sum, handed to the unit's statistics rows: the totals of the workgroup's CB = lanes_c * NV channels [c0, c0 + CB) go through LDS (red + 8 * CB
// floats ... + 10 * CB) so that CONSECUTIVE lanes add CONSECUTIVE channels -- the memory side serialises atomics per 128-byte line and instruction
// (~22 ns each, tools/microbench/stat_atomics.hip), so a wave instruction should cover whole lines: 2 * CB * 8 bytes in CB / 8 line operations instead of
// one line operation per lane and channel (measured: the register-window kernels ran 2 - 4.5x longer with one atomic instruction per channel of a lane).
template <int DIR, typename V>
__device__ __forceinline__ void fd_wg_stat_add(V &a, V &b, float *red, int lanes_c, int tid, const fd_stat_rows &sr, long blk, int C, int c0)
{
    constexpr int NV = sizeof(V) / sizeof(float);
    const int CB = lanes_c * NV;
    float *tot = red + 8 * CB;                                // [2][CB], behind the 4 x lanes_c x 2 x NV floats of the wave sums
    if (fd_wg_sum_by_channel_group(a, b, red, lanes_c, tid)) {
#pragma unroll
        for (int j = 0; j < NV; ++j) { tot[tid * NV + j] = a[j]; tot[CB + tid * NV + j] = b[j]; }
    }
    __syncthreads();
    for (int t = tid; t < 2 * CB; t += 256) {
        const int which = t >= CB ? 1 : 0, ch = t - which * CB;
        if (c0 + ch < C) fd_stat_add<DIR>(sr, blk, C, which, c0 + ch, tot[t]);
    }
}

// ------------------------------------------------------------------------------------------------
// Depthwise 3x3, stride S, train mode, register-window variant (fd_dw3_rows' design: no LDS staging, no barrier before the
// statistics) for the large maps with a power-of-two channel-group count C/4 in 8 ... 64: a work-item is q = x * (C/4) + c4 (one
// 16-byte channel group of one output column) and walks down TH output rows keeping the 3 x 3 window of the ACTIVATED input
// act1(z_in * s1 + t1) in registers -- per output row it loads and activates the 3*S new vectors; the horizontal neighbours are the words its
// neighbour work-items load (L1 / L2 hits).  Raw weights w[C][9]; output raw z (rounded to T) + this workgroup's per-channel sums of the
// rounded values, added to the unit's statistics rows (blk = (image * gridDim.y + strip) * gridDim.x + column block).  fin.rows != null: the
// PRODUCER's BatchNorm is finalised here (C <= 256 channels: work-item c finalises channel c into LDS; workgroup (0, 0, 0) is the writer).
// ------------------------------------------------------------------------------------------------
template <typename T, int S, int ACT1>
__global__ void __launch_bounds__(256)
fd_dw3_rows_train(const T *__restrict__ zin, const float *__restrict__ st1, const float *__restrict__ w, T *__restrict__ zout,
                  fd_stat_rows sr, int H, int W, int Ho, int Wo, int C, int TH, fd_bn_fin fin)
{
    __shared__ float red[4 * 64 * 8 + 2 * 256];              // wave sums + the workgroup's totals (fd_wg_stat_add)
    const int CG = C >> 2;                                   // a power of two, 8 <= CG <= 64 (plan)
    const fd_blk3 blk = fd_xcd_image_map();
    const int tid = threadIdx.x;
    const int q = blk.x * 256 + tid;
    const bool live = q < Wo * CG;
    const int qq = live ? q : 0;
    const int xo = qq / CG, c4 = qq - xo * CG;               // (256 % CG == 0: c4 == tid & (CG - 1) for every work-item, live or not)
    const int n = blk.z;
    const int oy0 = blk.y * TH;
    const int oy1 = (oy0 + TH < Ho) ? oy0 + TH : Ho;
    fd_f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { wv[t].x = w[(c4 * 4 + 0) * 9 + t]; wv[t].y = w[(c4 * 4 + 1) * 9 + t]; wv[t].z = w[(c4 * 4 + 2) * 9 + t]; wv[t].w = w[(c4 * 4 + 3) * 9 + t]; }
    fd_f32x4 sc, sh;
    if (fin.rows) {                                          // (scale, shift) of all C <= 256 channels -> red[0 .. 2C) -> this work-item's four
        fd_stat_table_all<256>(fin, C, C, tid, blk.x == 0 && blk.y == 0 && blk.z == 0, [&](int c, float a, float b) { red[c] = a; red[C + c] = b; });
        __syncthreads();
        sc = fd_ld4(red + c4 * 4); sh = fd_ld4(red + C + c4 * 4);
        __syncthreads();                                     // red is the statistics scratch again
    } else { sc = fd_ld4(st1 + FD_ST_SCALE * C + c4 * 4); sh = fd_ld4(st1 + FD_ST_SHIFT * C + c4 * 4); }
    const T *img = zin + (long)n * H * W * C + c4 * 4;
    const int x0 = xo * S - 1;                               // leftmost input column of the window
    const bool okl = x0 >= 0, okr = (x0 + 2) < W;            // the centre column x0 + 1 is always inside
    const int xl = okl ? x0 : x0 + 1, xr = okr ? x0 + 2 : x0 + 1;   // clamped: the three loads are always issued, the padding is a select
    auto load_row = [&](int iy, fd_f32x4 &l, fd_f32x4 &c, fd_f32x4 &r) {
        const bool oky = iy >= 0 && iy < H;
        const int qy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        const T *p = img + (long)qy * W * C;
        const fd_f32x4 vl = fd_ld4(p + (long)xl * C), vc = fd_ld4(p + (long)(x0 + 1) * C), vr = fd_ld4(p + (long)xr * C);
        l = (oky && okl) ? fd_bn_act4<ACT1>(vl, sc, sh) : fd_zero4();       // zero padding of the ACTIVATED input
        c = oky ? fd_bn_act4<ACT1>(vc, sc, sh) : fd_zero4();
        r = (oky && okr) ? fd_bn_act4<ACT1>(vr, sc, sh) : fd_zero4();
    };
    T *o = zout + (((long)n * Ho + oy0) * Wo) * C + (long)qq * 4;
    fd_f32x4 ssum = fd_zero4(), ssq = fd_zero4();
    fd_f32x4 r0l, r0c, r0r, r1l, r1c, r1r, r2l, r2c, r2r;
    load_row(S * oy0 - 1, r0l, r0c, r0r);
    if (S == 1) load_row(oy0, r1l, r1c, r1r);
    for (int oy = oy0; oy < oy1; ++oy) {
        if (S == 1) load_row(oy + 1, r2l, r2c, r2r);
        else { load_row(2 * oy, r1l, r1c, r1r); load_row(2 * oy + 1, r2l, r2c, r2r); }
        fd_f32x4 acc = r0l * wv[0];
        acc += r0c * wv[1]; acc += r0r * wv[2];
        acc += r1l * wv[3]; acc += r1c * wv[4]; acc += r1r * wv[5];
        acc += r2l * wv[6]; acc += r2c * wv[7]; acc += r2r * wv[8];
        const fd_f32x4 zr = fd_round4(T{}, acc);
        if (live) { fd_st4(o, zr); ssum += zr; ssq += zr * zr; }
        o += (long)Wo * C;
        if (S == 1) { r0l = r1l; r0c = r1c; r0r = r1r; r1l = r2l; r1c = r2c; r1r = r2r; }
        else { r0l = r2l; r0c = r2c; r0r = r2r; }
    }
    fd_wg_stat_add<FD_STAT_FWD>(ssum, ssq, red, CG, tid, sr, ((long)n * gridDim.y + blk.y) * gridDim.x + blk.x, C, 0);
}

// ------------------------------------------------------------------------------------------------
// Depthwise K x K, stride S, train mode (LDS-tiled, same geometry as fd_dwconv).
//   input  = act1(z_in * s1 + t1)                                   (MODE 0)
//          = up2(act1(z_in * s1 + t1))                              (MODE 1)
//          = up2(act1(z_in * s1 + t1)) + act2(z_skip * s2 + t2)     (MODE 2)
// weights are the live parameter w[C][K*K]; output is the raw conv result; the workgroup's per-channel partial sums are added to the unit's
// statistics rows (blk = image * gridDim.x + tile, logical indices: fd_xcd_image_map).
// ------------------------------------------------------------------------------------------------
template <typename T, int K, int S, int MODE, int ACT1, int ACT2, int N>
__global__ void __launch_bounds__(256)
fd_dwconv_train(const T *__restrict__ zin, const float *__restrict__ st1, const T *__restrict__ zskip,
                    const float *__restrict__ st2, const float *__restrict__ w, T *__restrict__ zout,
                    fd_stat_rows sr, int Hin, int Win, int Ho, int Wo, int C, int cbq, int TH, int TW, int tiles_x, int csplit, int pstr,
                    fd_bn_fin fin)
{
    constexpr int P = K / 2;
    constexpr int UNR_TAPROWS = K == 3 ? 3 : 1;           // 3x3: all tap rows of a strip unrolled (their LDS reads in flight together); 5x5: one row at a time (registers)
    constexpr int NIN = 3 * S + K;
    typedef fd_lane<T, N> LN;
    typedef typename LN::vec vec;
    typedef typename LN::lds_t lds_t;
    FD_DYN_SMEM(smem_raw);
    const int lanes_c = 1 << cbq, CB = lanes_c * N, PSTR = pstr;   // LDS patch pitch in LDS elements (a multiple of 16 bytes: chosen by the plan)
    const int TH_in = (TH - 1) * S + K, TW_in = (TW - 1) * S + K;
    lds_t *s_in = reinterpret_cast<lds_t *>(smem_raw);    // [TH_in*TW_in][PSTR] (>= 8 KiB: reused as fp32 scratch by the stats reduction)
    float *s_w = reinterpret_cast<float *>(smem_raw + fd_lds_patch_bytes<lds_t>(TH_in * TW_in, PSTR));   // [K*K][CB]
    const fd_blk3 bm = fd_xcd_image_map();                 // all tiles / channel blocks of an image on one XCD: halo re-reads hit its L2
    const int ty = bm.x / tiles_x, tx = bm.x - ty * tiles_x;
    const int c0 = bm.y * CB, n = bm.z;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - P, ix0 = ox0 * S - P;
    const int tid = threadIdx.x, c4 = tid & (lanes_c - 1), pt = tid >> cbq, npt = 256 >> cbq;
    FD_DW_PROBE_AT(0);
    const int cg = c0 + c4 * N;
    const bool c_ok = cg < C;

    // the taps of this channel block (w[c][tap] -> s_w[tap][c]): requested now, written to LDS after the patch loads have been issued, so that the two
    // global round trips overlap instead of following each other (a workgroup's life is a chain of such latencies, not arithmetic)
    constexpr int NWREG = (K * K * 8 * N + 255) / 256;     // CB <= 8 * N channels
    float wreg[NWREG];
#pragma unroll
    for (int j = 0; j < NWREG; ++j) {
        const int i = tid + 256 * j, t = i / CB, cc = i - t * CB;
        wreg[j] = (i < K * K * CB && c0 + cc < C) ? w[(long)(c0 + cc) * K * K + t] : 0.0f;
    }
    // MODE 3 (channel concatenation cat(up2(a_in), a_skip), MobileNetSkipConcat): channels [0, csplit) come from the low-resolution
    // producer (pitch csplit), the rest from the skip tensor (pitch C - csplit); a lane's 4 channels never straddle
    const bool from_skip = MODE == 3 && cg >= csplit;
    const int C1 = MODE == 3 ? csplit : C, C2 = MODE == 3 ? C - csplit : C, cl = from_skip ? cg - csplit : cg;
    vec s1 = LN::zero(), t1 = LN::zero(), s2 = LN::zero(), t2 = LN::zero();
    const int npx_in = TH_in * TW_in;
    // producer finalised here (MODE 0..2: its table is per channel of this tensor): the (scale, shift) of this block's channels land at the END of the dynamic LDS
    // (requested AFTER the first batch of patch loads, below: the partial rows' round trip runs under the patch's)
    const bool fin_here = MODE != 3 && fin.rows != nullptr;
    if (c_ok) {
        if (from_skip) { s1 = LN::ldf(st2 + FD_ST_SCALE * C2 + cl); t1 = LN::ldf(st2 + FD_ST_SHIFT * C2 + cl); }
        else if (!fin_here) { s1 = LN::ldf(st1 + FD_ST_SCALE * C1 + cl); t1 = LN::ldf(st1 + FD_ST_SHIFT * C1 + cl); }
        if (MODE == 2) { s2 = LN::ldf(st2 + FD_ST_SCALE * C + cg); t2 = LN::ldf(st2 + FD_ST_SHIFT * C + cg); }
    }
    constexpr int U = K == 3 ? FD_DW_U3 : 8;            // patch pixels in flight per work-item (3x3 whole-frame tiles: 16 x 18 pixels over 32 pixel-threads = 9 each)
    fd_px_walk wk(pt, npt, TW_in);
    for (int base = pt; base < npx_in || (fin_here && base == pt); base += npt * U) {      // (every work-item runs the first batch when it carries the finalisation's barriers)
        vec v[U], sk[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = base + u * npt;
            const int iy = wk.iy, ix = wk.ix;
            wk.next();
            const int gy = iy0 + iy, gx = ix0 + ix;
            sk[u] = LN::zero();
            ok[u] = px < npx_in && c_ok && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
            // branch-free staging: the address is clamped into the image (and the channel group into the tensor) so that every
            // lane issues its load unconditionally -- all U (x2) loads go out back to back; padding is applied when the value is used
            const int qy = gy < 0 ? 0 : (gy >= Hin ? Hin - 1 : gy), qx = gx < 0 ? 0 : (gx >= Win ? Win - 1 : gx);
            const int ql = c_ok ? cl : 0, qg = c_ok ? cg : 0;
            if (MODE == 0) {
                v[u] = FD_DW_ABL(2) ? LN::zero() : LN::ld(zin + fd_nhwc(n, Hin, qy, Win, qx, C, qg));
            } else {
                const int Hs = Hin >> 1, Ws = Win >> 1;
                if (from_skip) v[u] = LN::ld(zskip + fd_nhwc(n, Hin, qy, Win, qx, C2, ql));
                else v[u] = LN::ld(zin + fd_nhwc(n, Hs, (qy >> 1), Ws, (qx >> 1), C1, ql));
                if (MODE == 2) sk[u] = LN::ld(zskip + fd_nhwc(n, Hin, qy, Win, qx, C, qg));
            }
        }
        if (fin_here && base == pt) {
            float *s_st = s_w + K * K * CB;
            fd_stat_table_block(fin, reinterpret_cast<double *>(smem_raw), s_st, c0, CB, C, tid, bm.x == 0 && bm.z == 0);
            if (c_ok) { s1 = LN::ldf(s_st + c4 * N); t1 = LN::ldf(s_st + CB + c4 * N); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = base + u * npt;
            if (px < npx_in) {
                vec a = LN::zero();                   // zero padding applies to the ACTIVATED tensor
                if (ok[u]) {
                    a = from_skip ? fd_bn_act4<ACT2>(v[u], s1, t1) : fd_bn_act4<ACT1>(v[u], s1, t1);
                    if (MODE == 2) a += fd_bn_act4<ACT2>(sk[u], s2, t2);
                }
                LN::lds_st(s_in + px * PSTR + c4 * N, a);
            }
        }
    }
    FD_DW_PROBE_AT(1);
#pragma unroll
    for (int j = 0; j < NWREG; ++j) { const int i = tid + 256 * j; if (i < K * K * CB) s_w[i] = wreg[j]; }
    __syncthreads();
    FD_DW_PROBE_AT(2);

    const int TWS = TW >> 2, nstrips = TH * TWS;
    vec ssum = LN::zero(), ssq = LN::zero();
    for (int s = pt; s < nstrips; s += npt) {
        const int oy = s / TWS, ox = (s - oy * TWS) * 4;
        vec acc[4] = {LN::zero(), LN::zero(), LN::zero(), LN::zero()};
#pragma unroll UNR_TAPROWS                              // 3x3: all 3 x NIN patch reads of a strip in flight at once (probe: the tap phase was a chain of LDS latencies)
        for (int ky = 0; ky < K; ++ky) {
            const lds_t *row = s_in + ((oy * S + ky) * TW_in + ox * S) * PSTR + c4 * N;
            vec r[NIN];
#pragma unroll
            for (int i = 0; i < NIN; ++i) r[i] = LN::lds_ld(row + i * PSTR);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const vec wv = LN::ldf(s_w + (ky * K + kx) * CB + c4 * N);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += r[j * S + kx] * wv;
            }
        }
        const int gy = oy0 + oy;
        if (c_ok && gy < Ho) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gx = ox0 + ox + j;
                if (gx < Wo) {
                    const vec zr = LN::round(acc[j]);
                    if (!FD_DW_ABL(1)) LN::st(zout + fd_nhwc(n, Ho, gy, Wo, gx, C, cg), zr);
                    ssum += zr; ssq += zr * zr;
                }
            }
        }
    }
    // workgroup partial statistics: fixed-order sum over the pixel-threads that share a channel group
    FD_DW_PROBE_AT(3);
    fd_wg_stat_add<FD_STAT_FWD>(ssum, ssq, reinterpret_cast<float *>(smem_raw), lanes_c, tid, sr, (long)bm.z * gridDim.x + bm.x, C, c0);
    FD_DW_PROBE_AT(4);
    FD_DW_PROBE_AT(5);
}

// ------------------------------------------------------------------------------------------------
// Pointwise GEMM, train mode:  z[M][N] = act1(zin[M][K] * s1[K] + t1[K]) * W[N][K]^T   (W = live parameter)
// Same LDS-DMA / swizzle / 3-stage-ring structure as fd_pw_gemm_f32; the BatchNorm + activation of the PRODUCER
// is applied to the A fragments after the ds_read (a table of scale/shift per k sits in LDS).  The table is zero
// for k >= K, which also neutralises a ragged last K tile (the W source chunk is clamped to finite data).
// Epilogue: raw z + per-column partial statistics, added to the unit's statistics rows (row chosen by mt).  fin.rows != null: the producer's
// BatchNorm is finalised here -- every workgroup derives the whole [2][K] table after its first LDS-DMA stages are issued; workgroup 0 is the writer.
// ------------------------------------------------------------------------------------------------
template <int ACT1>
__global__ void __launch_bounds__(256)
fd_pw_gemm_train_f32(const float *__restrict__ A, const float *__restrict__ st1, const float *__restrict__ Wt,
                     float *__restrict__ out, fd_stat_rows sr, int M, int N, int K, int m_tiles, int n_tiles, fd_bn_fin fin)
{
    constexpr int BM = 64, BN = 64, BK = 32, ROWS = BM + BN, STAGE = ROWS * BK, RG = ROWS / 8 / 4;
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);      // 3 stages, then the scale/shift table [2][K32], then [2][2][64] stats
    const int K32 = (K + 31) / 32 * 32;
    float *tab = smem + (K32 < 3 * BK ? K32 / BK : 3) * STAGE;   // behind the ring's min(3, K tiles) stages
    float *red = tab + 2 * K32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % n_tiles, mt = (slot / n_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    // the producer's table is requested first and lands in LDS after the first LDS-DMA stages have been issued: one round trip instead of two
    // at the head of every workgroup
    constexpr int TABQ = 4;                                  // K <= 1024 (checked by the plan)
    float tsv[TABQ], ttv[TABQ];
    if (!fin.rows) {
#pragma unroll
        for (int i = 0; i < TABQ; ++i) {
            const int k = tid + 256 * i;
            const int kc = k < K ? k : 0;
            const float a = st1[FD_ST_SCALE * K + kc], b = st1[FD_ST_SHIFT * K + kc];
            tsv[i] = k < K ? a : 0.0f; ttv[i] = k < K ? b : 0.0f;
        }
    }
    const float *src[RG];
    int src_chunk[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        const int r = (wave + 4 * i) * 8 + (lane >> 3);
        src_chunk[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 4;
        if (r < BM) { long row = m0 + r; if (row > M - 1) row = M - 1; src[i] = A + row * K; }
        else { int row = n0 + (r - BM); if (row > N - 1) row = N - 1; src[i] = Wt + (long)row * K; }
    }
    auto issue = [&](int t) {
        float *dst = smem + (t % 3) * STAGE + wave * 8 * BK;
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            int k = t * BK + src_chunk[i];
            if (k >= K) k = 0;
            fd_glds16(src[i] + k, dst + i * 4 * 8 * BK);
        }
    };
    fd_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int h = lane >> 5;
    int a_off[4], b_off[4];
    {
        const int ra = wm * 32 + (lane & 31), rb = BM + wn * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a_off[g] = ra * BK + (((2 * g + h) ^ ((ra >> 1) & 7)) << 2);
            b_off[g] = rb * BK + (((2 * g + h) ^ ((rb >> 1) & 7)) << 2);
        }
    }
    const int T = K32 / BK;
    issue(0);
    if (T > 1) issue(1);
    if (fin.rows) fd_stat_table_all<256>(fin, K, K32, tid, blockIdx.x == 0, [&](int k, float a, float b) { tab[k] = a; tab[K32 + k] = b; });
    else {
#pragma unroll
        for (int i = 0; i < TABQ; ++i) {
            const int k = tid + 256 * i;
            if (k < K32) { tab[k] = tsv[i]; tab[K32 + k] = ttv[i]; }
        }
    }
    fd_block_barrier_lds();                                  // scale/shift table visible
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) fd_wait_vmcnt<RG>(); else fd_wait_vmcnt<0>();
        fd_block_barrier();
        if (t + 2 < T) issue(t + 2);
        const float *cur = smem + (t % 3) * STAGE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kb = t * BK + (2 * g + h) * 4;
            const fd_f32x4 a = fd_bn_act4<ACT1>(fd_ld4(cur + a_off[g]), fd_ld4(tab + kb), fd_ld4(tab + K32 + kb));
            const fd_f32x4 b = fd_ld4(cur + b_off[g]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
        }
    }
    // epilogue: raw z and per-column statistics over this tile's valid rows
    const int col = n0 + wn * 32 + (lane & 31);
    const long rbase = m0 + wm * 32 + 4 * (lane >> 5);
    float s = 0.0f, q = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < M && col < N) { out[row * N + col] = acc[r]; s += acc[r]; q = fmaf(acc[r], acc[r], q); }
    }
    s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);          // the two half-waves hold different rows of the same column
    __syncthreads();
    if (lane < 32) { red[(wm * 2 + 0) * 64 + wn * 32 + lane] = s; red[(wm * 2 + 1) * 64 + wn * 32 + lane] = q; }
    __syncthreads();
    if (tid < 64 && n0 + tid < N) {
        fd_stat_add<FD_STAT_FWD>(sr, mt, N, 0, n0 + tid, red[0 * 64 + tid] + red[2 * 64 + tid]);
        fd_stat_add<FD_STAT_FWD>(sr, mt, N, 1, n0 + tid, red[1 * 64 + tid] + red[3 * 64 + tid]);
    }
}

// ------------------------------------------------------------------------------------------------
// Head, train mode: z_low[p] = sum_k act1(zin[p][k]*s1[k]+t1[k]) * w[k]  at the low resolution (the nearest-x2
// upsampling commutes with the 1x1 conv; BN statistics over the replicated tensor equal the low-res ones, the
// unbiased correction uses the full-resolution count -- SURVEY.md Appendix F).  The workgroup's (sum, sum of squares) go to the head's 1-channel
// statistics rows.  fin.rows != null: the producer's BatchNorm (Cin <= FD_HEAD_FIN_MAX channels) is finalised here, workgroup 0 is the writer.
// ------------------------------------------------------------------------------------------------
#define FD_HEAD_FIN_MAX 256
template <typename T, int ACT1>
__global__ void __launch_bounds__(256)
fd_head_train(const T *__restrict__ zin, const float *__restrict__ st1, const float *__restrict__ w,
                  float *__restrict__ zlow, fd_stat_rows sr, long npix, int Cin, fd_bn_fin fin)
{
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float s_tab[2 * FD_HEAD_FIN_MAX];
    const float *tsc = st1 + FD_ST_SCALE * Cin, *tsh = st1 + FD_ST_SHIFT * Cin;
    if (fin.rows) {
        fd_stat_table_all<256>(fin, Cin, Cin, threadIdx.x, blockIdx.x == 0, [&](int c, float a, float b) { s_tab[c] = a; s_tab[FD_HEAD_FIN_MAX + c] = b; });
        __syncthreads();
        tsc = s_tab; tsh = s_tab + FD_HEAD_FIN_MAX;
    }
    // a workgroup walks pixel groups of 32 (8 lanes per pixel) grid-stride: the plan bounds the grid, so that the head's ONE channel sees at most
    // ~1024 additions to its statistics rows (12544 workgroups of 32 pixels each cost 47 us of serialised atomics at batch 32)
    const int l8 = threadIdx.x & 7;
    float v = 0.0f, sq = 0.0f;
    for (long base = (long)blockIdx.x * 32; base < npix; base += (long)gridDim.x * 32) {      // (workgroup-uniform trip count: the shuffles below are wave-wide)
        const long g = base + (threadIdx.x >> 3);
        const bool live = g < npix;
        const long gq = live ? g : npix - 1;
        float s = 0.0f;
        for (int c = l8 * 4; c < Cin; c += 32) {
            const fd_f32x4 a = fd_bn_act4<ACT1>(fd_ld4(zin + gq * Cin + c), fd_ld4(tsc + c), fd_ld4(tsh + c));
            const fd_f32x4 q = fd_ld4(w + c);
            s = fmaf(a.x, q.x, s); s = fmaf(a.y, q.y, s); s = fmaf(a.z, q.z, s); s = fmaf(a.w, q.w, s);
        }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        if (live && l8 == 0) { zlow[g] = s; v += s; sq = fmaf(s, s, sq); }
    }
    for (int m = 8; m < 64; m <<= 1) { v += __shfl_xor(v, m); sq += __shfl_xor(sq, m); }   // lanes with l8 != 0 contribute 0
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave * 2] = v; red[wave * 2 + 1] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        fd_stat_add<FD_STAT_FWD>(sr, blockIdx.x, 1, 0, 0, red[0] + red[2] + red[4] + red[6]);
        fd_stat_add<FD_STAT_FWD>(sr, blockIdx.x, 1, 1, 0, red[1] + red[3] + red[5] + red[7]);
    }
}

// prediction of the train-mode forward: pred = act(z_low*s + t), written as 2x2 blocks (up == 1) or 1:1.  fin.rows != null: the head's own
// 1-channel BatchNorm is finalised here (work-item 0 of every workgroup derives (s, t); workgroup 0 is the writer)
template <int ACT>
__global__ void __launch_bounds__(256)
fd_head_apply_f32(const float *__restrict__ zlow, const float *__restrict__ st, float *__restrict__ y, long npix, int h, int w, int up, fd_bn_fin fin)
{
    __shared__ float s_st[2];
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const float zq = zlow[g < npix ? g : npix - 1];
    float sc, sf;
    if (fin.rows) {
        fd_stat_table_all<256>(fin, 1, 1, threadIdx.x, blockIdx.x == 0, [&](int, float a, float b) { s_st[0] = a; s_st[1] = b; });
        __syncthreads();
        sc = s_st[0]; sf = s_st[1];
    } else { sc = st[FD_ST_SCALE]; sf = st[FD_ST_SHIFT]; }
    if (g >= npix) return;
    const float v = fd_act<ACT>(zq * sc + sf);
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
