// fd_kernels_bwd.h -- backward kernels of the FastDepth train step (gfx950).  T = storage type of the saved z and of the
// activation gradients G (float or fd_bf16); arithmetic, reductions, tables and parameter gradients are fp32.
//
// The train step is not in the reference tree; these kernels implement the autograd of the
// reference's forward (models.py:706-732 in .train()) by hand:
//   unit i:  z_i = conv_i(in_i),  y_i = z_i*s_i + t_i  (batch-stat BN),  a_i = act_i(y_i)
//   saved by the forward: z_i and the table st_i = (s, t, mean, invstd).
//   G_i  := dLoss/dy_i  (gradient AFTER the activation mask), produced by the backward-data kernel of the consumer,
//           which also emits the per-channel partial sums  sum(G_i), sum(G_i * xhat_i)   (xhat = (z-mean)*invstd).
//   fd_bn_bwd_finalize_f32: dbeta = sum(G), dgamma = sum(G*xhat)  -> parameter grads, and the coefficient table
//           dz_i = A*((G_i - C1) - (z_i - mean)*C2)   with  A = s,  C1 = dbeta/n,  C2 = invstd*dgamma/n
//           (the BatchNorm backward  dz = s*(G - dbeta/n - xhat*dgamma/n)  as a per-channel map of (G, z)).
//   The conv-backward kernels form dz_i ON LOAD from (G_i, z_i) and re-create their forward input
//   a_{i-1} = act(z_{i-1}*s+t) (+ nearest-x2 / skip composition) on load as well: no normalised, activated,
//   upsampled or BN-backward tensor is ever written to HBM.
//   Activation masks are strict (ReLU: y > 0; ReLU6: 0 < y < 6), as torch's threshold/hardtanh backward.
// All reductions are two-stage with a fixed summation order (deterministic).
#pragma once
#include "fd_kernels_train.h"

// coefficient table [4][C] of the BatchNorm backward, written in the cancellation-free form
//   dz = A * ((G - C1) - (z - MU) * C2),   A = gamma*invstd,  C1 = dbeta/n,  MU = batch mean,  C2 = invstd * dgamma / n
#ifndef FD_BWD_STAGES
#define FD_BWD_STAGES 2     // LDS ring depth of the backward GEMMs: 2 stages = 48 KiB -> 3 workgroups per CU (measured faster than 3 stages / 2 per CU)
#endif
#define FD_CF_A 0
#define FD_CF_C1 1
#define FD_CF_MU 2
#define FD_CF_C2 3
__device__ __forceinline__ float fd_dz(float g, float z, float cA, float c1, float mu, float c2) { return cA * ((g - c1) - (z - mu) * c2); }
template <typename V> __device__ __forceinline__ V fd_dz4(V g, V z, V cA, V c1, V mu, V c2) { return cA * ((g - c1) - (z - mu) * c2); }   // V: fd_f32x4 or fd_f32x8

template <int ACT>
__device__ __forceinline__ float fd_actmask(float y)
{
    if (ACT == FD_ACT_RELU_) return y > 0.0f ? 1.0f : 0.0f;
    if (ACT == FD_ACT_RELU6_) return (y > 0.0f && y < 6.0f) ? 1.0f : 0.0f;
    return 1.0f;
}
template <int ACT>
__device__ __forceinline__ fd_f32x4 fd_actmask4(fd_f32x4 y)
{
    fd_f32x4 r = {fd_actmask<ACT>(y.x), fd_actmask<ACT>(y.y), fd_actmask<ACT>(y.z), fd_actmask<ACT>(y.w)};
    return r;
}
template <int ACT>
__device__ __forceinline__ fd_f32x8 fd_actmask4(fd_f32x8 y)
{
    fd_f32x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = fd_actmask<ACT>(y[j]);
    return r;
}

// ---- BatchNorm backward finalisation: the unit's statistics rows (sum G, sum G*xhat -- added by its consumer's backward-data kernel, fd_stat_add)
// -> dbeta, dgamma and the coefficient table of dz.  Either inside the unit's own first backward kernel (fd_bstat_table_block: every workgroup
// derives the coefficients of ITS channels from the same integers -- identical bits --, the designated workgroup also writes the parameter
// gradients and the table) or as a launch of its own (fd_bn_bwd_finalize_rows_f32: one work-item per channel).
struct fd_bn_bwd_fin {
    const long long *rows;           // null: the table `coef` was finalised by its own launch
    int nr, cs;                      // statistics rows (power of two), their channel pitch
    int cf_off;                      // byte offset of s_cf in the kernel's dynamic LDS (the plan appends it to the kernel's own request)
    double inv_n;                    // 1 / (pixels per channel)
    const float *st;                 // the unit's forward table
    float *dgamma, *dbeta, *coef;
};
struct fd_bn_bwd_coef { float a, c1, mu, c2; };
__device__ __forceinline__ fd_bn_bwd_coef fd_bn_bwd_coef_of(double s, double q, double inv_n, float sc_c, float mean_c, float invstd_c)
{
    fd_bn_bwd_coef k;
    k.a = sc_c; k.c1 = (float)(s * inv_n); k.mu = mean_c; k.c2 = (float)((double)invstd_c * q * inv_n);
    return k;
}
__device__ __forceinline__ void fd_bn_bwd_publish(const fd_bn_bwd_fin &f, int C, int c, double s, double q, const fd_bn_bwd_coef &k)
{
    f.dbeta[c] = (float)s;
    f.dgamma[c] = (float)q;
    f.coef[FD_CF_A * C + c] = k.a; f.coef[FD_CF_C1 * C + c] = k.c1; f.coef[FD_CF_MU * C + c] = k.mu; f.coef[FD_CF_C2 * C + c] = k.c2;
}
// Channel-block form (the backward mirror of fd_stat_table_block, fd_kernels_train.h): 256 work-items, the CB <= 64 channels [c0, c0 + CB).
// sh: >= 4 KiB of LDS that is dead until the next barrier; s_cf: [4][CB] floats (A, C1, MU, C2) that nothing else touches.
__device__ __forceinline__ void fd_bstat_table_block(const fd_bn_bwd_fin &f, double *sh, float *s_cf, int c0, int CB, int C, int tid, bool writer)
{
    const int ch = tid & (CB - 1), rg = tid / CB, RG = 256 / CB;
    const int c = c0 + ch;
    const bool ok = c < C;
    float sc_c = 0.0f, mean_c = 0.0f, invstd_c = 0.0f;
    if (rg == 0 && ok) { sc_c = f.st[FD_ST_SCALE * C + c]; mean_c = f.st[FD_ST_MEAN * C + c]; invstd_c = f.st[FD_ST_INVSTD * C + c]; }
    double s = 0.0, q = 0.0;
    if (ok && rg < f.nr) {
        s = fd_stat_total<FD_STAT_BWD>(f.rows, f.nr, f.cs, 0, c, rg, RG);
        q = fd_stat_total<FD_STAT_BWD>(f.rows, f.nr, f.cs, 1, c, rg, RG);
    }
    sh[2 * tid] = s; sh[2 * tid + 1] = q;
    __syncthreads();
    if (rg == 0) {
        s = 0.0; q = 0.0;
        for (int r = 0; r < RG; ++r) { s += sh[2 * (r * CB + ch)]; q += sh[2 * (r * CB + ch) + 1]; }
        fd_bn_bwd_coef k = fd_bn_bwd_coef_of(s, q, f.inv_n, sc_c, mean_c, invstd_c);
        if (!ok) { k.a = 0.0f; k.c1 = 0.0f; k.mu = 0.0f; k.c2 = 0.0f; }
        s_cf[FD_CF_A * CB + ch] = k.a; s_cf[FD_CF_C1 * CB + ch] = k.c1; s_cf[FD_CF_MU * CB + ch] = k.mu; s_cf[FD_CF_C2 * CB + ch] = k.c2;
        if (writer && ok) fd_bn_bwd_publish(f, C, c, s, q, k);
    }
    __syncthreads();
}

static __global__ void __launch_bounds__(256)
fd_bn_bwd_finalize_rows_f32(const fd_bn_bwd_fin f, int C)
{
    __shared__ double sh[512];
    __shared__ float s_cf[4 * 16];
    fd_bstat_table_block(f, sh, s_cf, blockIdx.x * 16, 16, C, threadIdx.x, true);      // 16 channels per workgroup, a channel's rows dealt to 16 work-items
}

// ---- weight-gradient partials of a whole range of units, reduced by ONE launch at the end of the range (round 1: one launch per unit paired
// with its BatchNorm-backward finalisation, 36 x ~10 us per step).  Every unit's weight-gradient kernel leaves its partial rows in its
// own region; entry e describes one unit:  out[j] = sum_b part[b*n + j]  (KK == 0), or the depthwise form
// out[c*KK + t] = sum_b part[(b*KK + t)*C + c]  (tap-major partials -> torch's [C][1][k][k]) with n = KK*C, n % 4 == 0.
// Two shapes of work in one kernel (uniform per workgroup):
//   few rows (nblk <= 64: the pointwise layers' M splits)   each of the 16 waves owns 256 columns and adds ALL rows itself, 8 row loads
//                                                            in flight -- a pure stream, no LDS, no barrier (4096 columns per workgroup);
//   many rows (depthwise / stem partial rows, up to ~3000)   64 columns per workgroup, the rows dealt to 4 x 16 row lanes (shuffle across the four of
//                                                            a wave, then the 16 wave sums in wave order).
// Fixed orders: bit-reproducible.
#define FD_WBATCH_MAX 40
#define FD_WBATCH_FEW_ROWS 64
struct fd_wred_args { const float *part; float *out; int nblk, n, KK, C; };
struct fd_wbatch { int count; int cb_start[FD_WBATCH_MAX + 1]; fd_wred_args e[FD_WBATCH_MAX]; };
__device__ __forceinline__ void fd_wbatch_store(const fd_wred_args &W, int j, fd_f32x4 s)
{
    if (W.KK == 0) { W.out[j] = s.x; W.out[j + 1] = s.y; W.out[j + 2] = s.z; W.out[j + 3] = s.w; }   // (gradient views need not be 16-byte aligned)
    else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int t = (j + q) / W.C, c = (j + q) - t * W.C; W.out[(long)c * W.KK + t] = s[q]; }
    }
}
static __global__ void __launch_bounds__(1024)
fd_reduce_weights_batch_f32(const fd_wbatch B)
{
    __shared__ fd_f32x4 sh[16][64];
    int e = 0;
    while (e + 1 < B.count && (int)blockIdx.x >= B.cb_start[e + 1]) ++e;
    const fd_wred_args W = B.e[e];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = (int)blockIdx.x - B.cb_start[e];
    if (W.nblk <= FD_WBATCH_FEW_ROWS) {
        const int j = ((cb * 16 + wave) * 64 + lane) * 4;
        if (j >= W.n) return;
        const float *p = W.part + j;
        fd_f32x4 s = fd_zero4();
        int b = 0;
        for (; b + 8 <= W.nblk; b += 8) {
            fd_f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = fd_ld4(p + (long)(b + u) * W.n);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; b < W.nblk; ++b) s += fd_ld4(p + (long)b * W.n);
        fd_wbatch_store(W, j, s);
        return;
    }
    // many rows: 64 columns per workgroup (16 lanes x 4), the rows dealt to 64 row lanes (4 per wave x 16 waves: row lane r takes rows r, r + 64, ...,
    // eight loads in flight) -- a 3136-row entry (fd_dw_bwd1 of conv2: one row per input-space tile) is 49 rows deep per lane instead of 196
    const int cl = lane & 15, rsub = lane >> 4;
    const int j = (cb * 16 + cl) * 4;
    fd_f32x4 s = fd_zero4();
    if (j < W.n) {
        const float *p = W.part + j;
        int b = wave * 4 + rsub;
        for (; b + 7 * 64 < W.nblk; b += 8 * 64) {
            fd_f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = fd_ld4(p + (long)(b + 64 * u) * W.n);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; b < W.nblk; b += 64) s += fd_ld4(p + (long)b * W.n);
    }
    // the four row lanes of a wave (fixed order), then the 16 waves in wave order
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) { s.x += __shfl_xor(s.x, m); s.y += __shfl_xor(s.y, m); s.z += __shfl_xor(s.z, m); s.w += __shfl_xor(s.w, m); }
    sh[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && rsub == 0 && j < W.n) {
        s = sh[0][lane];
        for (int w = 1; w < 16; ++w) s += sh[w][lane];
        fd_wbatch_store(W, j, s);
    }
}

// ---- mean-L1 loss forward + backward (torch.nn.L1Loss) --------------------------------------------------------------
static __global__ void __launch_bounds__(256)
fd_l1_loss_f32(const float *__restrict__ pred, const float *__restrict__ target, float *__restrict__ dpred,
               float *__restrict__ part, long numel, float inv_numel)
{
    __shared__ float red[4];
    float s = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        s += fabsf(d);
        dpred[i] = d > 0.0f ? inv_numel : (d < 0.0f ? -inv_numel : 0.0f);
    }
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
static __global__ void __launch_bounds__(64)
fd_l1_loss_final_f32(const float *__restrict__ part, int nblk, float inv_numel, float *__restrict__ loss)
{
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[b];
    float f = (float)s;
    for (int m = 1; m < 64; m <<= 1) f += __shfl_xor(f, m);
    if (threadIdx.x == 0) loss[0] = f * inv_numel;
}

// ---- masked mean-L1 loss (the criterion of the upstream the reference was cut from, README.md:65: sparse-to-dense's MaskedL1Loss --
// valid = target > 0, loss = mean over the VALID pixels of |pred - target|; NYU depth maps hold invalid zeros, the reason metrics.py:32
// masks them too).  Pass 1 leaves per-workgroup (sum |d|, #valid) partials; pass 2: every workgroup adds the <= 1024 partial pairs in the
// same fixed order (so all agree on the count), writes dpred = sign(d) / #valid on the valid pixels and 0 elsewhere; workgroup 0 writes the
// loss (NaN when nothing is valid: the mean of an empty selection, as torch reports it; the gradient is all zeros then).
static __global__ void __launch_bounds__(256)
fd_l1_masked_partial_f32(const float *__restrict__ pred, const float *__restrict__ target, float *__restrict__ part, long numel)
{
    __shared__ float red[4][2];
    float s = 0.0f, c = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long)gridDim.x * 256) {
        const float t = target[i];
        if (t > 0.0f) { s += fabsf(pred[i] - t); c += 1.0f; }
    }
    for (int m = 1; m < 64; m <<= 1) { s += __shfl_xor(s, m); c += __shfl_xor(c, m); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        part[2 * blockIdx.x + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];      // <= 2^24 per workgroup slice: exact in fp32
    }
}
static __global__ void __launch_bounds__(256)
fd_l1_masked_apply_f32(const float *__restrict__ pred, const float *__restrict__ target, const float *__restrict__ part, int nblk,
                       float *__restrict__ dpred, float *__restrict__ loss, long numel)
{
    __shared__ double red[256][2];
    double s = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) { s += (double)part[2 * b]; c += (double)part[2 * b + 1]; }
    red[threadIdx.x][0] = s; red[threadIdx.x][1] = c;
    __syncthreads();
    if (threadIdx.x < 64) {
        for (int k = 1; k < 4; ++k) { s += red[threadIdx.x + 64 * k][0]; c += red[threadIdx.x + 64 * k][1]; }
        red[threadIdx.x][0] = s; red[threadIdx.x][1] = c;
    }
    __syncthreads();
    s = 0.0; c = 0.0;
    for (int k = 0; k < 64; ++k) { s += red[k][0]; c += red[k][1]; }      // same order in every work-item of every workgroup
    const float inv = c > 0.0 ? (float)(1.0 / c) : 0.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) loss[0] = c > 0.0 ? (float)(s / c) : __builtin_nanf("");
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long)gridDim.x * 256) {
        const float t = target[i], d = pred[i] - t;
        dpred[i] = t > 0.0f ? (d > 0.0f ? inv : (d < 0.0f ? -inv : 0.0f)) : 0.0f;
    }
}

// ---- depth metrics (row f-2): every sum of reference metrics.py:31-55 in ONE pass, no per-scalar host synchronisation --------
// sums[0] = #valid, [1] = sum ad^2, [2] = sum ad, [3] = sum |log10 o - log10 t|, [4] = sum ad/t, [5..7] = #(maxRatio < 1.25^k),
// [8] = sum (1/o - 1/t)^2, [9] = sum |1/o - 1/t|;  valid = (target > 0) or (output > 0), o = 1e3*output, t = 1e3*target (mm).
static __global__ void __launch_bounds__(256)
fd_depth_metrics_f32(const float *__restrict__ output, const float *__restrict__ target, long numel, double *__restrict__ part)
{
    // blockIdx.y = frame (numel elements each): the reference evaluates one image at a time (main.py:40-41 batch size 1, :80-82)
    __shared__ double red[4][10];
    double s[10];
    for (int k = 0; k < 10; ++k) s[k] = 0.0;
    output += (long)blockIdx.y * numel; target += (long)blockIdx.y * numel;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long)gridDim.x * 256) {
        const float ov = output[i], tv = target[i];
        if (tv > 0.0f || ov > 0.0f) {
            const float o = 1e3f * ov, t = 1e3f * tv;
            const float ad = fabsf(o - t);
            const float ratio = fmaxf(o / t, t / o);
            const float inv = fabsf(1.0f / o - 1.0f / t);
            s[0] += 1.0; s[1] += (double)(ad * ad); s[2] += ad;
            s[3] += fabsf(logf(o) / 2.302585093f - logf(t) / 2.302585093f);
            s[4] += ad / t;
            s[5] += ratio < 1.25f ? 1.0 : 0.0; s[6] += ratio < 1.5625f ? 1.0 : 0.0; s[7] += ratio < 1.953125f ? 1.0 : 0.0;
            s[8] += (double)(inv * inv); s[9] += inv;
        }
    }
    // fixed-order reduction: lanes -> wave (shuffles on the two 32-bit halves of each double), waves -> block
    for (int k = 0; k < 10; ++k) {
        double v = s[k];
        for (int m = 1; m < 64; m <<= 1) {
            unsigned long long u = __builtin_bit_cast(unsigned long long, v);
            float lo = __builtin_bit_cast(float, (unsigned)(u & 0xffffffffu)), hi = __builtin_bit_cast(float, (unsigned)(u >> 32));
            lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
            const unsigned long long w = ((unsigned long long)__builtin_bit_cast(unsigned, hi) << 32) | __builtin_bit_cast(unsigned, lo);
            v += __builtin_bit_cast(double, w);
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 10) part[((long)blockIdx.y * gridDim.x + blockIdx.x) * 10 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
static __global__ void __launch_bounds__(64)
fd_depth_metrics_final_f32(const double *__restrict__ part, int nblk, double *__restrict__ sums)
{
    if (threadIdx.x < 10) {                                  // blockIdx.x = frame
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += part[((long)blockIdx.x * nblk + b) * 10 + threadIdx.x];
        sums[(long)blockIdx.x * 10 + threadIdx.x] = s;
    }
}

// ---- fused multi-tensor SGD ------------------------------------------------------------------------------------------
struct fd_sgd_rec { float *param; const float *grad; float *buf; long numel; };
static __global__ void __launch_bounds__(256)
fd_sgd_f32(const fd_sgd_rec *__restrict__ table, int n_tensors, float lr, float momentum, float wd, float grad_scale, int first_step)
{
    // blockIdx.y = tensor, blockIdx.x strides over its elements: 16 bytes per lane where the tensor's three arrays are 16-byte aligned
    // (the large ones are: torch allocations and gradient-buffer slices behind multiples of 4 elements), scalar otherwise
    const fd_sgd_rec r = table[blockIdx.y];
    auto upd = [&](float p, float g, float m, float &pn, float &mn) {
        const float d = fmaf(wd, p, grad_scale * g);
        mn = first_step ? d : fmaf(momentum, m, d);
        pn = p - lr * mn;
    };
    const bool vec = (((size_t)r.param | (size_t)r.grad | (size_t)r.buf) & 15) == 0;
    const long n4 = vec ? r.numel >> 2 : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const fd_f32x4 p = fd_ld4(r.param + 4 * i), g = fd_ld4(r.grad + 4 * i), m = first_step ? fd_zero4() : fd_ld4(r.buf + 4 * i);
        float pn[4], mn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) upd(p[q], g[q], m[q], pn[q], mn[q]);
        const fd_f32x4 mv = {mn[0], mn[1], mn[2], mn[3]}, pv = {pn[0], pn[1], pn[2], pn[3]};
        fd_st4(r.buf + 4 * i, mv);
        fd_st4(r.param + 4 * i, pv);
    }
    for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < r.numel; i += (long)gridDim.x * 256) {
        float pn, mn;
        upd(r.param[i], r.grad[i], first_step ? 0.0f : r.buf[i], pn, mn);
        r.buf[i] = mn;
        r.param[i] = pn;
    }
}

// ---- head backward, step 1: G_head (low res) = mask(y) * (sum of the 2x2 block of dLoss/dpred) + BN partials ----------
template <int ACT>
__global__ void __launch_bounds__(256)
fd_head_bwd_reduce_f32(const float *__restrict__ dpred, const float *__restrict__ zlow, const float *__restrict__ st,
                       float *__restrict__ g, fd_stat_rows sr, long npix, int h, int w, int up)
{
    __shared__ float red[8];
    // grid-stride over blocks of 256 low-resolution pixels (the plan bounds the grid: the head's ONE channel takes one addition per workgroup)
    const float st_sc = st[FD_ST_SCALE], st_sh = st[FD_ST_SHIFT], st_mu = st[FD_ST_MEAN], st_is = st[FD_ST_INVSTD];
    float dy = 0.0f, dyx = 0.0f;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const float z = zlow[p];
        const float y = z * st_sc + st_sh;
        float d;
        if (up) {
            const int ox = (int)(p % w);
            const long t = p / w;
            const int oy = (int)(t % h);
            const long n = t / h;
            const float *q = dpred + ((n * 2 * h + 2 * oy) * 2 * (long)w + 2 * ox);
            d = (q[0] + q[1]) + (q[2 * w] + q[2 * w + 1]);
        } else {
            d = dpred[p];
        }
        const float gv = d * fd_actmask<ACT>(y);
        g[p] = gv;
        dy += gv;
        dyx = fmaf(gv, (z - st_mu) * st_is, dyx);
    }
    for (int m = 1; m < 64; m <<= 1) { dy += __shfl_xor(dy, m); dyx += __shfl_xor(dyx, m); }
    if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = dy; red[(threadIdx.x >> 6) * 2 + 1] = dyx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        fd_stat_add<FD_STAT_BWD>(sr, blockIdx.x, 1, 0, 0, red[0] + red[2] + red[4] + red[6]);
        fd_stat_add<FD_STAT_BWD>(sr, blockIdx.x, 1, 1, 0, red[1] + red[3] + red[5] + red[7]);
    }
}

// ---- head backward, step 2: the 1-channel pointwise conv.  per low-res pixel p, channel k:
//   dz = A*G + Bc*z + D (scalars, C == 1);  a_in = act_in(z_in*s+t);  dW[k] += dz*a_in;  dA = dz*w[k];
//   G_in = mask_in(y_in)*dA  (+ BN partials of the producer).  8 lanes share a pixel, PPB pixels per work-item group.
template <typename T, int ACT_IN, int PPB>
__global__ void __launch_bounds__(256)
fd_head_bwd(const float *__restrict__ g, const float *__restrict__ zlow, const float *__restrict__ coef,
                const T *__restrict__ zin, const float *__restrict__ st_in, const float *__restrict__ w,
                T *__restrict__ g_in, fd_stat_rows sr_in, float *__restrict__ wpart, long npix, int Cin)
{
    // work-item (group = tid>>3 in 0..31, l8): channel groups c = l8*4 + 32*j; block covers 32*PPB pixels
    FD_DYN_SMEM(smem_raw);
    float *red = reinterpret_cast<float *>(smem_raw);          // [32][Cin][3]  (sum G, sum G*xhat, dW)
    const int grp = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const float cA = coef[FD_CF_A], c1 = coef[FD_CF_C1], cM = coef[FD_CF_MU], c2 = coef[FD_CF_C2];
    for (int c = l8 * 4; c < Cin; c += 32) {
        const fd_f32x4 s4 = fd_ld4(st_in + FD_ST_SCALE * Cin + c), t4 = fd_ld4(st_in + FD_ST_SHIFT * Cin + c);
        const fd_f32x4 m4 = fd_ld4(st_in + FD_ST_MEAN * Cin + c), i4 = fd_ld4(st_in + FD_ST_INVSTD * Cin + c);
        const fd_f32x4 w4 = fd_ld4(w + c);
        fd_f32x4 sg = fd_zero4(), sgx = fd_zero4(), sw = fd_zero4();
        for (int j = 0; j < PPB; ++j) {
            const long p = ((long)blockIdx.x * PPB + j) * 32 + grp;
            if (p < npix) {
                const float dz = fd_dz(g[p], zlow[p], cA, c1, cM, c2);
                const fd_f32x4 z = fd_ld4(zin + p * Cin + c);
                const fd_f32x4 y = z * s4 + t4;
                const fd_f32x4 gi = fd_round4(T{}, fd_actmask4<ACT_IN>(y) * (w4 * dz));
                fd_st4(g_in + p * Cin + c, gi);
                sg += gi; sgx += gi * ((z - m4) * i4);
                sw += fd_act4<ACT_IN>(y) * dz;
            }
        }
        fd_st4(red + (grp * Cin + c) * 3, sg);           // three float4 interleaved per (grp, c)
        fd_st4(red + (grp * Cin + c) * 3 + 4, sgx);
        fd_st4(red + (grp * Cin + c) * 3 + 8, sw);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cin; c += 256) {
        float a = 0.0f, b = 0.0f, d = 0.0f;
        const int c4 = c & ~3, k = c & 3;
        for (int gI = 0; gI < 32; ++gI) {
            const float *r = red + (gI * Cin + c4) * 3;
            a += r[k]; b += r[4 + k]; d += r[8 + k];
        }
        fd_stat_add<FD_STAT_BWD>(sr_in, blockIdx.x, Cin, 0, c, a);
        fd_stat_add<FD_STAT_BWD>(sr_in, blockIdx.x, Cin, 1, c, b);
        wpart[(long)blockIdx.x * Cin + c] = d;
    }
}

// ------------------------------------------------------------------------------------------------
// Pointwise backward-data GEMM:  dA[M][K] = dz[M][N] * W[N][K],  dz = A*G + Bc*z + D per column n (formed on the
// fragment read), reduction over n.  Epilogue: G_in[m][k] = mask_in(y_in[m][k]) * (dA[m][k] (+ skipgrad[m][k])) and the
// producer's BN partials.  Tiles: 64 (m) x 64 (k), reduction step 32 (n).  LDS per stage: G tile [64][32], z tile
// [64][32] (swizzled 128-byte rows, LDS-DMA) and the W tile [32 n][64 k] (256-byte rows, read as 4-byte fragments:
// consecutive lanes = consecutive k).
// ------------------------------------------------------------------------------------------------
template <int ACT_IN, int ADD_SG>
__device__ __forceinline__ void              // blk: linear workgroup number (blockIdx.x of the plain kernel; the paired launch fd_pw_bwd_f32 passes its own)
fd_pw_dgrad_f32_body(const float *__restrict__ G, const float *__restrict__ Z, const float *__restrict__ coef,
                const float *__restrict__ Wt, const float *__restrict__ Zin, const float *__restrict__ st_in,
                const float *__restrict__ SG, float *__restrict__ Gin, fd_stat_rows sr,
                int M, int N, int K, int m_tiles, int k_tiles, const unsigned blk)
{
    constexpr int BM = 64, BKO = 64, BR = 32;                 // output tile 64 x 64, reduction step 32
    constexpr int STAGE = 2 * BM * BR + BR * BKO;             // floats: G, Z, W tiles
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int N32 = (N + 31) / 32 * 32;
    float *tab = smem + FD_BWD_STAGES * STAGE;                 // [4][N32] coefficient table (zero beyond N)
    float *red = tab + 4 * N32;                                // [2][2][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wk = wave & 1;
    const int xcd = blk & 7, slot = blk >> 3;
    const int kt = slot % k_tiles, mt = (slot / k_tiles) * 8 + xcd;
    if (mt >= m_tiles) return;
    const long m0 = (long)mt * BM;
    const int k0 = kt * BKO;
    // the coefficient table and the epilogue's operands (producer's table, z_in, skip gradient) are requested first; the table lands in LDS after
    // the first LDS-DMA stages have been issued and the epilogue operands wait in registers: their round trips run under the main loop instead
    // of in front of / behind it (they are older than or interleaved with the first stages: the loop's vmcnt waits still cover every stage)
    constexpr int TABQ = 4;                                    // N <= 1024 (checked by the plan)
    float tcf[TABQ][4];
#pragma unroll
    for (int i = 0; i < TABQ; ++i) {
        const int n = tid + 256 * i;
        const int nc = n < N ? n : 0;
        const float a = coef[FD_CF_A * N + nc], b = coef[FD_CF_C1 * N + nc], c = coef[FD_CF_MU * N + nc], d = coef[FD_CF_C2 * N + nc];
        tcf[i][0] = n < N ? a : 0.0f; tcf[i][1] = n < N ? b : 0.0f; tcf[i][2] = n < N ? c : 0.0f; tcf[i][3] = n < N ? d : 0.0f;
    }
    const int col = k0 + wk * 32 + (lane & 31);
    const long rbase = m0 + wm * 32 + 4 * (lane >> 5);
    float e_sc, e_sh, e_mu, e_is, e_z[16], e_sg[16];
    {
        const int cq = col < K ? col : 0;
        e_sc = st_in[FD_ST_SCALE * K + cq]; e_sh = st_in[FD_ST_SHIFT * K + cq];
        e_mu = st_in[FD_ST_MEAN * K + cq]; e_is = st_in[FD_ST_INVSTD * K + cq];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            long row = rbase + (r & 3) + 8 * (r >> 2);
            if (row > M - 1) row = M - 1;
            e_z[r] = Zin[row * K + cq];
            if (ADD_SG) e_sg[r] = SG[row * K + cq];
        }
    }
    // LDS-DMA sources.  G/Z tiles: 64 rows x 8 chunks = 8 row-groups each -> waves 0..3 take 2 groups of G and 2 of Z.
    // W tile: 32 rows (n) x 16 chunks (k): 64 lanes = 4 rows x 16 chunks -> 8 groups, 2 per wave.
    const float *src_g[2], *src_z[2], *src_w[2];
    int chunk_gz[2], nrow_w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave + 4 * i) * 8 + (lane >> 3);
        chunk_gz[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 4;
        long row = m0 + r; if (row > M - 1) row = M - 1;
        src_g[i] = G + row * N; src_z[i] = Z + row * N;
        nrow_w[i] = (wave + 4 * i) * 4 + (lane >> 4);          // n row within the tile
        int kc = k0 + (lane & 15) * 4; if (kc > K - 4) kc = K - 4;
        src_w[i] = Wt + kc;
    }
    auto issue = [&](int t) {
        float *dst = smem + (t % FD_BWD_STAGES) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int n = t * BR + chunk_gz[i]; if (n >= N) n = 0;
            fd_glds16(src_g[i] + n, dst + (wave + 4 * i) * 8 * BR);
            fd_glds16(src_z[i] + n, dst + BM * BR + (wave + 4 * i) * 8 * BR);
            int nr = t * BR + nrow_w[i]; if (nr > N - 1) nr = N - 1;
            fd_glds16(src_w[i] + (long)nr * K, dst + 2 * BM * BR + (wave + 4 * i) * 4 * BKO);
        }
    };
    fd_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int h = lane >> 5;
    int a_off[4];
    {
        const int ra = wm * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) a_off[g] = ra * BR + (((2 * g + h) ^ ((ra >> 1) & 7)) << 2);
    }
    const int b_col = wk * 32 + (lane & 31);
    const int T = N32 / BR;
    issue(0);
    if (FD_BWD_STAGES > 2 && T > 1) issue(1);
#pragma unroll
    for (int i = 0; i < TABQ; ++i) {
        const int n = tid + 256 * i;
        if (n < N32) { tab[n] = tcf[i][0]; tab[N32 + n] = tcf[i][1]; tab[2 * N32 + n] = tcf[i][2]; tab[3 * N32 + n] = tcf[i][3]; }
    }
    fd_block_barrier_lds();                                    // coefficient table visible
    for (int t = 0; t < T; ++t) {
        if (FD_BWD_STAGES > 2 && t + 1 < T) fd_wait_vmcnt<6>(); else fd_wait_vmcnt<0>();
        fd_block_barrier();
        if (t + FD_BWD_STAGES - 1 < T) issue(t + FD_BWD_STAGES - 1);
        const float *cur = smem + (t % FD_BWD_STAGES) * STAGE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nb = (2 * g + h) * 4;                     // this lane's 4 reduction indices within the tile
            const fd_f32x4 a = fd_dz4(fd_ld4(cur + a_off[g]), fd_ld4(cur + BM * BR + a_off[g]), fd_ld4(tab + t * BR + nb), fd_ld4(tab + N32 + t * BR + nb),
                                      fd_ld4(tab + 2 * N32 + t * BR + nb), fd_ld4(tab + 3 * N32 + t * BR + nb));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float b = cur[2 * BM * BR + (nb + q) * BKO + b_col];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b, acc, 0, 0, 0);
            }
        }
    }
    // epilogue
    float s = 0.0f, q = 0.0f;
    if (col < K) {
        const float sc = e_sc, sh = e_sh, mu = e_mu, is = e_is;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = rbase + (r & 3) + 8 * (r >> 2);
            if (row < M) {
                const float z = e_z[r];
                float v = acc[r];
                if (ADD_SG) v += e_sg[r];
                v *= fd_actmask<ACT_IN>(z * sc + sh);
                Gin[row * K + col] = v;
                s += v; q = fmaf(v, (z - mu) * is, q);
            }
        }
    }
    s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
    if (lane < 32) { red[(wm * 2 + 0) * 64 + wk * 32 + lane] = s; red[(wm * 2 + 1) * 64 + wk * 32 + lane] = q; }
    __syncthreads();
    if (tid < 64 && k0 + tid < K) {
        fd_stat_add<FD_STAT_BWD>(sr, mt, K, 0, k0 + tid, red[0 * 64 + tid] + red[2 * 64 + tid]);
        fd_stat_add<FD_STAT_BWD>(sr, mt, K, 1, k0 + tid, red[1 * 64 + tid] + red[3 * 64 + tid]);
    }
}
template <int ACT_IN, int ADD_SG>
__global__ void __launch_bounds__(256)
fd_pw_dgrad_f32(const float *__restrict__ G, const float *__restrict__ Z, const float *__restrict__ coef,
                const float *__restrict__ Wt, const float *__restrict__ Zin, const float *__restrict__ st_in,
                const float *__restrict__ SG, float *__restrict__ Gin, fd_stat_rows sr,
                int M, int N, int K, int m_tiles, int k_tiles)
{
    fd_pw_dgrad_f32_body<ACT_IN, ADD_SG>(G, Z, coef, Wt, Zin, st_in, SG, Gin, sr, M, N, K, m_tiles, k_tiles, blockIdx.x);
}


// ------------------------------------------------------------------------------------------------
// Pointwise backward-weights GEMM:  dW[N][K] = sum_m dz[m][n] * a_in[m][k],  a_in = act_in(z_in*s+t).
// Output tile 64 (n) x 64 (k) per workgroup; the reduction over the M pixels is split over `splits` workgroups
// (blockIdx.y), each writing a private partial tile (summed afterwards by fd_reduce_partials_f32).
// Both operands have the reduction index m as their ROW index, i.e. the MFMA fragments are 4-byte reads with
// consecutive lanes on consecutive n (resp. k): conflict-free without any transposition.  The per-lane n / k is
// fixed, so the BN-backward coefficients and the producer's scale/shift live in registers.
// LDS per stage: G [32 m][64 n], Z [32 m][64 n], Zin [32 m][64 k]  (256-byte rows, LDS-DMA, 3-stage ring).
// ------------------------------------------------------------------------------------------------
template <int ACT_IN>
__device__ __forceinline__ void              // (bx, by) = (output tile, pixel split): blockIdx of the plain kernel
fd_pw_wgrad_f32_body(const float *__restrict__ G, const float *__restrict__ Z, const float *__restrict__ coef,
                const float *__restrict__ Zin, const float *__restrict__ st_in, float *__restrict__ wpart,
                int M, int N, int K, int k_tiles, int rows_per_split, const int bx, const int by)
{
    constexpr int BR = 32, BT = 64, STAGE = 3 * BR * BT;
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int nt = bx / k_tiles, kt = bx - nt * k_tiles;
    const int n0 = nt * BT, k0 = kt * BT;
    const long mbeg = (long)by * rows_per_split;
    long mend = mbeg + rows_per_split; if (mend > M) mend = M;
    const int T = (int)((mend - mbeg + BR - 1) / BR);
    // LDS-DMA: each tile is 32 rows x 16 chunks; a wave instruction covers 4 rows -> 8 groups per tile, 2 per wave
    int colg[2], colk[2], rowi[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rowi[i] = (wave + 4 * i) * 4 + (lane >> 4);
        int cn = n0 + (lane & 15) * 4; if (cn > N - 4) cn = N - 4;
        int ck = k0 + (lane & 15) * 4; if (ck > K - 4) ck = K - 4;
        colg[i] = cn; colk[i] = ck;
    }
    auto issue = [&](int t) {
        float *dst = smem + (t % FD_BWD_STAGES) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            long m = mbeg + (long)t * BR + rowi[i]; if (m > M - 1) m = M - 1;
            fd_glds16(G + m * N + colg[i], dst + (wave + 4 * i) * 4 * BT);
            fd_glds16(Z + m * N + colg[i], dst + BR * BT + (wave + 4 * i) * 4 * BT);
            fd_glds16(Zin + m * K + colk[i], dst + 2 * BR * BT + (wave + 4 * i) * 4 * BT);
        }
    };
    // NOTE: column clamping (cn > N-4) duplicates valid columns into the tile tail; those tail columns belong to
    // n >= N (or k >= K) and are never stored.
    const int n = n0 + wn * 32 + (lane & 31), k = k0 + wk * 32 + (lane & 31);
    const bool n_ok = n < N, k_ok = k < K;
    const float cA = n_ok ? coef[FD_CF_A * N + n] : 0.0f, c1 = n_ok ? coef[FD_CF_C1 * N + n] : 0.0f, cM = n_ok ? coef[FD_CF_MU * N + n] : 0.0f, c2 = n_ok ? coef[FD_CF_C2 * N + n] : 0.0f;
    const float sc = k_ok ? st_in[FD_ST_SCALE * K + k] : 0.0f, sh = k_ok ? st_in[FD_ST_SHIFT * K + k] : 0.0f;
    fd_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int hh = lane >> 5;
    const int acol = wn * 32 + (lane & 31), bcol = wk * 32 + (lane & 31);
    if (T > 0) issue(0);
    if (FD_BWD_STAGES > 2 && T > 1) issue(1);
    for (int t = 0; t < T; ++t) {
        if (FD_BWD_STAGES > 2 && t + 1 < T) fd_wait_vmcnt<6>(); else fd_wait_vmcnt<0>();
        fd_block_barrier();
        if (t + FD_BWD_STAGES - 1 < T) issue(t + FD_BWD_STAGES - 1);
        const float *cur = smem + (t % FD_BWD_STAGES) * STAGE;
        const long mrow = mbeg + (long)t * BR;
#pragma unroll
        for (int j = 0; j < BR / 2; ++j) {
            const int r = 2 * j + hh;                           // reduction row of this half-wave for MFMA step j
            const bool ok = mrow + r < mend;                    // rows beyond the split contribute nothing
            const float g = cur[r * BT + acol], z = cur[BR * BT + r * BT + acol], zi = cur[2 * BR * BT + r * BT + bcol];
            const float a = ok ? fd_dz(g, z, cA, c1, cM, c2) : 0.0f;
            const float b = fd_act<ACT_IN>(fmaf(zi, sc, sh));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // partial tile: wpart[split][n][k]
    float *o = wpart + (long)by * N * K;
    const int col = k0 + wk * 32 + (lane & 31);
    const int rb = n0 + wn * 32 + 4 * (lane >> 5);
    if (col < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb + (r & 3) + 8 * (r >> 2);
            if (row < N) o[(long)row * K + col] = acc[r];
        }
    }
}
template <int ACT_IN>
__global__ void __launch_bounds__(256)
fd_pw_wgrad_f32(const float *__restrict__ G, const float *__restrict__ Z, const float *__restrict__ coef,
                const float *__restrict__ Zin, const float *__restrict__ st_in, float *__restrict__ wpart,
                int M, int N, int K, int k_tiles, int rows_per_split)
{
    fd_pw_wgrad_f32_body<ACT_IN>(G, Z, coef, Zin, st_in, wpart, M, N, K, k_tiles, rows_per_split, blockIdx.x, blockIdx.y);
}
// Both backward GEMMs of a pointwise unit in one launch (see fd_dw_bwd below for why): the backward-data workgroups first, then the
// tiles x splits weight-gradient workgroups.
template <int ACT_IN, int ADD_SG>
__global__ void __launch_bounds__(256)
fd_pw_bwd_f32(const float *__restrict__ G, const float *__restrict__ Z, const float *__restrict__ coef,
              const float *__restrict__ Wt, const float *__restrict__ Zin, const float *__restrict__ st_in,
              const float *__restrict__ SG, float *__restrict__ Gin, fd_stat_rows sr, float *__restrict__ wpart,
              int M, int N, int K, int m_tiles, int k_tiles, int n_dgrad, int tiles_w, int rows_per_split, int n_w)
{
    // (n_w > 0: the longer-lived weight-gradient workgroups take the first n_w numbers -- see fd_pw_bwd_h16, fd_kernels_train_h16.h)
    const int first_d = n_w > 0 ? n_w : 0;
    if ((int)blockIdx.x >= first_d && (int)blockIdx.x < first_d + n_dgrad) {
        fd_pw_dgrad_f32_body<ACT_IN, ADD_SG>(G, Z, coef, Wt, Zin, st_in, SG, Gin, sr, M, N, K, m_tiles, k_tiles, blockIdx.x - first_d);
    } else {
        const int b = n_w > 0 ? (int)blockIdx.x : (int)blockIdx.x - n_dgrad;
        const int by = b / tiles_w;
        fd_pw_wgrad_f32_body<ACT_IN>(G, Z, coef, Zin, st_in, wpart, M, N, K, k_tiles, rows_per_split, b - by * tiles_w, by);
    }
}


// ------------------------------------------------------------------------------------------------
// Depthwise backward-data.  din[y][x][c] = sum_{ky,kx} dz[(y+P-ky)/S][(x+P-kx)/S][c] * w[c][ky][kx]  (terms with a
// non-integral or out-of-range output index vanish).  A workgroup owns TH x TW INPUT positions x CB channels, stages
// the dz patch it needs (formed from G, z, coef on load) in LDS, then:
//   MODE 0: G_in = mask_in(y_in) * (din (+ skipgrad))                              (input was a_in)
//   MODE 1: G_in(low res) = mask_in(y_in) * sum_{2x2} din                          (input was up2(a_in))
//   MODE 2: as MODE 1, and skipgrad_out = din at full resolution                   (input was up2(a_in) + a_skip)
// plus the producer's BN partial sums, added to ITS statistics rows (fd_stat_add).
// ------------------------------------------------------------------------------------------------
// (body: bm = logical (tile, channel block, image) of this workgroup, grid_x = tiles per image -- the plain kernel passes fd_xcd_image_map() /
// gridDim.x, the paired launch fd_dw_bwd its own numbering)
template <typename T, int K, int S, int MODE, int ACT_IN, int ADD_SG, int N, bool FIN = false>     // FIN: instance with the in-kernel BatchNorm-backward finalisation (fd_bn_bwd_fin)
__device__ __forceinline__ void
fd_dw_dgrad_body(const T *__restrict__ G, const T *__restrict__ Z, const float *__restrict__ coef,
                const float *__restrict__ w, const T *__restrict__ Zin, const float *__restrict__ st_in,
                const T *__restrict__ SG, T *__restrict__ Gin, T *__restrict__ SGout, fd_stat_rows sr,
                int Hin, int Win, int Ho, int Wo, int C, int cbq, int TH, int TW, int tiles_x, int csplit, int pstr, const fd_blk3 bm, const int grid_x,
                const fd_bn_bwd_fin &fin)
{
    constexpr int P = K / 2;
    constexpr int UNR_TAPROWS = K == 3 ? 3 : 1;
    typedef fd_lane<T, N> LN;
    typedef typename LN::vec vec;
    typedef typename LN::lds_t lds_t;
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int lanes_c = 1 << cbq, CB = lanes_c * N, PSTR = pstr;
    // output (dz) positions that can touch input rows [iy0, iy0+TH): oy in [floor((iy0+P-(K-1))/S) .. floor((iy0+TH-1+P)/S)]
    // (bm: all tiles / channel blocks of an image on one XCD: halo re-reads hit its L2)
    const int ty = bm.x / tiles_x, tx = bm.x - ty * tiles_x;
    const int c0 = bm.y * CB, n = bm.z;
    const int iy0 = ty * TH, ix0 = tx * TW;
    const int oyb = (iy0 + P - (K - 1) >= 0) ? (iy0 + P - (K - 1)) / S : -((-(iy0 + P - (K - 1)) + S - 1) / S);
    const int oxb = (ix0 + P - (K - 1) >= 0) ? (ix0 + P - (K - 1)) / S : -((-(ix0 + P - (K - 1)) + S - 1) / S);
    const int PH = (iy0 + TH - 1 + P) / S - oyb + 1, PW = (ix0 + TW - 1 + P) / S - oxb + 1;
    lds_t *s_dz = reinterpret_cast<lds_t *>(smem_raw);    // [PH*PW][PSTR]  (N = 8: dz rounded to the storage type, as fd_bn_bwd_apply_h16 does for the GEMMs)
    float *s_w = reinterpret_cast<float *>(smem_raw + fd_lds_patch_bytes<lds_t>(PH * PW, PSTR));   // [K*K][CB]
    const int tid = threadIdx.x, c4 = tid & (lanes_c - 1), pt = tid >> cbq, npt = 256 >> cbq;
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
    vec cA = LN::zero(), c1 = LN::zero(), cM = LN::zero(), c2 = LN::zero();
    const bool fin_here = FIN && fin.rows != nullptr;               // this unit's BatchNorm backward is finalised here, after the first batch of patch loads has been issued
    if (c_ok && !fin_here) { cA = LN::ldf(coef + FD_CF_A * C + cg); c1 = LN::ldf(coef + FD_CF_C1 * C + cg); cM = LN::ldf(coef + FD_CF_MU * C + cg); c2 = LN::ldf(coef + FD_CF_C2 * C + cg); }
    const int npx = PH * PW;
    constexpr int U = K == 3 ? FD_DW_U3 : 8;
    fd_px_walk wk(pt, npt, PW);
    for (int base = pt; base < npx || (fin_here && base == pt); base += npt * U) {
        typename LN::raw g[U], z[U];                        // (storage-typed until they are used: half the registers of a 16-bit plan's batch in flight)
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = base + u * npt;
            const int py = wk.iy, pxx = wk.ix;
            wk.next();
            const int oy = oyb + py, ox = oxb + pxx;
            ok[u] = px < npx && c_ok && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
            {   // branch-free staging (clamped addresses, unconditional loads): see fd_dwconv_train
                const int qy = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy), qx = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
                const long o = fd_nhwc(n, Ho, qy, Wo, qx, C, (c_ok ? cg : 0));
                g[u] = LN::ldraw(G + o); z[u] = LN::ldraw(Z + o);
            }
        }
        if (fin_here && base == pt) {
            float *s_cf = reinterpret_cast<float *>(smem_raw + fin.cf_off);
            fd_bstat_table_block(fin, reinterpret_cast<double *>(smem_raw), s_cf, c0, CB, C, tid, bm.x == 0 && bm.z == 0);
            if (c_ok) { cA = LN::ldf(s_cf + FD_CF_A * CB + c4 * N); c1 = LN::ldf(s_cf + FD_CF_C1 * CB + c4 * N); cM = LN::ldf(s_cf + FD_CF_MU * CB + c4 * N); c2 = LN::ldf(s_cf + FD_CF_C2 * CB + c4 * N); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = base + u * npt;
            if (px < npx) LN::lds_st(s_dz + px * PSTR + c4 * N, ok[u] ? fd_dz4(LN::cvt(g[u]), LN::cvt(z[u]), cA, c1, cM, c2) : LN::zero());
        }
    }
#pragma unroll
    for (int j = 0; j < NWREG; ++j) { const int i = tid + 256 * j; if (i < K * K * CB) s_w[i] = wreg[j]; }
    __syncthreads();

    // producer's tables.  MODE 3 (concatenating consumer): channels [0, csplit) belong to the low-resolution producer (pitch
    // csplit: 2x2 sum, mask, BN partials as in MODE 1), the rest to the skip tensor (pitch C - csplit: full-resolution gradient
    // into its skip-gradient buffer, masked later by the skip source's own consumer)
    const bool to_skip = MODE == 3 && cg >= csplit;
    const int Cp = MODE == 3 ? csplit : C, C2 = MODE == 3 ? C - csplit : C, cl = to_skip ? cg - csplit : cg;
    vec sc = LN::zero(), sh = LN::zero(), mu = LN::zero(), is = LN::zero();
    if (c_ok && !to_skip) { sc = LN::ldf(st_in + FD_ST_SCALE * Cp + cg); sh = LN::ldf(st_in + FD_ST_SHIFT * Cp + cg); mu = LN::ldf(st_in + FD_ST_MEAN * Cp + cg); is = LN::ldf(st_in + FD_ST_INVSTD * Cp + cg); }
    vec ssum = LN::zero(), ssx = LN::zero();
    auto din_at = [&](int iy, int ix) {                    // gradient w.r.t. the conv input at tile-local (iy, ix)
        vec acc = LN::zero();
        const int gy = iy0 + iy, gx = ix0 + ix;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int ny = gy + P - ky;
            if (S == 2 && (ny & 1)) continue;
            const int py = (S == 2 ? (ny >> 1) : ny) - oyb;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int nx = gx + P - kx;
                if (S == 2 && (nx & 1)) continue;
                const int pxx = (S == 2 ? (nx >> 1) : nx) - oxb;
                acc += LN::lds_ld(s_dz + (py * PW + pxx) * PSTR + c4 * N) * LN::ldf(s_w + (ky * K + kx) * CB + c4 * N);
            }
        }
        return acc;
    };
    if (MODE == 0 && S == 1) {
        // stride 1: din = correlation of the dz patch with the FLIPPED taps -- same strip scheme as the forward kernel: 4 adjacent
        // positions share K + 3 patch vectors per tap row (13 LDS reads per 20 FMAs instead of 40)
        const int TWS = TW >> 2;
        for (int st = pt; st < TH * TWS; st += npt) {
            const int iy = st / TWS, ix = (st - iy * TWS) * 4;
            const int gy = iy0 + iy;
            if (!c_ok || gy >= Hin) continue;
            vec z[4], sgv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                      // requested before the tap loop; clamped column, validity re-checked at the store
                const int gx = ix0 + ix + j, qx = gx < Win ? gx : Win - 1;
                const long o = fd_nhwc(n, Hin, gy, Win, qx, C, cg);
                z[j] = LN::ld(Zin + o);
                sgv[j] = ADD_SG ? LN::ld(SG + o) : LN::zero();
            }
            vec acc[4] = {LN::zero(), LN::zero(), LN::zero(), LN::zero()};
#pragma unroll UNR_TAPROWS
            for (int a = 0; a < K; ++a) {
                const lds_t *row = s_dz + ((iy + a) * PW + ix) * PSTR + c4 * N;
                vec r[K + 3];
#pragma unroll
                for (int i = 0; i < K + 3; ++i) r[i] = LN::lds_ld(row + i * PSTR);
#pragma unroll
                for (int b = 0; b < K; ++b) {
                    const vec wv = LN::ldf(s_w + ((K - 1 - a) * K + (K - 1 - b)) * CB + c4 * N);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += r[j + b] * wv;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gx = ix0 + ix + j;
                if (gx >= Win) continue;
                const long o = fd_nhwc(n, Hin, gy, Win, gx, C, cg);
                vec v = acc[j];
                if (ADD_SG) v += sgv[j];
                v = LN::round(v * fd_actmask4<ACT_IN>(z[j] * sc + sh));
                LN::st(Gin + o, v);
                ssum += v; ssx += v * ((z[j] - mu) * is);
            }
        }
    } else if (MODE == 0) {
        for (int p = pt; p < TH * TW; p += npt) {
            const int iy = p / TW, ix = p - iy * TW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            if (!c_ok || gy >= Hin || gx >= Win) continue;
            const long o = fd_nhwc(n, Hin, gy, Win, gx, C, cg);
            const vec z = LN::ld(Zin + o);                 // requested before the tap loop: the latency hides behind the LDS work
            vec sgv = LN::zero();
            if (ADD_SG) sgv = LN::ld(SG + o);
            vec v = din_at(iy, ix);
            if (ADD_SG) v += sgv;
            v = LN::round(v * fd_actmask4<ACT_IN>(z * sc + sh));
            LN::st(Gin + o, v);
            ssum += v; ssx += v * ((z - mu) * is);
        }
    } else {
        const int TH2 = TH >> 1, TW2 = TW >> 1, Hs = Hin >> 1, Ws = Win >> 1;
        for (int p = pt; p < TH2 * TW2; p += npt) {
            const int ly = p / TW2, lx = p - ly * TW2;
            const int gy = iy0 + 2 * ly, gx = ix0 + 2 * lx;     // top-left full-res position of the 2x2 block
            if (!c_ok || gy >= Hin || gx >= Win) continue;
            const long ol = fd_nhwc(n, Hs, (gy >> 1), Ws, (gx >> 1), Cp, (to_skip ? 0 : cg));
            const vec z = LN::ld(Zin + ol);                // requested before the tap loop (unused by the lanes that feed the skip tensor)
            vec d00, d01, d10, d11;
            if (S == 1) {
                // the four positions of the block read a (K+1) x (K+1) window of the dz patch: walk it row by row, every row feeds
                // the block's upper row with tap row ky = K-1-r and its lower row with ky = K-r (one pass over LDS, few registers)
                d00 = LN::zero(); d01 = LN::zero(); d10 = LN::zero(); d11 = LN::zero();
                const int pyb = gy + P - (K - 1) - oyb, pxb = gx + P - (K - 1) - oxb;
#pragma unroll 1
                for (int r = 0; r <= K; ++r) {
                    vec v[K + 1];
                    const lds_t *row = s_dz + ((pyb + r) * PW + pxb) * PSTR + c4 * N;
#pragma unroll
                    for (int c = 0; c <= K; ++c) v[c] = LN::lds_ld(row + c * PSTR);
                    if (r < K) {
                        const float *wr = s_w + (K - 1 - r) * K * CB + c4 * N;
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) { const vec wv = LN::ldf(wr + kx * CB); d00 += v[K - 1 - kx] * wv; d01 += v[K - kx] * wv; }
                    }
                    if (r > 0) {
                        const float *wr = s_w + (K - r) * K * CB + c4 * N;
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) { const vec wv = LN::ldf(wr + kx * CB); d10 += v[K - 1 - kx] * wv; d11 += v[K - kx] * wv; }
                    }
                }
            } else {
                d00 = din_at(2 * ly, 2 * lx); d01 = din_at(2 * ly, 2 * lx + 1); d10 = din_at(2 * ly + 1, 2 * lx); d11 = din_at(2 * ly + 1, 2 * lx + 1);
            }
            if (MODE == 2 || to_skip) {
                const long o = fd_nhwc(n, Hin, gy, Win, gx, C2, cl);
                LN::st(SGout + o, d00); LN::st(SGout + o + C2, d01);
                LN::st(SGout + o + (long)Win * C2, d10); LN::st(SGout + o + (long)Win * C2 + C2, d11);
                if (to_skip) continue;
            }
            vec v = (d00 + d01) + (d10 + d11);
            v = LN::round(v * fd_actmask4<ACT_IN>(z * sc + sh));
            LN::st(Gin + ol, v);
            ssum += v; ssx += v * ((z - mu) * is);
        }
    }
    float *red = smem;
    fd_wg_stat_add<FD_STAT_BWD>(ssum, ssx, red, lanes_c, tid, sr, (long)bm.z * grid_x + bm.x, Cp, c0);
}
template <typename T, int K, int S, int MODE, int ACT_IN, int ADD_SG, int N>
__global__ void __launch_bounds__(256)
fd_dw_dgrad(const T *__restrict__ G, const T *__restrict__ Z, const float *__restrict__ coef,
                const float *__restrict__ w, const T *__restrict__ Zin, const float *__restrict__ st_in,
                const T *__restrict__ SG, T *__restrict__ Gin, T *__restrict__ SGout, fd_stat_rows sr,
                int Hin, int Win, int Ho, int Wo, int C, int cbq, int TH, int TW, int tiles_x, int csplit, int pstr)
{
    const fd_bn_bwd_fin no_fin{};
    fd_dw_dgrad_body<T, K, S, MODE, ACT_IN, ADD_SG, N>(G, Z, coef, w, Zin, st_in, SG, Gin, SGout, sr, Hin, Win, Ho, Wo, C, cbq, TH, TW, tiles_x, csplit, pstr,
                                                    fd_xcd_image_map(), (int)gridDim.x, no_fin);
}


// ------------------------------------------------------------------------------------------------
// Depthwise backward-weights: dW[c][ky][kx] = sum over (n, oy, ox) of dz[oy][ox][c] * in[oy*S-P+ky][ox*S-P+kx][c].
// Same tiling and input staging as the forward kernel (the input a_in / up2 / +skip is re-created on load); dz of the tile's
// outputs is staged next to it.  A work-item then owns ONE tap row of 4 channels and walks a share of the output strips; the
// shares are summed through LDS once per workgroup, which writes wpart[blk][K*K][C].
// ------------------------------------------------------------------------------------------------
template <typename T, int K, int S, int MODE, int ACT1, int ACT2, int N, bool FIN = false>
__device__ __forceinline__ void
fd_dw_wgrad_body(const T *__restrict__ zin, const float *__restrict__ st1, const T *__restrict__ zskip,
                const float *__restrict__ st2, const T *__restrict__ G, const T *__restrict__ Z,
                const float *__restrict__ coef, float *__restrict__ wpart, int Hin, int Win, int Ho, int Wo, int C,
                int cbq, int TH, int TW, int tiles_x, int tpw, int csplit, int pstr, const fd_blk3 bm, const int grid_x, const fd_bn_bwd_fin &fin)
{
    constexpr int P = K / 2;
    constexpr int NIN = 3 * S + K;
    typedef fd_lane<T, N> LN;
    typedef typename LN::vec vec;
    typedef typename LN::lds_t lds_t;
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int lanes_c = 1 << cbq, CB = lanes_c * N, PSTR = pstr;
    const int TH_in = (TH - 1) * S + K, TW_in = (TW - 1) * S + K;
    lds_t *s_in = reinterpret_cast<lds_t *>(smem_raw);     // [TH_in*TW_in][PSTR] activated input patch; reused for the final reduction
    lds_t *s_dz = s_in + TH_in * TW_in * PSTR;             // [TH*TW][PSTR]       dz of the tile's outputs (0 outside the image)
    // a workgroup walks `tpw` horizontally adjacent tiles and keeps its tap sums in registers across them: one workgroup
    // reduction and one partial row per `tpw` tiles
    const int groups_x = (tiles_x + tpw - 1) / tpw;
    const int ty = bm.x / groups_x, tgx = bm.x - ty * groups_x;
    const int c0 = bm.y * CB, n = bm.z;
    const int oy0 = ty * TH;
    const int iy0 = oy0 * S - P;
    const int tid = threadIdx.x, c4 = tid & (lanes_c - 1), pt = tid >> cbq, npt = 256 >> cbq;
    const int cg = c0 + c4 * N;
    const bool c_ok = cg < C;
    int tab_c = c_ok ? cg : 0;                             // channel offset of this work-item's table entries (re-read per tile, see FD_OPAQUE)
    const bool from_skip = MODE == 3 && cg >= csplit;      // MODE 3: cat(up2(a_in), a_skip), see fd_dwconv_train
    const int C1 = MODE == 3 ? csplit : C, C2 = MODE == 3 ? C - csplit : C, cl = from_skip ? cg - csplit : cg;
    // compute-phase mapping: work-item = (channel group c4, tap row ky, pixel group pg).  It owns only the K taps of row ky
    // (K float4 accumulators instead of K*K: the 5x5 kernel needed 216+ VGPRs and a 25 x 3-step shuffle reduction per tile
    // before) and walks the output strips pg, pg + ngroups, ... of every tile; the pixel groups meet once, through LDS, at the end.
    const int ngroups = npt / K;
    const int ky = pt % K, pg = pt / K;
    const bool worker = pg < ngroups;
    vec acc[K];
#pragma unroll
    for (int t = 0; t < K; ++t) acc[t] = LN::zero();
#pragma unroll 1
    for (int ti = 0; ti < tpw; ++ti) {
    const int tx = tgx * tpw + ti;
    if (tx >= tiles_x) break;
    const int ox0 = tx * TW, ix0 = ox0 * S - P;
    if (ti > 0) __syncthreads();                           // the previous tile's patch reads are done
    // the BN-backward operands (G, z) of this work-item's FIRST output strip are requested before the input patch is staged,
    // so that their latency overlaps the staging loads instead of following the barrier
    const int TWS = TW >> 2, nstrips = TH * TWS;
    constexpr bool PREFETCH = true;
    typename LN::raw g0[4], z0[4];
    {
        const int oy = pt / TWS, ox = (pt - oy * TWS) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gy = oy0 + oy, gx = ox0 + ox + j;
            if (PREFETCH) {                                   // branch-free: clamped address, validity is re-checked where the value is used
                const int qy = gy < Ho ? gy : Ho - 1, qx = gx < Wo ? gx : Wo - 1;
                const long o = fd_nhwc(n, Ho, qy, Wo, qx, C, (c_ok ? cg : 0));
                g0[j] = LN::ldraw(G + o); z0[j] = LN::ldraw(Z + o);
            }
        }
    }
    FD_OPAQUE(tab_c);
    int tab_l = c_ok ? cl : 0;
    FD_OPAQUE(tab_l);
    vec s1, t1, s2 = LN::zero(), t2 = LN::zero();
    if (from_skip) { s1 = LN::ldf(st2 + FD_ST_SCALE * C2 + tab_l); t1 = LN::ldf(st2 + FD_ST_SHIFT * C2 + tab_l); }
    else { s1 = LN::ldf(st1 + FD_ST_SCALE * C1 + tab_l); t1 = LN::ldf(st1 + FD_ST_SHIFT * C1 + tab_l); }
    if (MODE == 2) { s2 = LN::ldf(st2 + FD_ST_SCALE * C + tab_c); t2 = LN::ldf(st2 + FD_ST_SHIFT * C + tab_c); }
    const int npx_in = TH_in * TW_in;
    constexpr int U = sizeof(T) == 2 ? (K == 3 ? FD_DW_WU3 : FD_DW_WU5) : 4;
    const bool fin_now = FIN && fin.rows != nullptr && ti == 0;   // the unit's BatchNorm backward finalised here (every role of a paired launch computes the same bits; role 0 writes them)
    fd_px_walk wk(pt, npt, TW_in);
    for (int base = pt; base < npx_in || (fin_now && base == pt); base += npt * U) {
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
            // branch-free staging (clamped addresses, unconditional loads): see fd_dwconv_train
            const int qy = gy < 0 ? 0 : (gy >= Hin ? Hin - 1 : gy), qx = gx < 0 ? 0 : (gx >= Win ? Win - 1 : gx);
            const int ql = c_ok ? cl : 0, qg = c_ok ? cg : 0;
            if (MODE == 0) {
                v[u] = LN::ld(zin + fd_nhwc(n, Hin, qy, Win, qx, C, qg));
            } else {
                const int Hs = Hin >> 1, Ws = Win >> 1;
                if (from_skip) v[u] = LN::ld(zskip + fd_nhwc(n, Hin, qy, Win, qx, C2, ql));
                else v[u] = LN::ld(zin + fd_nhwc(n, Hs, (qy >> 1), Ws, (qx >> 1), C1, ql));
                if (MODE == 2) sk[u] = LN::ld(zskip + fd_nhwc(n, Hin, qy, Win, qx, C, qg));
            }
        }
        if (fin_now && base == pt)
            fd_bstat_table_block(fin, reinterpret_cast<double *>(smem_raw), reinterpret_cast<float *>(smem_raw + fin.cf_off), c0, CB, C, tid, false);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = base + u * npt;
            if (px < npx_in) {
                vec a = LN::zero();
                if (ok[u]) {
                    a = from_skip ? fd_bn_act4<ACT2>(v[u], s1, t1) : fd_bn_act4<ACT1>(v[u], s1, t1);
                    if (MODE == 2) a += fd_bn_act4<ACT2>(sk[u], s2, t2);
                }
                LN::lds_st(s_in + px * PSTR + c4 * N, a);
            }
        }
    }
    // dz of this work-item's output strip (pixel-thread pt stages strip pt: TH*TW/4 <= npt) -> s_dz
    FD_OPAQUE(tab_c);
    {
        const bool cf_lds = FIN && fin.rows != nullptr;
        const float *cf = cf_lds ? reinterpret_cast<const float *>(smem_raw + fin.cf_off) + c4 * N : coef + tab_c;     // [4][CB] in LDS / [4][C] in memory
        const int cfp = cf_lds ? CB : C;
        const vec cA = LN::ldf(cf + FD_CF_A * cfp), c1 = LN::ldf(cf + FD_CF_C1 * cfp), cM = LN::ldf(cf + FD_CF_MU * cfp), c2 = LN::ldf(cf + FD_CF_C2 * cfp);
        if (pt < nstrips) {
            const int oy = pt / TWS, ox = (pt - oy * TWS) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = c_ok && oy0 + oy < Ho && ox0 + ox + j < Wo;
                const vec dz = fd_dz4(LN::cvt(g0[j]), LN::cvt(z0[j]), cA, c1, cM, c2);
                LN::lds_st(s_dz + (oy * TW + ox + j) * PSTR + c4 * N, ok ? dz : LN::zero());
            }
        }
    }
    __syncthreads();
    if (worker) {
        for (int s = pg; s < nstrips; s += ngroups) {
            const int oy = s / TWS, ox = (s - oy * TWS) * 4;
            const lds_t *row = s_in + ((oy * S + ky) * TW_in + ox * S) * PSTR + c4 * N;
            const lds_t *dzp = s_dz + (oy * TW + ox) * PSTR + c4 * N;
            vec r[NIN], dz[4];
#pragma unroll
            for (int i = 0; i < NIN; ++i) r[i] = LN::lds_ld(row + i * PSTR);
#pragma unroll
            for (int j = 0; j < 4; ++j) dz[j] = LN::lds_ld(dzp + j * PSTR);
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[kx] += r[j * S + kx] * dz[j];
        }
    }
    }   // tile loop
    // the pixel groups' tap sums meet in LDS: red[pg][ky*K + kx][c4] (fixed order -> deterministic)
    __syncthreads();
    float *red = smem;
    if (worker) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) LN::stf(red + ((pg * K * K + ky * K + kx) * lanes_c + c4) * N, acc[kx]);
    }
    __syncthreads();
    const long blk = (long)bm.z * grid_x + bm.x;
    for (int i = tid; i < K * K * lanes_c; i += 256) {
        const int t = i >> cbq, cc = i & (lanes_c - 1);
        if (c0 + cc * N < C) {
            vec a = LN::zero();
            for (int g = 0; g < ngroups; ++g) a += LN::ldf(red + ((g * K * K + t) * lanes_c + cc) * N);
            LN::stf(wpart + (blk * K * K + t) * C + c0 + cc * N, a);
        }
    }
}
template <typename T, int K, int S, int MODE, int ACT1, int ACT2, int N>
__global__ void __launch_bounds__(256)
fd_dw_wgrad(const T *__restrict__ zin, const float *__restrict__ st1, const T *__restrict__ zskip,
                const float *__restrict__ st2, const T *__restrict__ G, const T *__restrict__ Z,
                const float *__restrict__ coef, float *__restrict__ wpart, int Hin, int Win, int Ho, int Wo, int C,
                int cbq, int TH, int TW, int tiles_x, int tpw, int csplit, int pstr)
{
    const fd_bn_bwd_fin no_fin{};
    fd_dw_wgrad_body<T, K, S, MODE, ACT1, ACT2, N>(zin, st1, zskip, st2, G, Z, coef, wpart, Hin, Win, Ho, Wo, C, cbq, TH, TW, tiles_x, tpw, csplit, pstr,
                                                fd_xcd_image_map(), (int)gridDim.x, no_fin);
}

// ------------------------------------------------------------------------------------------------
// One launch for BOTH backward kernels of a depthwise unit ("horizontal" pairing).  They are independent of each other (both read G, z and
// the unit's saved input; one writes the producer's gradient, the other the weight-gradient partials), and as two launches on one stream
// they serialise: on the 14x14 / 7x7 maps each is a single round of workgroups bound by its own load -> LDS -> taps -> store latency
// (12-13 us for ~2 us of HBM time), and every launch pays its ramp and its tail.  Here the 1-D grid is dealt in groups of eight images:
// first the backward-data workgroups of the group, then its weight-gradient workgroups -- workgroup b still runs on XCD b % 8, so all work
// of an image stays on one XCD and the second role finds G / z in that XCD's L2.  LDS and registers are those of the larger role.
// ------------------------------------------------------------------------------------------------
struct fd_pair_blk { int role; fd_blk3 b; };
__device__ __forceinline__ fd_pair_blk fd_pair_map(unsigned b, unsigned per0, unsigned per1, unsigned nimg)
{
    const unsigned per = per0 + per1;
    const unsigned grp = b / (8u * per);
    unsigned rem = b - grp * 8u * per;
    const unsigned m = nimg - grp * 8u < 8u ? nimg - grp * 8u : 8u;   // images in this group (the last one may be short)
    fd_pair_blk r;
    r.role = rem >= m * per0;
    if (r.role) rem -= m * per0;
    r.b.z = (int)(grp * 8u + rem % m); r.b.x = (int)(rem / m); r.b.y = 0;
    return r;
}
template <typename T> struct fd_dw_bwd_args {
    const T *G, *Z, *Zin, *Zskip, *SG;
    T *Gin, *SGout;
    const float *coef, *w, *st_in, *st_skip;
    fd_stat_rows sr;                                   // statistics rows of the PRODUCER (sum G_in, sum G_in * xhat_in)
    float *wpart;
    int Hin, Win, Ho, Wo, C, cbq, csplit, pstr;        // pstr: LDS patch pitch in floats
    int d_th, d_tw, d_tiles_x, d_gx, d_gy;             // backward-data: INPUT-space tiles; grid (d_gx tiles, d_gy channel blocks) per image
    int w_th, w_tw, w_tiles_x, w_tpw, w_gx, w_gy;      // backward-weights: OUTPUT-space tiles, tpw of them per workgroup
    int B;
    fd_bn_bwd_fin fin;                                 // rows != null: this unit's BatchNorm backward is finalised by these workgroups (fd_bstat_table_block)
};
template <typename T, int K, int S, int MODE, int ACT1, int ACT2, int ADD_SG, int N, bool FIN = false>
__global__ void __launch_bounds__(256)
fd_dw_bwd(const fd_dw_bwd_args<T> a)
{
    const fd_pair_blk pb = fd_pair_map(blockIdx.x, (unsigned)(a.d_gx * a.d_gy), (unsigned)(a.w_gx * a.w_gy), (unsigned)a.B);
    fd_blk3 bm = pb.b;
    if (pb.role == 0) {
        bm.y = bm.x / a.d_gx; bm.x -= bm.y * a.d_gx;
        fd_dw_dgrad_body<T, K, S, MODE, ACT1, ADD_SG, N, FIN>(a.G, a.Z, a.coef, a.w, a.Zin, a.st_in, a.SG, a.Gin, a.SGout, a.sr, a.Hin, a.Win, a.Ho, a.Wo, a.C, a.cbq,
                                                      a.d_th, a.d_tw, a.d_tiles_x, a.csplit, a.pstr, bm, a.d_gx, a.fin);
    } else {
        bm.y = bm.x / a.w_gx; bm.x -= bm.y * a.w_gx;
        fd_dw_wgrad_body<T, K, S, MODE, ACT1, ACT2, N, FIN>(a.Zin, a.st_in, a.Zskip, a.st_skip, a.G, a.Z, a.coef, a.wpart, a.Hin, a.Win, a.Ho, a.Wo, a.C, a.cbq,
                                                    a.w_th, a.w_tw, a.w_tiles_x, a.w_tpw, a.csplit, a.pstr, bm, a.w_gx, a.fin);
    }
}

template <typename T, int K, int S, int MODE, int ACT_IN, int ACT2, int ADD_SG, int N, bool FIN = false>
__device__ __forceinline__ void
fd_dw_bwd1_body(const T *__restrict__ G, const T *__restrict__ Z, const float *__restrict__ coef,
                const float *__restrict__ w, const T *__restrict__ Zin, const float *__restrict__ st_in,
                const T *__restrict__ Zskip, const float *__restrict__ st_skip,
                const T *__restrict__ SG, T *__restrict__ Gin, T *__restrict__ SGout, fd_stat_rows sr, float *__restrict__ wpart,
                int Hin, int Win, int Ho, int Wo, int C, int cbq, int TH, int TW, int tiles_x, int csplit, int pstr, const fd_blk3 bm, const int grid_x,
                const fd_bn_bwd_fin &fin)
{
    constexpr int P = K / 2;
    constexpr int UNR_TAPROWS = K == 3 ? 3 : 1;
    typedef fd_lane<T, N> LN;
    typedef typename LN::vec vec;
    typedef typename LN::lds_t lds_t;
    FD_DYN_SMEM(smem_raw);
    float *smem = reinterpret_cast<float *>(smem_raw);
    const int lanes_c = 1 << cbq, CB = lanes_c * N, PSTR = pstr;
    // output (dz) positions that can touch input rows [iy0, iy0+TH): oy in [floor((iy0+P-(K-1))/S) .. floor((iy0+TH-1+P)/S)]
    // (bm: all tiles / channel blocks of an image on one XCD: halo re-reads hit its L2)
    const int ty = bm.x / tiles_x, tx = bm.x - ty * tiles_x;
    const int c0 = bm.y * CB, n = bm.z;
    const int iy0 = ty * TH, ix0 = tx * TW;
    const int oyb = (iy0 + P - (K - 1) >= 0) ? (iy0 + P - (K - 1)) / S : -((-(iy0 + P - (K - 1)) + S - 1) / S);
    const int oxb = (ix0 + P - (K - 1) >= 0) ? (ix0 + P - (K - 1)) / S : -((-(ix0 + P - (K - 1)) + S - 1) / S);
    const int PH = (iy0 + TH - 1 + P) / S - oyb + 1, PW = (ix0 + TW - 1 + P) / S - oxb + 1;
    // the positions the OWNED outputs (those whose receptive field starts in this tile: rows [iy0 / S, (iy0 + TH) / S)) read of the unit's forward input
    const int OTH = TH / S, OTW = TW / S, oy0 = iy0 / S, ox0 = ix0 / S;
    const int TH_in = (OTH - 1) * S + K, TW_in = (OTW - 1) * S + K;
    const int jy0 = oy0 * S - P, jx0 = ox0 * S - P;       // input position of patch pixel (0, 0)
    lds_t *s_dz = reinterpret_cast<lds_t *>(smem_raw);    // [PH*PW][PSTR]        dz of every output that touches the tile (0 outside the image)
    lds_t *s_in = s_dz + PH * PW * PSTR;                  // [TH_in*TW_in][PSTR]  the activated forward input under the owned outputs (0 = padding)
    float *s_w = reinterpret_cast<float *>(smem_raw + fd_lds_patch_bytes<lds_t>(PH * PW + TH_in * TW_in, PSTR));   // [K*K][CB]
    const int tid = threadIdx.x, c4 = tid & (lanes_c - 1), pt = tid >> cbq, npt = 256 >> cbq;
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
    vec cA = LN::zero(), c1 = LN::zero(), cM = LN::zero(), c2 = LN::zero();
    const bool fin_here = FIN && fin.rows != nullptr;               // (see fd_dw_dgrad_body)
    if (c_ok && !fin_here) { cA = LN::ldf(coef + FD_CF_A * C + cg); c1 = LN::ldf(coef + FD_CF_C1 * C + cg); cM = LN::ldf(coef + FD_CF_MU * C + cg); c2 = LN::ldf(coef + FD_CF_C2 * C + cg); }
    const int npx = PH * PW;
    constexpr int U = 8;
    fd_px_walk wk(pt, npt, PW);
    for (int base = pt; base < npx || (fin_here && base == pt); base += npt * U) {
        typename LN::raw g[U], z[U];                        // (storage-typed until they are used: half the registers of a 16-bit plan's batch in flight)
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = base + u * npt;
            const int py = wk.iy, pxx = wk.ix;
            wk.next();
            const int oy = oyb + py, ox = oxb + pxx;
            ok[u] = px < npx && c_ok && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
            {   // branch-free staging (clamped addresses, unconditional loads): see fd_dwconv_train
                const int qy = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy), qx = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
                const long o = fd_nhwc(n, Ho, qy, Wo, qx, C, (c_ok ? cg : 0));
                g[u] = LN::ldraw(G + o); z[u] = LN::ldraw(Z + o);
            }
        }
        if (fin_here && base == pt) {
            float *s_cf = reinterpret_cast<float *>(smem_raw + fin.cf_off);
            fd_bstat_table_block(fin, reinterpret_cast<double *>(smem_raw), s_cf, c0, CB, C, tid, bm.x == 0 && bm.z == 0);
            if (c_ok) { cA = LN::ldf(s_cf + FD_CF_A * CB + c4 * N); c1 = LN::ldf(s_cf + FD_CF_C1 * CB + c4 * N); cM = LN::ldf(s_cf + FD_CF_MU * CB + c4 * N); c2 = LN::ldf(s_cf + FD_CF_C2 * CB + c4 * N); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int px = base + u * npt;
            if (px < npx) LN::lds_st(s_dz + px * PSTR + c4 * N, ok[u] ? fd_dz4(LN::cvt(g[u]), LN::cvt(z[u]), cA, c1, cM, c2) : LN::zero());
        }
    }
    {   // ---- the forward input patch, re-created on load as in fd_dwconv_train: act(z_in * s + t), nearest x2, + / cat skip ----
        const bool from_skip = MODE == 3 && cg >= csplit;
        const int C1w = MODE == 3 ? csplit : C, C2w = MODE == 3 ? C - csplit : C, clw = from_skip ? cg - csplit : cg;
        vec s1 = LN::zero(), t1 = LN::zero(), s2 = LN::zero(), t2 = LN::zero();
        if (c_ok) {
            if (from_skip) { s1 = LN::ldf(st_skip + FD_ST_SCALE * C2w + clw); t1 = LN::ldf(st_skip + FD_ST_SHIFT * C2w + clw); }
            else { s1 = LN::ldf(st_in + FD_ST_SCALE * C1w + clw); t1 = LN::ldf(st_in + FD_ST_SHIFT * C1w + clw); }
            if (MODE == 2) { s2 = LN::ldf(st_skip + FD_ST_SCALE * C + cg); t2 = LN::ldf(st_skip + FD_ST_SHIFT * C + cg); }
        }
        const int npx_in = TH_in * TW_in;
        constexpr int UI = 4;
        fd_px_walk wi(pt, npt, TW_in);
        for (int base = pt; base < npx_in; base += npt * UI) {
            vec v[UI], sk[UI];
            bool oki[UI];
#pragma unroll
            for (int u = 0; u < UI; ++u) {
                const int px = base + u * npt;
                const int iy = wi.iy, ix = wi.ix;
                wi.next();
                const int gy = jy0 + iy, gx = jx0 + ix;
                sk[u] = LN::zero();
                oki[u] = px < npx_in && c_ok && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
                const int qy = gy < 0 ? 0 : (gy >= Hin ? Hin - 1 : gy), qx = gx < 0 ? 0 : (gx >= Win ? Win - 1 : gx);
                const int ql = c_ok ? clw : 0, qg = c_ok ? cg : 0;
                if (MODE == 0) {
                    v[u] = LN::ld(Zin + fd_nhwc(n, Hin, qy, Win, qx, C, qg));
                } else {
                    const int Hs = Hin >> 1, Ws = Win >> 1;
                    if (from_skip) v[u] = LN::ld(Zskip + fd_nhwc(n, Hin, qy, Win, qx, C2w, ql));
                    else v[u] = LN::ld(Zin + fd_nhwc(n, Hs, (qy >> 1), Ws, (qx >> 1), C1w, ql));
                    if (MODE == 2) sk[u] = LN::ld(Zskip + fd_nhwc(n, Hin, qy, Win, qx, C, qg));
                }
            }
#pragma unroll
            for (int u = 0; u < UI; ++u) {
                const int px = base + u * npt;
                if (px < npx_in) {
                    vec av = LN::zero();
                    if (oki[u]) {
                        av = from_skip ? fd_bn_act4<ACT2>(v[u], s1, t1) : fd_bn_act4<ACT_IN>(v[u], s1, t1);
                        if (MODE == 2) av += fd_bn_act4<ACT2>(sk[u], s2, t2);
                    }
                    LN::lds_st(s_in + px * PSTR + c4 * N, av);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NWREG; ++j) { const int i = tid + 256 * j; if (i < K * K * CB) s_w[i] = wreg[j]; }
    __syncthreads();

    // producer's tables.  MODE 3 (concatenating consumer): channels [0, csplit) belong to the low-resolution producer (pitch
    // csplit: 2x2 sum, mask, BN partials as in MODE 1), the rest to the skip tensor (pitch C - csplit: full-resolution gradient
    // into its skip-gradient buffer, masked later by the skip source's own consumer)
    const bool to_skip = MODE == 3 && cg >= csplit;
    const int Cp = MODE == 3 ? csplit : C, C2 = MODE == 3 ? C - csplit : C, cl = to_skip ? cg - csplit : cg;
    vec sc = LN::zero(), sh = LN::zero(), mu = LN::zero(), is = LN::zero();
    if (c_ok && !to_skip) { sc = LN::ldf(st_in + FD_ST_SCALE * Cp + cg); sh = LN::ldf(st_in + FD_ST_SHIFT * Cp + cg); mu = LN::ldf(st_in + FD_ST_MEAN * Cp + cg); is = LN::ldf(st_in + FD_ST_INVSTD * Cp + cg); }
    vec ssum = LN::zero(), ssx = LN::zero();
    auto din_at = [&](int iy, int ix) {                    // gradient w.r.t. the conv input at tile-local (iy, ix)
        vec acc = LN::zero();
        const int gy = iy0 + iy, gx = ix0 + ix;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int ny = gy + P - ky;
            if (S == 2 && (ny & 1)) continue;
            const int py = (S == 2 ? (ny >> 1) : ny) - oyb;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int nx = gx + P - kx;
                if (S == 2 && (nx & 1)) continue;
                const int pxx = (S == 2 ? (nx >> 1) : nx) - oxb;
                acc += LN::lds_ld(s_dz + (py * PW + pxx) * PSTR + c4 * N) * LN::ldf(s_w + (ky * K + kx) * CB + c4 * N);
            }
        }
        return acc;
    };
    if (MODE == 0 && S == 1) {
        // stride 1: din = correlation of the dz patch with the FLIPPED taps -- same strip scheme as the forward kernel: 4 adjacent
        // positions share K + 3 patch vectors per tap row (13 LDS reads per 20 FMAs instead of 40)
        const int TWS = TW >> 2;
        for (int st = pt; st < TH * TWS; st += npt) {
            const int iy = st / TWS, ix = (st - iy * TWS) * 4;
            const int gy = iy0 + iy;
            if (!c_ok || gy >= Hin) continue;
            vec z[4], sgv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                      // requested before the tap loop; clamped column, validity re-checked at the store
                const int gx = ix0 + ix + j, qx = gx < Win ? gx : Win - 1;
                const long o = fd_nhwc(n, Hin, gy, Win, qx, C, cg);
                z[j] = LN::ld(Zin + o);
                sgv[j] = ADD_SG ? LN::ld(SG + o) : LN::zero();
            }
            vec acc[4] = {LN::zero(), LN::zero(), LN::zero(), LN::zero()};
#pragma unroll UNR_TAPROWS
            for (int a = 0; a < K; ++a) {
                const lds_t *row = s_dz + ((iy + a) * PW + ix) * PSTR + c4 * N;
                vec r[K + 3];
#pragma unroll
                for (int i = 0; i < K + 3; ++i) r[i] = LN::lds_ld(row + i * PSTR);
#pragma unroll
                for (int b = 0; b < K; ++b) {
                    const vec wv = LN::ldf(s_w + ((K - 1 - a) * K + (K - 1 - b)) * CB + c4 * N);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += r[j + b] * wv;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gx = ix0 + ix + j;
                if (gx >= Win) continue;
                const long o = fd_nhwc(n, Hin, gy, Win, gx, C, cg);
                vec v = acc[j];
                if (ADD_SG) v += sgv[j];
                v = LN::round(v * fd_actmask4<ACT_IN>(z[j] * sc + sh));
                LN::st(Gin + o, v);
                ssum += v; ssx += v * ((z[j] - mu) * is);
            }
        }
    } else if (MODE == 0) {
        for (int p = pt; p < TH * TW; p += npt) {
            const int iy = p / TW, ix = p - iy * TW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            if (!c_ok || gy >= Hin || gx >= Win) continue;
            const long o = fd_nhwc(n, Hin, gy, Win, gx, C, cg);
            const vec z = LN::ld(Zin + o);                 // requested before the tap loop: the latency hides behind the LDS work
            vec sgv = LN::zero();
            if (ADD_SG) sgv = LN::ld(SG + o);
            vec v = din_at(iy, ix);
            if (ADD_SG) v += sgv;
            v = LN::round(v * fd_actmask4<ACT_IN>(z * sc + sh));
            LN::st(Gin + o, v);
            ssum += v; ssx += v * ((z - mu) * is);
        }
    } else {
        const int TH2 = TH >> 1, TW2 = TW >> 1, Hs = Hin >> 1, Ws = Win >> 1;
        for (int p = pt; p < TH2 * TW2; p += npt) {
            const int ly = p / TW2, lx = p - ly * TW2;
            const int gy = iy0 + 2 * ly, gx = ix0 + 2 * lx;     // top-left full-res position of the 2x2 block
            if (!c_ok || gy >= Hin || gx >= Win) continue;
            const long ol = fd_nhwc(n, Hs, (gy >> 1), Ws, (gx >> 1), Cp, (to_skip ? 0 : cg));
            const vec z = LN::ld(Zin + ol);                // requested before the tap loop (unused by the lanes that feed the skip tensor)
            vec d00, d01, d10, d11;
            if (S == 1) {
                // the four positions of the block read a (K+1) x (K+1) window of the dz patch: walk it row by row, every row feeds
                // the block's upper row with tap row ky = K-1-r and its lower row with ky = K-r (one pass over LDS, few registers)
                d00 = LN::zero(); d01 = LN::zero(); d10 = LN::zero(); d11 = LN::zero();
                const int pyb = gy + P - (K - 1) - oyb, pxb = gx + P - (K - 1) - oxb;
#pragma unroll 1
                for (int r = 0; r <= K; ++r) {
                    vec v[K + 1];
                    const lds_t *row = s_dz + ((pyb + r) * PW + pxb) * PSTR + c4 * N;
#pragma unroll
                    for (int c = 0; c <= K; ++c) v[c] = LN::lds_ld(row + c * PSTR);
                    if (r < K) {
                        const float *wr = s_w + (K - 1 - r) * K * CB + c4 * N;
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) { const vec wv = LN::ldf(wr + kx * CB); d00 += v[K - 1 - kx] * wv; d01 += v[K - kx] * wv; }
                    }
                    if (r > 0) {
                        const float *wr = s_w + (K - r) * K * CB + c4 * N;
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) { const vec wv = LN::ldf(wr + kx * CB); d10 += v[K - 1 - kx] * wv; d11 += v[K - kx] * wv; }
                    }
                }
            } else {
                d00 = din_at(2 * ly, 2 * lx); d01 = din_at(2 * ly, 2 * lx + 1); d10 = din_at(2 * ly + 1, 2 * lx); d11 = din_at(2 * ly + 1, 2 * lx + 1);
            }
            if (MODE == 2 || to_skip) {
                const long o = fd_nhwc(n, Hin, gy, Win, gx, C2, cl);
                LN::st(SGout + o, d00); LN::st(SGout + o + C2, d01);
                LN::st(SGout + o + (long)Win * C2, d10); LN::st(SGout + o + (long)Win * C2 + C2, d11);
                if (to_skip) continue;
            }
            vec v = (d00 + d01) + (d10 + d11);
            v = LN::round(v * fd_actmask4<ACT_IN>(z * sc + sh));
            LN::st(Gin + ol, v);
            ssum += v; ssx += v * ((z - mu) * is);
        }
    }
    // ---- backward-weights from the same two patches: work-item = (channel group c4, tap row ky, pixel group pg) owns the K taps of row ky (K
    // accumulators) and walks the owned output strips pg, pg + ngroups, ...; the pixel groups meet once, through LDS, below ----
    constexpr int NIN = 3 * S + K;
    const int ngroups = npt / K, ky_w = pt % K, pg = pt / K;
    const bool worker = pg < ngroups;
    vec wacc[K];
#pragma unroll
    for (int t = 0; t < K; ++t) wacc[t] = LN::zero();
    if (worker) {
        const int OTWS = OTW >> 2, nstrips = OTH * OTWS;
        for (int s = pg; s < nstrips; s += ngroups) {
            const int oy = s / OTWS, ox = (s - oy * OTWS) * 4;
            const lds_t *row = s_in + ((oy * S + ky_w) * TW_in + ox * S) * PSTR + c4 * N;
            const lds_t *dzp = s_dz + ((oy0 + oy - oyb) * PW + (ox0 + ox - oxb)) * PSTR + c4 * N;
            vec r[NIN], dzv[4];
#pragma unroll
            for (int i = 0; i < NIN; ++i) r[i] = LN::lds_ld(row + i * PSTR);
#pragma unroll
            for (int j = 0; j < 4; ++j) dzv[j] = LN::lds_ld(dzp + j * PSTR);
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) wacc[kx] += r[j * S + kx] * dzv[j];
        }
    }
    float *red = smem;
    fd_wg_stat_add<FD_STAT_BWD>(ssum, ssx, red, lanes_c, tid, sr, (long)bm.z * grid_x + bm.x, Cp, c0);
    // the pixel groups' tap sums meet in LDS: red[pg][ky*K + kx][c4] (fixed order -> deterministic); one partial row per tile
    __syncthreads();
    if (worker) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) LN::stf(red + ((pg * K * K + ky_w * K + kx) * lanes_c + c4) * N, wacc[kx]);
    }
    __syncthreads();
    {
        const long blk = (long)bm.z * grid_x + bm.x;
        for (int i = tid; i < K * K * lanes_c; i += 256) {
            const int t = i >> cbq, cc = i & (lanes_c - 1);
            if (c0 + cc * N < C) {
                vec a = LN::zero();
                for (int g = 0; g < ngroups; ++g) a += LN::ldf(red + ((g * K * K + t) * lanes_c + cc) * N);
                LN::stf(wpart + (blk * K * K + t) * C + c0 + cc * N, a);
            }
        }
    }
}

// Depthwise backward, ONE workgroup per tile for both gradients ("vertical" fusion of fd_dw_dgrad and fd_dw_wgrad): the two kernels stage nearly the
// same data -- dz of the outputs around the tile (from G, z, the BatchNorm-backward coefficients) and the unit's forward input around the tile
// (re-created from the producer's raw output) -- and each spends most of its life waiting for those loads.  Here a tile's workgroup stages
// both patches once (all loads of a work-item in flight together), runs the backward-data taps and the weight-gradient taps from LDS, and
// leaves the producer's gradient, its BatchNorm partial sums and one weight-gradient partial row.  Tiles are INPUT-space (TH x TW, multiples of
// the stride); an output belongs to the tile its receptive field starts in.
template <typename T, int K, int S, int MODE, int ACT_IN, int ACT2, int ADD_SG, int N, bool FIN = false>
__global__ void __launch_bounds__(256)
fd_dw_bwd1(const fd_dw_bwd_args<T> a)
{
    fd_dw_bwd1_body<T, K, S, MODE, ACT_IN, ACT2, ADD_SG, N, FIN>(a.G, a.Z, a.coef, a.w, a.Zin, a.st_in, a.Zskip, a.st_skip, a.SG, a.Gin, a.SGout, a.sr, a.wpart,
                                                         a.Hin, a.Win, a.Ho, a.Wo, a.C, a.cbq, a.d_th, a.d_tw, a.d_tiles_x, a.csplit, a.pstr, fd_xcd_image_map(), (int)gridDim.x, a.fin);
}


// ------------------------------------------------------------------------------------------------
// Backward of the 3x3 STRIDE-2 depthwise units on the large maps, register-window form (the forward's fd_dw3_rows_train design: no LDS
// staging; channel-group count C/4 a power of two in 8 ... 64).  Two kernels:
//
// fd_dw3s2_dgrad_rows -- a work-item is q = x_in * (C/4) + c4, one 16-byte channel group of one INPUT column, and walks down pairs of input
// rows (2b, 2b+1).  With  out(oy, ox) = sum in(2oy-1+ky, 2ox-1+kx) w[ky][kx]  an input column x = 2a receives only ox = a through kx = 1, a
// column x = 2a+1 receives ox = a through kx = 2 and ox = a+1 through kx = 0; rows likewise.  So per row pair the work-item needs dz at two
// columns (A = x/2, B = x/2 + 1) of two output rows (b: kept from the previous pair, b+1: loaded), and six vector FMAs with taps selected
// once by the column's parity:   din(2b, x)   = dzA(b) wA[1] + dzB(b) wB[1]
//                                din(2b+1, x) = dzA(b) wA[2] + dzB(b) wB[2] + dzA(b+1) wA[0] + dzB(b+1) wB[0].
// Epilogue per input pixel as in fd_dw_dgrad (skip-gradient add, producer's activation mask, rounding, BN partials of the producer):
// part[blk*2*C + {0, C} + c], blk = (image * gridDim.y + strip) * gridDim.x + column block.
// ------------------------------------------------------------------------------------------------
template <typename T, int ACT_IN, int ADD_SG>
__global__ void __launch_bounds__(256)
fd_dw3s2_dgrad_rows(const T *__restrict__ G, const T *__restrict__ Z, const float *__restrict__ coef, const float *__restrict__ w,
                    const T *__restrict__ Zin, const float *__restrict__ st_in, const T *__restrict__ SG, T *__restrict__ Gin,
                    fd_stat_rows sr, int Hin, int Win, int Ho, int Wo, int C, int TH2)
{
    __shared__ float red[4 * 64 * 8 + 2 * 256];             // wave sums + the workgroup's totals (fd_wg_stat_add)
    const int CG = C >> 2;
    const fd_blk3 blk = fd_xcd_image_map();
    const int tid = threadIdx.x;
    const int q = blk.x * 256 + tid;
    const bool live = q < Win * CG;
    const int qq = live ? q : 0;
    const int x = qq / CG, c4 = qq - x * CG;
    const int n = blk.z;
    const int b0 = blk.y * TH2, H2 = Hin >> 1;
    const int b1 = (b0 + TH2 < H2) ? b0 + TH2 : H2;
    const int cg = c4 * 4;
    const bool odd = x & 1;
    const int oxA = x >> 1, oxB = (x >> 1) + 1;
    const bool okB = odd && oxB < Wo;
    fd_f32x4 wA[3], wB[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ka = odd ? 2 : 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wA[ky][j] = w[(cg + j) * 9 + ky * 3 + ka];
            const float wb = w[(cg + j) * 9 + ky * 3 + 0];
            wB[ky][j] = okB ? wb : 0.0f;
        }
    }
    const fd_f32x4 cA = fd_ld4(coef + FD_CF_A * C + cg), c1 = fd_ld4(coef + FD_CF_C1 * C + cg), cM = fd_ld4(coef + FD_CF_MU * C + cg), c2 = fd_ld4(coef + FD_CF_C2 * C + cg);
    const fd_f32x4 sc = fd_ld4(st_in + FD_ST_SCALE * C + cg), sh = fd_ld4(st_in + FD_ST_SHIFT * C + cg);
    const fd_f32x4 mu = fd_ld4(st_in + FD_ST_MEAN * C + cg), is = fd_ld4(st_in + FD_ST_INVSTD * C + cg);
    const int qB = okB ? oxB : oxA;                          // clamped column: the load is always issued, wB is zero where B does not exist
    auto load_dz = [&](int oy, fd_f32x4 &a, fd_f32x4 &b) {
        const bool ok = oy < Ho;
        const long row = ((long)n * Ho + (ok ? oy : Ho - 1)) * Wo;
        const long oa = (row + oxA) * C + cg, ob = (row + qB) * C + cg;
        const fd_f32x4 da = fd_dz4(fd_ld4(G + oa), fd_ld4(Z + oa), cA, c1, cM, c2), db = fd_dz4(fd_ld4(G + ob), fd_ld4(Z + ob), cA, c1, cM, c2);
        a = ok ? da : fd_zero4(); b = ok ? db : fd_zero4();
    };
    fd_f32x4 ssum = fd_zero4(), ssx = fd_zero4();
    fd_f32x4 dA, dB, nA, nB;
    load_dz(b0, dA, dB);
    for (int b = b0; b < b1; ++b) {
        const long o0 = fd_nhwc(n, Hin, 2 * b, Win, x, C, cg), o1 = o0 + (long)Win * C;
        const fd_f32x4 z0 = fd_ld4(Zin + o0), z1 = fd_ld4(Zin + o1);              // requested together with the next dz row
        fd_f32x4 g0 = fd_zero4(), g1 = fd_zero4();
        if (ADD_SG) { g0 = fd_ld4(SG + o0); g1 = fd_ld4(SG + o1); }
        load_dz(b + 1, nA, nB);
        fd_f32x4 v0 = dA * wA[1] + dB * wB[1];
        fd_f32x4 v1 = (dA * wA[2] + dB * wB[2]) + (nA * wA[0] + nB * wB[0]);
        if (ADD_SG) { v0 += g0; v1 += g1; }
        v0 = fd_round4(T{}, v0 * fd_actmask4<ACT_IN>(z0 * sc + sh));
        v1 = fd_round4(T{}, v1 * fd_actmask4<ACT_IN>(z1 * sc + sh));
        if (live) {
            fd_st4(Gin + o0, v0); fd_st4(Gin + o1, v1);
            ssum += v0; ssx += v0 * ((z0 - mu) * is);
            ssum += v1; ssx += v1 * ((z1 - mu) * is);
        }
        dA = nA; dB = nB;
    }
    fd_wg_stat_add<FD_STAT_BWD>(ssum, ssx, red, CG, tid, sr, ((long)n * gridDim.y + blk.y) * gridDim.x + blk.x, C, 0);
}

// fd_dw3_wgrad_rows -- dW[c][ky][kx] = sum dz(oy, ox) a_in(S oy - 1 + ky, S ox - 1 + kx): a work-item is q = x_out * (C/4) + c4 and walks down TH
// output rows with the 3 x 3 window of the activated input in registers (as the forward) and nine vector accumulators; at the end the
// accumulators of the work-items that share a channel group meet by shuffle butterfly (within a wave) and one LDS hop (across the four
// waves).  wpart[(blk*9 + t)*C + c], tap-major like the other depthwise weight-gradient kernels.
template <typename T, int S, int ACT1>
__global__ void __launch_bounds__(256)
fd_dw3_wgrad_rows(const T *__restrict__ zin, const float *__restrict__ st1, const T *__restrict__ G, const T *__restrict__ Z,
                  const float *__restrict__ coef, float *__restrict__ wpart, int H, int W, int Ho, int Wo, int C, int TH)
{
    __shared__ float red[4 * 64 * 36];                      // [wave][channel group][9 taps][4]
    const int CG = C >> 2;
    const fd_blk3 blk = fd_xcd_image_map();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blk.x * 256 + tid;
    const bool live = q < Wo * CG;
    const int qq = live ? q : 0;
    const int xo = qq / CG, c4 = qq - xo * CG;
    const int n = blk.z, cg = c4 * 4;
    const int oy0 = blk.y * TH;
    const int oy1 = (oy0 + TH < Ho) ? oy0 + TH : Ho;
    const fd_f32x4 sc = fd_ld4(st1 + FD_ST_SCALE * C + cg), sh = fd_ld4(st1 + FD_ST_SHIFT * C + cg);
    const fd_f32x4 cA = fd_ld4(coef + FD_CF_A * C + cg), c1 = fd_ld4(coef + FD_CF_C1 * C + cg), cM = fd_ld4(coef + FD_CF_MU * C + cg), c2 = fd_ld4(coef + FD_CF_C2 * C + cg);
    const T *img = zin + (long)n * H * W * C + cg;
    const int x0 = xo * S - 1;
    const bool okl = x0 >= 0, okr = (x0 + 2) < W;
    const int xl = okl ? x0 : x0 + 1, xr = okr ? x0 + 2 : x0 + 1;
    auto load_row = [&](int iy, fd_f32x4 &l, fd_f32x4 &c, fd_f32x4 &r) {
        const bool oky = iy >= 0 && iy < H;
        const int qy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        const T *p = img + (long)qy * W * C;
        const fd_f32x4 vl = fd_ld4(p + (long)xl * C), vc = fd_ld4(p + (long)(x0 + 1) * C), vr = fd_ld4(p + (long)xr * C);
        l = (oky && okl) ? fd_bn_act4<ACT1>(vl, sc, sh) : fd_zero4();
        c = oky ? fd_bn_act4<ACT1>(vc, sc, sh) : fd_zero4();
        r = (oky && okr) ? fd_bn_act4<ACT1>(vr, sc, sh) : fd_zero4();
    };
    fd_f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = fd_zero4();
    fd_f32x4 r0l, r0c, r0r, r1l, r1c, r1r, r2l, r2c, r2r;
    load_row(S * oy0 - 1, r0l, r0c, r0r);
    if (S == 1) load_row(oy0, r1l, r1c, r1r);
    const T *gp = G + (((long)n * Ho + oy0) * Wo) * C + (long)qq * 4, *zp = Z + (((long)n * Ho + oy0) * Wo) * C + (long)qq * 4;
    for (int oy = oy0; oy < oy1; ++oy) {
        const fd_f32x4 gv = fd_ld4(gp), zv = fd_ld4(zp);
        if (S == 1) load_row(oy + 1, r2l, r2c, r2r);
        else { load_row(2 * oy, r1l, r1c, r1r); load_row(2 * oy + 1, r2l, r2c, r2r); }
        const fd_f32x4 dz = live ? fd_dz4(gv, zv, cA, c1, cM, c2) : fd_zero4();
        acc[0] += dz * r0l; acc[1] += dz * r0c; acc[2] += dz * r0r;
        acc[3] += dz * r1l; acc[4] += dz * r1c; acc[5] += dz * r1r;
        acc[6] += dz * r2l; acc[7] += dz * r2c; acc[8] += dz * r2r;
        gp += (long)Wo * C; zp += (long)Wo * C;
        if (S == 1) { r0l = r1l; r0c = r1c; r0r = r1r; r1l = r2l; r1c = r2c; r1r = r2r; }
        else { r0l = r2l; r0c = r2c; r0r = r2r; }
    }
    // lanes c4, c4 + CG, ... of a wave hold the same channel group (256 % CG == 0): butterfly, then the four wave sums through LDS
#pragma unroll
    for (int t = 0; t < 9; ++t)
        for (int m = CG; m < 64; m <<= 1) {
            acc[t].x += __shfl_xor(acc[t].x, m); acc[t].y += __shfl_xor(acc[t].y, m); acc[t].z += __shfl_xor(acc[t].z, m); acc[t].w += __shfl_xor(acc[t].w, m);
        }
    if (lane < CG) {
#pragma unroll
        for (int t = 0; t < 9; ++t) fd_st4(red + ((wave * CG + lane) * 9 + t) * 4, acc[t]);
    }
    __syncthreads();
    const long row = ((long)n * gridDim.y + blk.y) * gridDim.x + blk.x;
    for (int i = tid; i < 9 * CG; i += 256) {
        const int t = i / CG, c = i - t * CG;
        fd_f32x4 s = fd_ld4(red + ((0 * CG + c) * 9 + t) * 4);
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) s += fd_ld4(red + ((wv * CG + c) * 9 + t) * 4);
        fd_st4(wpart + (row * 9 + t) * C + c * 4, s);
    }
}

// ------------------------------------------------------------------------------------------------
// Stem backward-weights: dW[co][t] = sum_px dz[px][co] * patch[px][t], t over the 27 taps -- a [Cout x P] x [P x 27] matrix
// product over the P = B*Ho*Wo output pixels.  A workgroup walks blocks of 256 pixels (grid-stride), stages dz (formed from
// G, z on load) and the 27-tap input patches in LDS and feeds them to v_mfma_f32_32x32x2_f32: A[i = co][k = pixel],
// B[k = pixel][j = tap], each wave accumulating 64 pixels of every block.  wpart[blk][Cout*27], blk = blockIdx.x.
// (Staging the input band with 16-byte row loads as the forward kernel does and reading the B operand straight from it was measured
// slower: 58.6 vs 56.1 us bf16, 77 vs 72 us fp32 -- this kernel is one round of two workgroups per CU, bound by its dz staging and barriers.)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
fd_stem_wgrad(const float *__restrict__ x, const T *__restrict__ G, const T *__restrict__ Z,
              const float *__restrict__ coef, float *__restrict__ wpart, int B, int H, int W, int Cout, int nblocks)
{
    FD_DYN_SMEM(smem_raw);
    float *s_in = reinterpret_cast<float *>(smem_raw);     // [256][33]: taps 0..26, zeros in 27..31
    float *s_dz = s_in + 256 * 33;                          // [256][Cout + 1]
    const int Ho = H >> 1, Wo = W >> 1;
    const long npix = (long)B * Ho * Wo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int DP = Cout + 1;
    const int cgroups = (Cout + 31) / 32;                   // 32-channel column groups of the A operand (<= 2 supported)
    fd_f32x16 acc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.0f;
    for (int t = 27; t < 32; ++t) s_in[tid * 33 + t] = 0.0f;
    for (int pb = blockIdx.x; pb < nblocks; pb += gridDim.x) {
        const long p = (long)pb * 256 + tid;
        const bool valid = p < npix;
        int n = 0, oy = 0, ox = 0;
        if (valid) { ox = (int)(p % Wo); const long t = p / Wo; oy = (int)(t % Ho); n = (int)(t / Ho); }
        __syncthreads();                                    // the previous block's fragment reads are done
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                    const bool ok = valid && iy >= 0 && iy < H && ix >= 0 && ix < W;
                    const int qy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), qx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);   // branch-free: clamp, load, select
                    const float v = x[fd_nhwc(n, 3, c, H, qy, W, qx)];
                    s_in[tid * 33 + (c * 3 + ky) * 3 + kx] = ok ? v : 0.0f;
                }
        for (int c = 0; c < Cout; c += 4) {
            fd_f32x4 dz = fd_zero4();
            if (valid) dz = fd_dz4(fd_ld4(G + p * Cout + c), fd_ld4(Z + p * Cout + c), fd_ld4(coef + FD_CF_A * Cout + c), fd_ld4(coef + FD_CF_C1 * Cout + c),
                                   fd_ld4(coef + FD_CF_MU * Cout + c), fd_ld4(coef + FD_CF_C2 * Cout + c));
            s_dz[tid * DP + c] = dz.x; s_dz[tid * DP + c + 1] = dz.y; s_dz[tid * DP + c + 2] = dz.z; s_dz[tid * DP + c + 3] = dz.w;
        }
        __syncthreads();
        const int i = lane & 31, kk = lane >> 5;
#pragma unroll 4
        for (int q = 0; q < 32; ++q) {
            const int px = wave * 64 + 2 * q + kk;
            const float b = s_in[px * 33 + i];
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (g < cgroups) {
                    const int co = g * 32 + i;
                    const float a = co < Cout ? s_dz[px * DP + co] : 0.0f;
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
                }
        }
    }
    // cross-wave reduction through LDS: red[wave][32 co][33]
    __syncthreads();
    float *red = s_in;                                      // 4 * 32 * 33 floats <= 256 * 33
    for (int g = 0; g < cgroups; ++g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 33 + (lane & 31)] = acc[g][r];
        __syncthreads();
        for (int o = tid; o < 32 * 27; o += 256) {
            const int co = o / 27, t = o - co * 27;
            if (g * 32 + co < Cout)
                wpart[(long)blockIdx.x * Cout * 27 + (g * 32 + co) * 27 + t] =
                    (red[(0 * 32 + co) * 33 + t] + red[(1 * 32 + co) * 33 + t]) + (red[(2 * 32 + co) * 33 + t] + red[(3 * 32 + co) * 33 + t]);
        }
        __syncthreads();
    }
}
