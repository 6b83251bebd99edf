// fd_bn_stats.h -- train-mode BatchNorm from statistics rows: the per-channel table, its derivation inside a consumer kernel's prologue (or as a launch
// of its own) and the side effects of nn.BatchNorm2d in .train() (running statistics, num_batches_tracked; reference modules at imagenet/mobilenet.py:25,32,36
// and models.py:66,73).  Shared by the train kernels (fd_kernels_train.h, fd_kernels_train_h16.h) and the train mode of fd_pw_gemm16_f32.
#pragma once
#include "fd_device.h"

// per-channel table written by the finalising workgroup (fd_stat_table_block / fd_bn_finalize_rows_f32):  [0..C) scale, [C..2C) shift, [2C..3C) mean, [3C..4C) invstd
#define FD_ST_SCALE 0
#define FD_ST_SHIFT 1
#define FD_ST_MEAN 2
#define FD_ST_INVSTD 3

// ------------------------------------------------------------------------------------------------
// BatchNorm finalisation of the PRODUCER inside its consumer.  The producer's workgroups have added their partial sums into the unit's
// statistics rows (fd_stat_add); every consumer workgroup turns the totals of the channels IT needs into (scale, shift) in its prologue --
// nr <= 16 rows x 3 bins x 2 sums of 8 bytes per channel, one round trip that runs under the consumer's first data loads -- with the same code
// on the same integers: identical bits in every workgroup.  The workgroup the caller designates (`writer`) also stores the table the backward
// pass and the skip consumers read, updates the running statistics and num_batches_tracked.  Replaces a finalisation launch (~5 us at the
// per-launch floor) between every producer and its consumer.
// ------------------------------------------------------------------------------------------------
struct fd_bn_fin {
    const long long *rows;           // null: the table st1 was finalised by a launch of its own (fd_bn_finalize_rows_f32)
    int nr, cs;                      // statistics rows of the producer (power of two), their channel pitch
    double n, n_unbiased, inv_n;     // pixels per channel, the count of the unbiased variance, 1 / n
    float eps, momentum;
    const float *gamma, *beta;
    float *run_mean, *run_var, *st;
    long long *nbt;
};
struct fd_bn_coef { double mean, var; float invstd, scale; };
// one channel's coefficients from its totals:  mean = S/n, var_b = Q/n - mean^2 (biased, used to normalise), scale = gamma * invstd.
// Runs in the prologue of EVERY consumer workgroup, so it is written for few instructions: the cancellation-prone part (mean, variance) in double with
// the reciprocal count from the host, 1 / sqrt(var + eps) as v_rsq_f32 + one Newton step (<= 1.2e-7 relative: the table is fp32 anyway).
__device__ __forceinline__ float fd_rsqrt_nr(float x)
{
#ifdef FD_EMU
    float y = 1.0f / sqrtf(x);
#else
    float y = __builtin_amdgcn_rsqf(x);
#endif
    return y * (1.5f - 0.5f * x * y * y);
}
__device__ __forceinline__ fd_bn_coef fd_bn_coef_of(double s, double q, double inv_n, float eps, float g_c)
{
    fd_bn_coef k;
    k.mean = s * inv_n;
    k.var = q * inv_n - k.mean * k.mean;
    if (k.var < 0.0) k.var = 0.0;
    k.invstd = fd_rsqrt_nr((float)(k.var + (double)eps));
    k.scale = g_c * k.invstd;
    return k;
}
__device__ __forceinline__ float fd_bn_shift_of(const fd_bn_coef &k, float b_c) { return (float)((double)b_c - k.mean * (double)k.scale); }
// the finalising workgroup's side effects for channel c: the table [4][C], the running statistics (running_var takes var_b * n_u / (n_u - 1)),
// num_batches_tracked (nn.BatchNorm2d: the owner of channel 0 counts the batch)
__device__ __forceinline__ void fd_bn_publish(const fd_bn_fin &f, int C, int c, const fd_bn_coef &k, float b_c, float rm_c, float rv_c)
{
    f.st[FD_ST_SCALE * C + c] = k.scale;
    f.st[FD_ST_SHIFT * C + c] = fd_bn_shift_of(k, b_c);
    f.st[FD_ST_MEAN * C + c] = (float)k.mean;
    f.st[FD_ST_INVSTD * C + c] = k.invstd;
    f.run_mean[c] = (float)((1.0 - f.momentum) * rm_c + f.momentum * k.mean);
    f.run_var[c] = (float)((1.0 - f.momentum) * rv_c + f.momentum * k.var * (f.n_unbiased / (f.n_unbiased - 1.0)));
    if (c == 0 && f.nbt) f.nbt[0] += 1;
}
// Channel-block form (depthwise consumers): 256 work-items, the CB <= 64 channels [c0, c0 + CB) of the C-channel producer; a channel's rows are dealt
// to the 256 / CB work-items that share it (integer sums per work-item, their doubles meet in LDS in fixed order).
// sh: >= 4 KiB of LDS that is dead until the next barrier; s_st: [2][CB] floats (scale, shift) that nothing else touches.
__device__ __forceinline__ void fd_stat_table_block(const fd_bn_fin &f, double *sh, float *s_st, int c0, int CB, int C, int tid, bool writer)
{
    const int ch = tid & (CB - 1), rg = tid / CB, RG = 256 / CB;
    const int c = c0 + ch;
    const bool ok = c < C;
    float g_c = 0.0f, b_c = 0.0f, rm_c = 0.0f, rv_c = 0.0f;
    if (rg == 0 && ok) { g_c = f.gamma[c]; b_c = f.beta[c]; if (writer) { rm_c = f.run_mean[c]; rv_c = f.run_var[c]; } }
    double s = 0.0, q = 0.0;
    if (ok && rg < f.nr) {
        s = fd_stat_total<FD_STAT_FWD>(f.rows, f.nr, f.cs, 0, c, rg, RG);
        q = fd_stat_total<FD_STAT_FWD>(f.rows, f.nr, f.cs, 1, c, rg, RG);
    }
    sh[2 * tid] = s; sh[2 * tid + 1] = q;
    __syncthreads();
    if (rg == 0) {
        s = 0.0; q = 0.0;
        for (int r = 0; r < RG; ++r) { s += sh[2 * (r * CB + ch)]; q += sh[2 * (r * CB + ch) + 1]; }
        const fd_bn_coef k = fd_bn_coef_of(s, q, f.inv_n, f.eps, g_c);
        s_st[ch] = ok ? k.scale : 0.0f; s_st[CB + ch] = ok ? fd_bn_shift_of(k, b_c) : 0.0f;
        if (writer && ok) fd_bn_publish(f, C, c, k, b_c, rm_c, rv_c);
    }
    __syncthreads();
}
// All-channels form (pointwise consumers, which contract over every channel of the producer): work-item tid of NT finalises channels tid, tid + NT,
// ... < C and hands (scale, shift) to `put(c, scale, shift)` (the consumer's LDS table); no barrier inside -- the caller's own "table visible" barrier
// follows.  Called AFTER the consumer has issued its first operand loads.  Channels >= C up to Cpad get (0, 0) (ragged last K tile).
template <int NT, typename PUT>
__device__ __forceinline__ void fd_stat_table_all(const fd_bn_fin &f, int C, int Cpad, int tid, bool writer, PUT &&put)
{
    // two phases, so that the integer loads of ALL of this work-item's channels are in flight together (one channel at a time was a chain of
    // dependent round trips: 8.6 us in front of a 1024-channel GEMM at batch 32)
    constexpr int MAXQ = 1024 / NT;                          // Cpad <= 1024 (checked by the plans)
    double s[MAXQ], q[MAXQ];
    float g[MAXQ], b[MAXQ], rm[MAXQ], rv[MAXQ];
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
        const int c = tid + NT * i;
        s[i] = 0.0; q[i] = 0.0; g[i] = 0.0f; b[i] = 0.0f; rm[i] = 0.0f; rv[i] = 0.0f;
        if (c < C) {                                         // (work-items beyond the channel count issue no loads: a 1-channel head must not read its rows 1024 times)
            s[i] = fd_stat_total<FD_STAT_FWD>(f.rows, f.nr, f.cs, 0, c, 0, 1); q[i] = fd_stat_total<FD_STAT_FWD>(f.rows, f.nr, f.cs, 1, c, 0, 1);
            g[i] = f.gamma[c]; b[i] = f.beta[c];
            if (writer) { rm[i] = f.run_mean[c]; rv[i] = f.run_var[c]; }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
        const int c = tid + NT * i;
        if (c >= Cpad) continue;
        if (c >= C) { put(c, 0.0f, 0.0f); continue; }
        const fd_bn_coef k = fd_bn_coef_of(s[i], q[i], f.inv_n, f.eps, g[i]);
        put(c, k.scale, fd_bn_shift_of(k, b[i]));
        if (writer) fd_bn_publish(f, C, c, k, b[i], rm[i], rv[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm finalisation as a launch of its own (where no consumer kernel does it: plans with FD_TUNE_NO_CONSUMER_FINALIZE, the head's 1-channel
// statistics, units with many rows, unusual unit combinations): statistics rows -> table + running statistics.
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256)
fd_bn_finalize_rows_f32(const fd_bn_fin f, int C)
{
    // 16 channels per workgroup, a channel's rows dealt to 16 work-items: one batch of loads whatever the row count (one work-item per channel walked
    // 16 rows x 6 integers as a chain of dependent batches: 11 us per launch)
    __shared__ double sh[512];
    __shared__ float s_st[2 * 16];
    fd_stat_table_block(f, sh, s_st, blockIdx.x * 16, 16, C, threadIdx.x, true);
}

