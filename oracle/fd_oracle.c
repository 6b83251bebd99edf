/*
 * fd_oracle.c -- CPU restatement of the FastDepth MobileNetSkipAdd hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: a deliberately plain, unfused, NCHW, op-by-op restatement of what
 * the reference executes for `models.MobileNetSkipAdd.forward`.  It is compiled by oracle/build.py
 * (gcc -O2 -fopenmp) and may be loaded only by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py -- never by the product package (fast-depth_amd/), which has no
 * CPU path at all.
 *
 * Where the arithmetic lives in the reference: in PyTorch (ATen), a third-party dependency that
 * /root/reference does not vendor and pins only in prose ("PyTorch v0.4.1", README.md:19).  The
 * semantics restated here are the published ones of the five ops the reference calls:
 *   nn.Conv2d (zero padding, cross-correlation, groups)   imagenet/mobilenet.py:24,31,35; models.py:65,72
 *   nn.BatchNorm2d (eps 1e-5, momentum 0.1)               imagenet/mobilenet.py:25,32,36; models.py:66,73
 *   nn.ReLU6 / nn.ReLU                                    imagenet/mobilenet.py:16-20; models.py:67,74
 *   F.interpolate(scale_factor=2, mode='nearest')         models.py:723
 *   tensor +                                              models.py:725,727,729
 * Parity pin: oracle/make_golden.py runs the reference module itself (imported from /root/reference,
 * container torch 2.10 CPU) and this file on the same seeded weights/inputs; tests/golden/ holds the
 * reference's outputs, tests/test_oracle.py re-checks this file against them on every run.
 *
 * Numerics: every convolution output is accumulated in double and rounded once to float, BatchNorm
 * statistics are accumulated in double.  That makes the oracle at least as accurate as the
 * reference's own fp32 kernels (measured distance to the reference: see tests/golden/README.md).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FDO_ACT_NONE 0
#define FDO_ACT_RELU 1
#define FDO_ACT_RELU6 2

/* One "unit" = Conv2d(bias=False) -> BatchNorm2d -> activation, the only composite the path uses
 * (reference conv_bn mobilenet.py:22-27, each half of conv_dw :29-38, depthwise/pointwise models.py:61-75). */
typedef struct {
    int32_t cin, cout, ksize, stride, groups, act;
    const float *w;      /* [cout][cin/groups][k][k]  (torch Conv2d.weight layout) */
    const float *gamma;  /* BatchNorm2d.weight */
    const float *beta;   /* BatchNorm2d.bias */
    float *mean;         /* running_mean (updated in train mode) */
    float *var;          /* running_var  (updated in train mode) */
} fdo_unit;

/* y[n,co,oy,ox] = sum_{ci in group, ky, kx} w[co,ci,ky,kx] * x[n, g*cin_g+ci, oy*s-p+ky, ox*s-p+kx], zero padded. */
void fdo_conv2d(const float *x, const float *w, float *y, int N, int Cin, int H, int W,
                int Cout, int k, int stride, int pad, int groups)
{
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int cin_g = Cin / groups, cout_g = Cout / groups;
    const long plane_in = (long)H * W, plane_out = (long)Ho * Wo;
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * plane_out);
#pragma omp for collapse(2) schedule(dynamic, 1)
        for (int n = 0; n < N; ++n)
            for (int co = 0; co < Cout; ++co) {
                const int g = co / cout_g;
                for (long i = 0; i < plane_out; ++i) acc[i] = 0.0;
                for (int ci = 0; ci < cin_g; ++ci) {
                    const float *xp = x + ((long)n * Cin + (long)g * cin_g + ci) * plane_in;
                    const float *wp = w + (((long)co * cin_g + ci) * k) * k;
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx) {
                            const double wv = wp[ky * k + kx];
                            /* valid ox range: 0 <= ox*stride - pad + kx < W */
                            int ox0 = 0, ox1 = Wo;
                            while (ox0 < Wo && ox0 * stride - pad + kx < 0) ++ox0;
                            while (ox1 > ox0 && (ox1 - 1) * stride - pad + kx >= W) --ox1;
                            for (int oy = 0; oy < Ho; ++oy) {
                                const int iy = oy * stride - pad + ky;
                                if (iy < 0 || iy >= H) continue;
                                const float *xr = xp + (long)iy * W - pad + kx;
                                double *ar = acc + (long)oy * Wo;
                                if (stride == 1)
                                    for (int ox = ox0; ox < ox1; ++ox) ar[ox] += wv * xr[ox];
                                else
                                    for (int ox = ox0; ox < ox1; ++ox) ar[ox] += wv * xr[(long)ox * stride];
                            }
                        }
                }
                float *yp = y + ((long)n * Cout + co) * plane_out;
                for (long i = 0; i < plane_out; ++i) yp[i] = (float)acc[i];
            }
        free(acc);
    }
}

/* BatchNorm2d, inference form: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta. In place. */
void fdo_batchnorm_eval(float *x, int N, int C, long HW, const float *gamma, const float *beta,
                        const float *mean, const float *var, float eps)
{
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float inv = 1.0f / sqrtf(var[c] + eps);
            float *p = x + ((long)n * C + c) * HW;
            for (long i = 0; i < HW; ++i) p[i] = (p[i] - mean[c]) * inv * gamma[c] + beta[c];
        }
}

/* BatchNorm2d, training form: normalise with the batch mean and the BIASED batch variance; update
 * running_mean / running_var with momentum using the UNBIASED variance (torch semantics, verified
 * against the container's torch in SURVEY.md Appendix F).  Optionally returns the batch statistics. */
void fdo_batchnorm_train(float *x, int N, int C, long HW, const float *gamma, const float *beta,
                         float *run_mean, float *run_var, float eps, float momentum,
                         float *save_mean, float *save_invstd)
{
    const double cnt = (double)N * (double)HW;
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int n = 0; n < N; ++n) {
            const float *p = x + ((long)n * C + c) * HW;
            for (long i = 0; i < HW; ++i) s += p[i];
        }
        const double mu = s / cnt;
        double q = 0.0;
        for (int n = 0; n < N; ++n) {
            const float *p = x + ((long)n * C + c) * HW;
            for (long i = 0; i < HW; ++i) { const double d = p[i] - mu; q += d * d; }
        }
        const double var_b = q / cnt;
        const double inv = 1.0 / sqrt(var_b + (double)eps);
        for (int n = 0; n < N; ++n) {
            float *p = x + ((long)n * C + c) * HW;
            for (long i = 0; i < HW; ++i) p[i] = (float)((p[i] - mu) * inv * gamma[c] + beta[c]);
        }
        if (run_mean) run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
        if (run_var) run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * (q / (cnt - 1.0)));
        if (save_mean) save_mean[c] = (float)mu;
        if (save_invstd) save_invstd[c] = (float)inv;
    }
}

/* ReLU: max(x, 0).  ReLU6: min(max(x, 0), 6)  (nn.ReLU6 == hardtanh(0, 6)). */
void fdo_activation(float *x, long n, int act)
{
    if (act == FDO_ACT_RELU) {
#pragma omp parallel for
        for (long i = 0; i < n; ++i) x[i] = x[i] > 0.0f ? x[i] : 0.0f;
    } else if (act == FDO_ACT_RELU6) {
#pragma omp parallel for
        for (long i = 0; i < n; ++i) x[i] = x[i] < 0.0f ? 0.0f : (x[i] > 6.0f ? 6.0f : x[i]);
    }
}

/* F.interpolate(scale_factor=2, mode='nearest'): out[h][w] = in[h/2][w/2]  (models.py:723). */
void fdo_upsample_nearest2x(const float *x, float *y, long planes, int H, int W)
{
#pragma omp parallel for
    for (long p = 0; p < planes; ++p) {
        const float *xp = x + p * H * W;
        float *yp = y + p * 4 * H * W;
        for (int h = 0; h < 2 * H; ++h)
            for (int w = 0; w < 2 * W; ++w) yp[(long)h * 2 * W + w] = xp[(long)(h >> 1) * W + (w >> 1)];
    }
}

void fdo_add_inplace(float *x, const float *s, long n)
{
#pragma omp parallel for
    for (long i = 0; i < n; ++i) x[i] += s[i];
}

static long unit_out_hw(const fdo_unit *u, int H) { return (H + 2 * (u->ksize / 2) - u->ksize) / u->stride + 1; }

/* conv -> BN -> act on a fresh buffer; returns malloc'ed output, writes output spatial size. */
static float *run_unit(fdo_unit *u, const float *x, int N, int H, int W, int *Ho, int *Wo,
                       int train, float eps, float momentum)
{
    *Ho = (int)unit_out_hw(u, H);
    *Wo = (int)unit_out_hw(u, W);
    const long hw = (long)(*Ho) * (*Wo);
    float *y = (float *)malloc(sizeof(float) * (size_t)N * u->cout * hw);
    fdo_conv2d(x, u->w, y, N, u->cin, H, W, u->cout, u->ksize, u->stride, u->ksize / 2, u->groups);
    if (train)
        fdo_batchnorm_train(y, N, u->cout, hw, u->gamma, u->beta, u->mean, u->var, eps, momentum, 0, 0);
    else
        fdo_batchnorm_eval(y, N, u->cout, hw, u->gamma, u->beta, u->mean, u->var, eps);
    fdo_activation(y, (long)N * u->cout * hw, u->act);
    return y;
}

/*
 * MobileNetSkipAdd.forward, statement by statement (models.py:706-732):
 *   units[0]            = conv0 (stem)                       models.py:710-712, mobilenet.py:41
 *   units[1+2i], [2+2i] = conv{i+1}: dw3x3 unit, pw unit     i = 0..12, mobilenet.py:42-54
 *   x1 = conv1 out, x2 = conv3 out, x3 = conv5 out           models.py:714-719
 *   units[27+2j], [28+2j] = decode_conv{j+1}: dw5x5, pw      j = 0..4, models.py:720-722
 *   after each decode_conv1..5: nearest x2 (:723); then j+1 == 4: += x1, == 3: += x2, == 2: += x3 (:724-729)
 *   units[37]           = decode_conv6 (head)                models.py:731
 * `taps`, if non-NULL, receives a malloc'ed copy of every unit's output (38 pointers, NCHW), for
 * layer-wise parity checks; the caller frees them with fdo_free.
 * Returns 0, or -1 on a structural error.
 */
int fdo_skipadd_forward(fdo_unit *units, int n_units, const float *x, int N, int H, int W,
                        float *y, float **taps, int train, float eps, float momentum)
{
    if (n_units != 38) return -1;
    int h = H, w = W, ho, wo;
    float *cur = (float *)x, *skip[3] = {0, 0, 0};
    int skip_c[3] = {0, 0, 0}, skip_h[3] = {0, 0, 0};
    for (int i = 0; i < 27; ++i) {
        float *nxt = run_unit(&units[i], cur, N, h, w, &ho, &wo, train, eps, momentum);
        if (taps) { size_t b = sizeof(float) * (size_t)N * units[i].cout * ho * wo; taps[i] = (float *)malloc(b); memcpy(taps[i], nxt, b); }
        int is_skip = (cur == skip[0] || cur == skip[1] || cur == skip[2]);
        if (cur != x && !is_skip) free(cur);
        cur = nxt; h = ho; w = wo;
        if (i == 2) { skip[0] = cur; skip_c[0] = units[i].cout; skip_h[0] = h; }   /* x1 = conv1 output */
        if (i == 6) { skip[1] = cur; skip_c[1] = units[i].cout; skip_h[1] = h; }   /* x2 = conv3 output */
        if (i == 10) { skip[2] = cur; skip_c[2] = units[i].cout; skip_h[2] = h; }  /* x3 = conv5 output */
    }
    for (int j = 1; j <= 5; ++j) {
        for (int half = 0; half < 2; ++half) {
            const int i = 27 + 2 * (j - 1) + half;
            float *nxt = run_unit(&units[i], cur, N, h, w, &ho, &wo, train, eps, momentum);
            if (taps) { size_t b = sizeof(float) * (size_t)N * units[i].cout * ho * wo; taps[i] = (float *)malloc(b); memcpy(taps[i], nxt, b); }
            int is_skip = (cur == skip[0] || cur == skip[1] || cur == skip[2]);
            if (!is_skip) free(cur);
            cur = nxt; h = ho; w = wo;
        }
        const int c = units[28 + 2 * (j - 1)].cout;
        float *up = (float *)malloc(sizeof(float) * (size_t)N * c * 4 * h * w);
        fdo_upsample_nearest2x(cur, up, (long)N * c, h, w);
        free(cur); cur = up; h *= 2; w *= 2;
        const int s = (j == 4) ? 0 : (j == 3) ? 1 : (j == 2) ? 2 : -1;
        if (s >= 0) {
            if (skip_c[s] != c || skip_h[s] != h) return -1;
            fdo_add_inplace(cur, skip[s], (long)N * c * h * w);
        }
    }
    float *out = run_unit(&units[37], cur, N, h, w, &ho, &wo, train, eps, momentum);
    if (taps) { size_t b = sizeof(float) * (size_t)N * units[37].cout * ho * wo; taps[37] = (float *)malloc(b); memcpy(taps[37], out, b); }
    memcpy(y, out, sizeof(float) * (size_t)N * units[37].cout * ho * wo);
    free(out); free(cur);
    for (int s = 0; s < 3; ++s) free(skip[s]);
    return 0;
}

void fdo_free(void *p) { free(p); }
