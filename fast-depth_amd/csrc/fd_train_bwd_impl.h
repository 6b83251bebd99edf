// fd_train_bwd_impl.h -- host side of fd_train_backward, fd_l1_loss, fd_sgd_step and the library-issued gradient exchange (translation unit fd_train_bwd.hip)
#pragma once
#include "fd_train_plan.h"

namespace {

struct BwdCtx {
    fd_train_plan *p;
    const fd_layer_params *params;
    const fd_layer_grads *grads;
    hipStream_t s;
    // weight-gradient partials of the units processed so far in this range: reduced by ONE launch at the end of the range
    // (fd_reduce_weights_batch_f32)
    fd_wbatch wb{};
};

int flush_weights(BwdCtx &c)
{
    if (!c.wb.count) return FD_OK;
    FD_LAUNCH(fd_reduce_weights_batch_f32, dim3((unsigned)c.wb.cb_start[c.wb.count]), dim3(1024), 0, c.s, c.wb);
    c.wb.count = 0;
    return check_launch("fd_reduce_weights_batch_f32");
}
int defer_weights(BwdCtx &c, const float *part, int nrows, int n, int KK, int C, float *out)
{
    if (n % 4) return fail(FD_ERR_INVALID, "weight tensor of %d elements: the batched reduction needs a multiple of 4", n);
    if (c.wb.count == FD_WBATCH_MAX) { int rc = flush_weights(c); if (rc) return rc; }
    if (c.wb.count == 0) c.wb.cb_start[0] = 0;
    c.wb.e[c.wb.count] = fd_wred_args{part, out, nrows, n, KK, C};
    c.wb.cb_start[c.wb.count + 1] = c.wb.cb_start[c.wb.count] + ceil_div(n, nrows <= FD_WBATCH_FEW_ROWS ? 4096 : 64);
    ++c.wb.count;
    return FD_OK;
}

// input rows per backward-data tile: balanced over the map (14 -> 7 + 7 instead of 8 + 6: both tiles full, smaller patches, one more workgroup per
// CU); the upsampled modes produce 2 x 2 blocks per low-resolution pixel and need an even count
inline int dw_dgrad_rows(const fd_train_plan *p, const TLayer &L)
{
    if (p->tune & FD_TUNE_DW_TH8) return 8;
    // 3x3 stride-1 units: 14 rows where they divide the map (same reasoning as the forward kernel's larger tiles: fewer, fatter workgroups; the
    // dz patch of 16 x 18 pixels = 41.5 KB stays below the 44.4 KB the paired weight-gradient role needs anyway)
    // (measured, bf16 step: conv1 55.3 -> 52.1 us, conv3 59.2 -> 55.1, conv5 35.3 -> 32.7, 14x14 maps 20.5 -> 19.5)
    if (L.d.ksize == 3 && L.d.stride == 1 && L.mode == 0 && L.in_h % 14 == 0) return 14;
    const int th = ceil_div(L.in_h, ceil_div(L.in_h, L.d.ksize == 5 ? FD_T_DW5_DTH : (L.d.stride == 2 ? FD_T_S2_DTH : 8)));
    return (L.mode != 0 || L.d.stride == 2) ? (th + 1) / 2 * 2 : th;     // (stride 2: the tile must hold whole receptive-field rows of its owned outputs)
}
inline int dw_dgrad_cols(const TLayer &L) { return L.d.ksize == 5 ? FD_T_DW5_DTW : (L.d.stride == 2 ? FD_T_S2_DTW : 16); }
// Rows (columns) of the dz patch that an INPUT-space tile of t rows (columns, a multiple of the stride, starting on a multiple of it) reads: the
// EXACT extent the kernel computes (fd_dw_dgrad_body: PH, PW) -- stride 1: t + K - 1; stride 2: t / 2 + 2.  (Rounds 1-2 requested up to 4 rows and
// columns more: 58 KB instead of 38 for the 5x5 units = 2 resident workgroups per CU instead of 4.)
inline int dw_dz_patch(int t, int k, int s) { return (t + k - 2) / s + (s == 2 ? 2 : 1); }
inline size_t dw_bwd_lds(int ph, int pw, int cb, int k, int pstr, int le) { return lds_patch_bytes((long)ph * pw, pstr, le) + (size_t)k * k * cb * 4; }

// (bwd_rows(plan, u, nblk), fd_train_plan.h: the statistics rows of unit u as the backward-data kernel of its consumer -- nblk workgroups per channel -- sees them)
inline fd_bn_bwd_fin bwd_fin_args(BwdCtx &c, int i, size_t cf_off)
{
    TLayer &L = c.p->layers[i];
    return fd_bn_bwd_fin{stat_ptr(c.p, L.sb_off), L.nr_b, stat_pitch(L.d.cout), (int)cf_off, 1.0 / L.n_stat, tws(c.p, L.st_off), c.grads[i].bn_weight, c.grads[i].bn_bias, tws(c.p, L.coef_off)};
}
int bn_bwd_finalize(BwdCtx &c, int i)
{
    TLayer &L = c.p->layers[i];
    const fd_bn_bwd_fin fa = bwd_fin_args(c, i, 0);
    FD_LAUNCH(fd_bn_bwd_finalize_rows_f32, dim3((unsigned)ceil_div(L.d.cout, 16)), dim3(256), 0, c.s, fa, L.d.cout);
    return check_launch("fd_bn_bwd_finalize_rows_f32");
}
// unit u's statistics rows are complete (its consumer's backward kernels have been launched): a capable first kernel of u finalises them itself
// (TLayer::bwd_fin_rows, consumed when unit u is processed -- possibly by a later range call); otherwise the separate launch
int finalize_or_defer(BwdCtx &c, int u)
{
    TLayer &U = c.p->layers[u];
    if (U.bwd_fin && U.nr_b <= (dw_bwd_row_kernel(c.p, u) ? FD_STAT_FIN_MAX_ROWS_ROWK : FD_STAT_FIN_MAX_ROWS_BLOCK)) { U.bwd_fin_rows = U.nr_b; return FD_OK; }     // (more rows: re-reading them in every workgroup costs more than the launch)
    U.bwd_fin_rows = 0;
    return bn_bwd_finalize(c, u);
}

template <typename T, int K, int S, int MODE, int ACT_IN, int ADD_SG>
int launch_dw_dgrad(BwdCtx &c, int i, int *nblk_out)
{
    TLayer &L = c.p->layers[i];
    TLayer &P = c.p->layers[L.d.src];
    const int cb = L.dw_n << L.cbq, le = L.dw_n == 8 ? 2 : 4;
    L.lds_rounding = (L.lds_rounding & ~2) | (L.dw_n == 8 ? 2 : 0);
    L.bwd_rows = 0;
    const int TH = dw_dgrad_rows(c.p, L), TW = dw_dgrad_cols(L);
    const int tiles_x = ceil_div(L.in_w, TW), tiles_y = ceil_div(L.in_h, TH);
    const int ph = dw_dz_patch(TH, K, S), pw = dw_dz_patch(TW, K, S);
    const size_t lds = dw_bwd_lds(ph, pw, cb, K, L.bpstr, le);
    dim3 grid(tiles_x * tiles_y, ceil_div(L.d.cin, cb), c.p->B);
    const TLayer *Kp = L.d.skip >= 0 ? &c.p->layers[L.d.skip] : nullptr;
    fd_by_lane_width<T>(L.dw_n, [&](auto nt) {
        constexpr int NL = decltype(nt)::value;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)fd_dw_dgrad<T, K, S, MODE, ACT_IN, ADD_SG, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        FD_LAUNCH((fd_dw_dgrad<T, K, S, MODE, ACT_IN, ADD_SG, NL>), grid, dim3(256), lds, c.s, twt<T>(c.p, L.g_off), twt<T>(c.p, L.z_off), tws(c.p, L.coef_off),
                  c.params[i].conv_weight, twt<T>(c.p, P.z_off), tws(c.p, P.st_off), ADD_SG ? twt<T>(c.p, P.sg_off) : (const T *)nullptr,
                  twt<T>(c.p, P.g_off), Kp ? twt<T>(c.p, Kp->sg_off) : (T *)nullptr, bwd_rows(c.p, L.d.src, (long)tiles_x * tiles_y * c.p->B),
                  L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.cbq, TH, TW, tiles_x, L.csplit, L.bpstr);
    });
    *nblk_out = tiles_x * tiles_y * c.p->B;
    return check_launch("fd_dw_dgrad");
}

template <typename T, int ACT_IN, int ADD_SG>
int dispatch_dw_dgrad(BwdCtx &c, int i, int *nblk)
{
    const TLayer &L = c.p->layers[i];
    const int key = L.d.ksize * 100 + L.d.stride * 10 + L.mode;
    switch (key) {
    case 310: return launch_dw_dgrad<T, 3, 1, 0, ACT_IN, ADD_SG>(c, i, nblk);
    case 320: return launch_dw_dgrad<T, 3, 2, 0, ACT_IN, ADD_SG>(c, i, nblk);
    case 510: return launch_dw_dgrad<T, 5, 1, 0, ACT_IN, ADD_SG>(c, i, nblk);
    case 511: return launch_dw_dgrad<T, 5, 1, 1, ACT_IN, ADD_SG>(c, i, nblk);
    case 512: return launch_dw_dgrad<T, 5, 1, 2, ACT_IN, ADD_SG>(c, i, nblk);
    case 513: return launch_dw_dgrad<T, 5, 1, 3, ACT_IN, ADD_SG>(c, i, nblk);
    }
    return fail(FD_ERR_INVALID, "train: depthwise backward k=%d stride=%d mode=%d has no kernel", L.d.ksize, L.d.stride, L.mode);
}

template <typename T, int ACT1, int ACT2>
int launch_dw_wgrad_acts(BwdCtx &c, int i)
{
    TLayer &L = c.p->layers[i];
    TLayer &P = c.p->layers[L.d.src];
    const TLayer *Kp = L.d.skip >= 0 ? &c.p->layers[L.d.skip] : nullptr;
    const int key = L.d.ksize * 100 + L.d.stride * 10 + L.mode;
    float *wpart = tws(c.p, L.wp_off);
    // tiles per workgroup (along x): as many as keep >= ~1536 workgroups in flight
    const int ncb_w = ceil_div(L.d.cin, L.dw_n << L.cbq);
    const int btx = ceil_div(L.out_w, L.btw), bty = ceil_div(L.out_h, L.bth);     // the backward-weights kernel's own output tiles (the forward's may be larger)
    int tpw = std::max(1, std::min(btx, (int)((long)btx * bty * ncb_w * c.p->B / FD_DW_WGRAD_TARGET_WGS)));
    if (c.p->tune & FD_TUNE_WGRAD_TILE_ROWS) tpw = btx;
    const int groups_x = ceil_div(btx, tpw);
    tpw = ceil_div(btx, groups_x);
    const dim3 wgrid(groups_x * bty, ncb_w, c.p->B);
    // LDS: activated input patch + dz tile, both [pixels][cb + 4] floats; the final reduction (npt/K groups x K*K taps x cb) reuses it
    const int cbw = L.dw_n << L.cbq, le = L.dw_n == 8 ? 2 : 4, th_in = (L.bth - 1) * L.d.stride + L.d.ksize, tw_in = (L.btw - 1) * L.d.stride + L.d.ksize;
    const size_t wlds = std::max(lds_patch_bytes((long)th_in * tw_in + L.bth * L.btw, L.bpstr, le), (size_t)((256 >> L.cbq) / L.d.ksize) * L.d.ksize * L.d.ksize * cbw * 4);
    if (wlds > 64 * 1024) return fail(FD_ERR_INVALID, "depthwise wgrad: LDS request %zu exceeds 64 KiB", wlds);
    const int wblk = groups_x * bty * c.p->B;
#define FD_DWW(K_, S_, M_)                                                                                                         \
    case K_ * 100 + S_ * 10 + M_:                                                                                                  \
        fd_by_lane_width<T>(L.dw_n, [&](auto nt) {                                                                                \
            constexpr int NL = decltype(nt)::value;                                                                               \
            FD_LAUNCH((fd_dw_wgrad<T, K_, S_, M_, ACT1, ACT2, NL>), wgrid, dim3(256), wlds, c.s, twt<T>(c.p, P.z_off), tws(c.p, P.st_off),  \
                      Kp ? twt<T>(c.p, Kp->z_off) : (const T *)nullptr, Kp ? tws(c.p, Kp->st_off) : (const float *)nullptr,           \
                      twt<T>(c.p, L.g_off), twt<T>(c.p, L.z_off), tws(c.p, L.coef_off), wpart, L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin,     \
                      L.cbq, L.bth, L.btw, btx, tpw, L.csplit, L.bpstr);                                                          \
        });                                                                                                                        \
        break;
    switch (key) {
        FD_DWW(3, 1, 0) FD_DWW(3, 2, 0) FD_DWW(5, 1, 0) FD_DWW(5, 1, 1) FD_DWW(5, 1, 2) FD_DWW(5, 1, 3)
    default: return fail(FD_ERR_INVALID, "train: depthwise wgrad has no kernel for this layer");
    }
#undef FD_DWW
    int rc = check_launch("fd_dw_wgrad");
    if (rc) return rc;
    const int kk = L.d.ksize * L.d.ksize;
    if ((size_t)wblk * kk * L.d.cin > L.wp_elems) return fail(FD_ERR_STATE, "depthwise weight-gradient partial region too small");
    return defer_weights(c, wpart, wblk, kk * L.d.cin, kk, L.d.cin, c.grads[i].conv_weight);
}

template <typename T>
int launch_dw_wgrad(BwdCtx &c, int i)
{
    const TLayer &L = c.p->layers[i];
    const int a1 = c.p->layers[L.d.src].d.act, a2 = L.d.skip >= 0 ? c.p->layers[L.d.skip].d.act : FD_ACT_RELU6;
    if (a1 == FD_ACT_RELU6 && a2 == FD_ACT_RELU6) return launch_dw_wgrad_acts<T, FD_ACT_RELU6_, FD_ACT_RELU6_>(c, i);
    if (a1 == FD_ACT_RELU && a2 == FD_ACT_RELU6) return launch_dw_wgrad_acts<T, FD_ACT_RELU_, FD_ACT_RELU6_>(c, i);
    if (a1 == FD_ACT_RELU && a2 == FD_ACT_RELU) return launch_dw_wgrad_acts<T, FD_ACT_RELU_, FD_ACT_RELU_>(c, i);
    return launch_dw_wgrad_acts<T, FD_ACT_RELU6_, FD_ACT_RELU_>(c, i);
}

// Both backward kernels of a depthwise unit in ONE launch (fd_dw_bwd); returns FD_OK with *paired = false when this unit's combination of
// kernel size / stride / input composition / activations has no paired instance (the caller then launches the two kernels one after the other).
#ifndef FD_DW3_ROWS_SMALL_BAND
#define FD_DW3_ROWS_SMALL_BAND 7
#endif
template <typename T, int K, int S, int MODE, int ACT1, int ACT2, int ADD_SG>
int launch_dw_bwd_pair(BwdCtx &c, int i, int *nblk_out)
{
    TLayer &L = c.p->layers[i];
    TLayer &P = c.p->layers[L.d.src];
    const TLayer *Kp = L.d.skip >= 0 ? &c.p->layers[L.d.skip] : nullptr;
    const int cb = L.dw_n << L.cbq, le = L.dw_n == 8 ? 2 : 4;
    L.lds_rounding = (L.lds_rounding & ~2) | (L.dw_n == 8 ? 2 : 0);
    fd_dw_bwd_args<T> a{};
    a.G = twt<T>(c.p, L.g_off); a.Z = twt<T>(c.p, L.z_off); a.Zin = twt<T>(c.p, P.z_off);
    a.Zskip = Kp ? twt<T>(c.p, Kp->z_off) : nullptr; a.SG = ADD_SG ? twt<T>(c.p, P.sg_off) : nullptr;
    a.Gin = twt<T>(c.p, P.g_off); a.SGout = Kp ? twt<T>(c.p, Kp->sg_off) : nullptr;
    a.coef = tws(c.p, L.coef_off); a.w = c.params[i].conv_weight; a.st_in = tws(c.p, P.st_off); a.st_skip = Kp ? tws(c.p, Kp->st_off) : nullptr;
    a.wpart = tws(c.p, L.wp_off);
    a.Hin = L.in_h; a.Win = L.in_w; a.Ho = L.out_h; a.Wo = L.out_w; a.C = L.d.cin; a.cbq = L.cbq; a.csplit = L.csplit; a.pstr = L.bpstr; a.B = c.p->B;
    // backward-data geometry (launch_dw_dgrad)
    a.d_th = dw_dgrad_rows(c.p, L); a.d_tw = dw_dgrad_cols(L);
    a.d_tiles_x = ceil_div(L.in_w, a.d_tw);
    a.d_gx = a.d_tiles_x * ceil_div(L.in_h, a.d_th); a.d_gy = ceil_div(L.d.cin, cb);
    const int ph = dw_dz_patch(a.d_th, K, S), pw = dw_dz_patch(a.d_tw, K, S);
    const size_t lds_d = dw_bwd_lds(ph, pw, cb, K, L.bpstr, le);
    // backward-weights geometry (launch_dw_wgrad_acts); the pair keeps roughly the same number of workgroups in flight per role
    const int btx = ceil_div(L.out_w, L.btw), bty = ceil_div(L.out_h, L.bth);
    int tpw = std::max(1, std::min(btx, (int)((long)btx * bty * ceil_div(L.d.cin, cb) * c.p->B / FD_DW_WGRAD_TARGET_WGS)));
    if (c.p->tune & FD_TUNE_WGRAD_TILE_ROWS) tpw = btx;
    const int groups_x = ceil_div(btx, tpw);
    tpw = ceil_div(btx, groups_x);
    a.w_th = L.bth; a.w_tw = L.btw; a.w_tiles_x = btx; a.w_tpw = tpw; a.w_gx = groups_x * bty; a.w_gy = ceil_div(L.d.cin, cb);
    const int th_in = (L.bth - 1) * S + K, tw_in = (L.btw - 1) * S + K;
    const size_t lds_w = std::max(lds_patch_bytes((long)th_in * tw_in + L.bth * L.btw, L.bpstr, le), (size_t)((256 >> L.cbq) / K) * K * K * cb * 4);
    size_t lds = align_up(std::max(lds_d, lds_w), 16);
    if (L.bwd_fin_rows) lds += (size_t)4 * cb * 4;          // + the coefficient block of the in-kernel finalisation (fd_bn_bwd_fin::cf_off)
    if (lds > 160 * 1024) return fail(FD_ERR_INVALID, "depthwise backward pair: LDS request %zu exceeds 160 KiB", lds);
    const int kk = K * K;
    // 16-bit plans, 5x5 on up2 + skip (decode_conv3 / 4 / 5 .0): both gradients on the row-walking pixel-pair kernel (fd_kernels_dw5p_bwd.h) -- one launch,
    // backward-data workgroups first, then the weight-gradient workgroups of the same image (same XCD: the second role finds G / z in its L2)
    if constexpr (K == 5 && S == 1 && MODE == 2 && ADD_SG == 0 && !std::is_same<T, float>::value) {
        if (L.in_w % 4 == 0 && L.in_h % 2 == 0 && L.d.cin % 8 == 0 && (double)L.in_h * L.in_w * L.d.cin * 2.0 < 2147483648.0 &&
            !(c.p->tune & (FD_TUNE_NO_DW5_ROWS | FD_TUNE_DW_BWD1 | FD_TUNE_DW_BWD_PAIR))) {
            fd_dw5_bwd_args<T> b{};
            b.G = a.G; b.Z = a.Z; b.Zin = a.Zin; b.Zskip = a.Zskip; b.Gin = a.Gin; b.SGout = a.SGout;
            b.coef = a.coef; b.w = a.w; b.st_in = a.st_in; b.st_skip = a.st_skip; b.wpart = a.wpart;
            b.H = L.in_h; b.W = L.in_w; b.C = L.d.cin; b.groups_x = ceil_div(L.in_w, 8);
            const int bands = std::max(1, (L.in_h + 7) / 14);
            b.bh_d = b.bh_w = ceil_div(ceil_div(L.in_h, bands), 2) * 2;
            b.wgs_d = b.wgs_w = ceil_div((long)b.groups_x * ceil_div(L.in_h, b.bh_d), 4);
            b.sr = bwd_rows(c.p, L.d.src, (long)b.wgs_d * c.p->B);
            if (L.bwd_fin_rows) b.fin = bwd_fin_args(c, i, 0);
            const int wrows = b.wgs_w * c.p->B;
            if ((size_t)wrows * kk * L.d.cin > L.wp_elems) return fail(FD_ERR_STATE, "depthwise weight-gradient partial region too small");
            L.lds_rounding = (L.lds_rounding & ~(2 | 8)) | 2 | 8;       // dz and the re-created input rounded to the storage type; the backward-data taps too
            L.bwd_rows = 1;
            FD_LAUNCH((fd_dw5_bwd_rows<T, ACT1, ACT2>), dim3((unsigned)(b.wgs_d + b.wgs_w), (unsigned)ceil_div(L.d.cin, 64), (unsigned)c.p->B), dim3(256), 0, c.s, b);
            int rc5 = check_launch("fd_dw5_bwd_rows");
            if (rc5) return rc5;
            *nblk_out = b.wgs_d * c.p->B;
            return defer_weights(c, b.wpart, wrows, kk * L.d.cin, kk, L.d.cin, c.grads[i].conv_weight);
        }
    }
    L.lds_rounding &= ~8;
    // 16-bit plans, 3x3 stride 1 on plain inputs (conv1.0 / conv3.0 / conv5.0 / the 14x14 units ...): both gradients on the row-walking fp32-window kernel
    // (fd_kernels_dw5p_bwd.h: fd_dw3_bwd_rows); nothing is rounded there, so the unit reports no LDS rounding
    if constexpr (K == 3 && S == 1 && MODE == 0 && ADD_SG == 0 && !std::is_same<T, float>::value) {
        if (L.d.cin % 8 == 0 && (double)L.in_h * L.in_w * L.d.cin * 2.0 < 2147483648.0 && (long)L.in_h * L.in_w >= FD_DW3_ROWS_MIN_PIXELS &&
            !(c.p->tune & (FD_TUNE_NO_DW5_ROWS | FD_TUNE_DW_BWD1 | FD_TUNE_DW_BWD_PAIR | FD_TUNE_FORCE_DW_H8))) {
            fd_dw3_bwd_args<T> b{};
            b.G = a.G; b.Z = a.Z; b.Zin = a.Zin; b.Gin = a.Gin; b.coef = a.coef; b.w = a.w; b.st_in = a.st_in; b.wpart = a.wpart;
            const int cl = L.d.cin <= 32 ? 16 : 32;          // channel lanes per strip: a 32-channel unit (conv1.0) would leave half of every wave idle at 32
            b.H = L.in_h; b.W = L.in_w; b.C = L.d.cin; b.groups_x = ceil_div(L.in_w, 4 * (64 / cl));
            // bands of ~14 rows on the large maps; the 14x14 / 7x7 maps take bands of 7 (twice the waves: their launches are latency-, not issue-bound)
            const int bands = L.in_h <= 7 ? ceil_div(L.in_h, 4) : L.in_h <= 14 ? ceil_div(L.in_h, FD_DW3_ROWS_SMALL_BAND) : std::max(1, (L.in_h + 7) / 14);     // (7x7: 512 -> 1024 waves, -2.7 us)
            b.bh_d = b.bh_w = ceil_div(L.in_h, bands);
            b.wgs_d = b.wgs_w = ceil_div((long)b.groups_x * ceil_div(L.in_h, b.bh_d), 4);
            b.sr = bwd_rows(c.p, L.d.src, (long)b.wgs_d * c.p->B);
            if (L.bwd_fin_rows) b.fin = bwd_fin_args(c, i, 0);
            const int wrows = b.wgs_w * c.p->B;
            if ((size_t)wrows * kk * L.d.cin > L.wp_elems) return fail(FD_ERR_STATE, "depthwise weight-gradient partial region too small");
            L.lds_rounding &= ~2;
            L.bwd_rows = 1;
            const dim3 g3((unsigned)(b.wgs_d + b.wgs_w), (unsigned)ceil_div(L.d.cin, 2 * cl), (unsigned)c.p->B);
            if (cl == 16) FD_LAUNCH((fd_dw3_bwd_rows<T, ACT1, 16>), g3, dim3(256), 0, c.s, b);
            else FD_LAUNCH((fd_dw3_bwd_rows<T, ACT1, 32>), g3, dim3(256), 0, c.s, b);
            int rc3 = check_launch("fd_dw3_bwd_rows");
            if (rc3) return rc3;
            *nblk_out = b.wgs_d * c.p->B;
            return defer_weights(c, b.wpart, wrows, kk * L.d.cin, kk, L.d.cin, c.grads[i].conv_weight);
        }
    }
    // 16-bit plans, 3x3 stride 2 (conv2.0 / conv4.0 / conv6.0 / conv12.0): ONE row-walking kernel produces both gradients from one pass over z_in, G, z and the
    // skip gradient (fd_kernels_dw5p_bwd.h: fd_dw3s2_bwd_rows); these units are byte-bound and the paired forms read their operands twice
    if constexpr (K == 3 && S == 2 && MODE == 0 && !std::is_same<T, float>::value) {
        if (L.d.cin % 8 == 0 && (double)L.in_h * L.in_w * L.d.cin * 2.0 < 2147483648.0 &&
            !(c.p->tune & (FD_TUNE_NO_DW5_ROWS | FD_TUNE_DW_BWD1 | FD_TUNE_DW_BWD_PAIR | FD_TUNE_FORCE_DW_H8 | FD_TUNE_DW_FORCE_ROWS | FD_TUNE_DW_NO_ROWS))) {
            fd_dw3s2_bwd_args<T> b{};
            b.G = a.G; b.Z = a.Z; b.Zin = a.Zin; b.SG = a.SG; b.Gin = a.Gin; b.coef = a.coef; b.w = a.w; b.st_in = a.st_in; b.wpart = a.wpart;
            b.Ho = L.out_h; b.Wo = L.out_w; b.C = L.d.cin; b.groups_x = ceil_div(L.out_w, 8);
            const int bands = L.out_h <= 7 ? ceil_div(L.out_h, 2) : L.out_h <= 14 ? ceil_div(L.out_h, 4) : std::max(1, (L.out_h + 3) / 7);     // ~7 output rows (14 input rows) per band; the small maps take 4 / 2
            b.bh = ceil_div(L.out_h, bands);
            b.wgs = ceil_div((long)b.groups_x * ceil_div(L.out_h, b.bh), 4);
            b.sr = bwd_rows(c.p, L.d.src, (long)b.wgs * c.p->B);
            if (L.bwd_fin_rows) b.fin = bwd_fin_args(c, i, 0);
            const int wrows = b.wgs * c.p->B;
            if ((size_t)wrows * kk * L.d.cin > L.wp_elems) return fail(FD_ERR_STATE, "depthwise weight-gradient partial region too small");
            L.lds_rounding &= ~2;
            L.bwd_rows = 1;
            FD_LAUNCH((fd_dw3s2_bwd_rows<T, ACT1, ADD_SG>), dim3((unsigned)b.wgs, (unsigned)ceil_div(L.d.cin, 64), (unsigned)c.p->B), dim3(256), 0, c.s, b);
            int rc2 = check_launch("fd_dw3s2_bwd_rows");
            if (rc2) return rc2;
            *nblk_out = b.wgs * c.p->B;
            return defer_weights(c, b.wpart, wrows, kk * L.d.cin, kk, L.d.cin, c.grads[i].conv_weight);
        }
    }
    // The stride-2 3x3 units of the large maps (channel-group count a power of two in 8 ... 64): two register-window kernels without LDS staging
    // (fd_dw3s2_dgrad_rows over input columns, fd_dw3_wgrad_rows over output columns), row strips as high as still leave >= ~1024 workgroups
    {
        const int cgn = L.d.cin / 4;
        // measured (us, rows pair vs single-staging kernel): bf16 conv2.0 48.9 + 16.8 vs 79.3, conv4.0 33.4 + 11.0 vs 48.2, conv6.0 (14x14 outputs) 23.4 + 7.8 vs 29.6;
        // fp32 conv2.0 69.0 + 26.7 vs 96.2, conv4.0 40.7 + 15.3 vs 55.2 (equal: both forms move the fp32 bytes at the same rate) -> 16-bit plans, maps >= 28x28
        if (dw_bwd_on_rows(c.p, L)) {
            L.lds_rounding &= ~2;
            const int gxd = ceil_div((long)L.in_w * cgn, 256), h2 = L.in_h / 2;
            int th2 = h2;
            while (th2 > 2 && (long)gxd * ceil_div(h2, th2) * c.p->B < 1024) th2 = (th2 + 1) / 2;
            const int gyd = ceil_div(h2, th2);
            a.sr = bwd_rows(c.p, L.d.src, (long)gxd * gyd * c.p->B);
            FD_LAUNCH((fd_dw3s2_dgrad_rows<T, ACT1, ADD_SG>), dim3(gxd, gyd, c.p->B), dim3(256), 0, c.s, a.G, a.Z, a.coef, a.w, a.Zin, a.st_in, a.SG, a.Gin, a.sr,
                      L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, th2);
            int rcr = check_launch("fd_dw3s2_dgrad_rows");
            if (rcr) return rcr;
            const int gxw = ceil_div((long)L.out_w * cgn, 256);
            int thw = L.out_h;
            while (thw > 4 && (long)gxw * ceil_div(L.out_h, thw) * c.p->B < 1024) thw = (thw + 1) / 2;
            const int gyw = ceil_div(L.out_h, thw), wrows = gxw * gyw * c.p->B;
            if ((size_t)wrows * kk * L.d.cin > L.wp_elems) return fail(FD_ERR_STATE, "depthwise weight-gradient partial region too small");
            FD_LAUNCH((fd_dw3_wgrad_rows<T, 2, ACT1>), dim3(gxw, gyw, c.p->B), dim3(256), 0, c.s, a.Zin, a.st_in, a.G, a.Z, a.coef, a.wpart,
                      L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, thw);
            if ((rcr = check_launch("fd_dw3_wgrad_rows"))) return rcr;
            *nblk_out = gxd * gyd * c.p->B;
            return defer_weights(c, a.wpart, wrows, kk * L.d.cin, kk, L.d.cin, c.grads[i].conv_weight);
        }
    }
    // (stride-1 3x3 units: the LDS-tiled backward-data kernel on its own + fd_dw3_wgrad_rows instead of the paired launch measured equal -- conv1.0
    // 31.2 + 20.3 vs 52.8 us, conv3.0 34.8 + 20.5 vs 56.4, conv5.0 22.1 + 14.6 vs 33.1 -- the pair stays)
    // measured (bf16, batch 32): the single-staging kernel wins on the stride-2 units (conv2.0 84 vs 103 us, conv4.0 50 vs 57, conv6.0 31 vs 35) and
    // loses on the stride-1 3x3 ones (conv1.0 68 vs 57, 14x14 maps 22.4 vs 19.5: two tap phases back to back in one workgroup at lower residency);
    // the 5x5 units tie.  FD_TUNE_DW_BWD1 forces it everywhere (tests), FD_TUNE_DW_BWD_PAIR nowhere.
    if (!(c.p->tune & FD_TUNE_DW_BWD_PAIR) && (S == 2 || (c.p->tune & FD_TUNE_DW_BWD1))) {
        // ONE workgroup per input-space tile stages the dz patch and the forward-input patch once and produces both gradients (fd_dw_bwd1)
        const int oth = a.d_th / S, otw = a.d_tw / S;
        const int th_in1 = (oth - 1) * S + K, tw_in1 = (otw - 1) * S + K;
        const int ph1 = (a.d_th + K - 2) / S + 2, pw1 = (a.d_tw + K - 2) / S + 2;          // upper bound of the dz patch
        size_t lds1 = align_up(std::max(lds_patch_bytes((long)ph1 * pw1 + th_in1 * tw_in1, L.bpstr, le) + (size_t)kk * cb * 4, (size_t)((256 >> L.cbq) / K) * kk * cb * 4 + 8192), 16);
        if (L.bwd_fin_rows) { a.fin = bwd_fin_args(c, i, lds1); lds1 += (size_t)4 * cb * 4; }
        if (lds1 > 160 * 1024) return fail(FD_ERR_INVALID, "depthwise backward: LDS request %zu exceeds 160 KiB", lds1);
        const int wblk1 = a.d_gx * c.p->B;
        if ((size_t)wblk1 * kk * L.d.cin > L.wp_elems) return fail(FD_ERR_STATE, "depthwise weight-gradient partial region too small");
        a.sr = bwd_rows(c.p, L.d.src, wblk1);
        fd_by_lane_width<T>(L.dw_n, [&](auto nt) {
            constexpr int NL = decltype(nt)::value;
            if (a.fin.rows) {                                 // (the instance with the in-kernel finalisation costs registers: only where it replaces a launch)
                (void)hipFuncSetAttribute((const void *)fd_dw_bwd1<T, K, S, MODE, ACT1, ACT2, ADD_SG, NL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
                FD_LAUNCH((fd_dw_bwd1<T, K, S, MODE, ACT1, ACT2, ADD_SG, NL, true>), dim3((unsigned)a.d_gx, (unsigned)a.d_gy, (unsigned)c.p->B), dim3(256), lds1, c.s, a);
            } else {
                (void)hipFuncSetAttribute((const void *)fd_dw_bwd1<T, K, S, MODE, ACT1, ACT2, ADD_SG, NL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
                FD_LAUNCH((fd_dw_bwd1<T, K, S, MODE, ACT1, ACT2, ADD_SG, NL, false>), dim3((unsigned)a.d_gx, (unsigned)a.d_gy, (unsigned)c.p->B), dim3(256), lds1, c.s, a);
            }
        });
        int rc1 = check_launch("fd_dw_bwd1");
        if (rc1) return rc1;
        *nblk_out = a.d_gx * c.p->B;
        return defer_weights(c, a.wpart, wblk1, kk * L.d.cin, kk, L.d.cin, c.grads[i].conv_weight);
    }
    const int wblk = a.w_gx * c.p->B;
    if ((size_t)wblk * kk * L.d.cin > L.wp_elems) return fail(FD_ERR_STATE, "depthwise weight-gradient partial region too small");
    if (L.bwd_fin_rows) a.fin = bwd_fin_args(c, i, lds - (size_t)4 * cb * 4);
    a.sr = bwd_rows(c.p, L.d.src, (long)a.d_gx * c.p->B);
    const long total = (long)c.p->B * ((long)a.d_gx * a.d_gy + (long)a.w_gx * a.w_gy);
    fd_by_lane_width<T>(L.dw_n, [&](auto nt) {
        constexpr int NL = decltype(nt)::value;
        if (a.fin.rows) {
            if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)fd_dw_bwd<T, K, S, MODE, ACT1, ACT2, ADD_SG, NL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            FD_LAUNCH((fd_dw_bwd<T, K, S, MODE, ACT1, ACT2, ADD_SG, NL, true>), dim3((unsigned)total), dim3(256), lds, c.s, a);
        } else {
            if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)fd_dw_bwd<T, K, S, MODE, ACT1, ACT2, ADD_SG, NL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            FD_LAUNCH((fd_dw_bwd<T, K, S, MODE, ACT1, ACT2, ADD_SG, NL, false>), dim3((unsigned)total), dim3(256), lds, c.s, a);
        }
    });
    int rc = check_launch("fd_dw_bwd");
    if (rc) return rc;
    *nblk_out = a.d_gx * c.p->B;
    return defer_weights(c, a.wpart, wblk, kk * L.d.cin, kk, L.d.cin, c.grads[i].conv_weight);
}

template <typename T>
int dispatch_dw_bwd_pair(BwdCtx &c, int i, int *nblk, bool *paired)
{
    const TLayer &L = c.p->layers[i];
    const TLayer &P = c.p->layers[L.d.src];
    const int a1 = P.d.act, a2 = L.d.skip >= 0 ? c.p->layers[L.d.skip].d.act : FD_ACT_RELU6;
    const bool add = P.skip_consumer >= 0 && L.mode == 0;
    const int key = L.d.ksize * 100 + L.d.stride * 10 + L.mode;
    *paired = true;
    if (a1 == FD_ACT_RELU6 && !add) {
        if (key == 310) return launch_dw_bwd_pair<T, 3, 1, 0, FD_ACT_RELU6_, FD_ACT_RELU6_, 0>(c, i, nblk);
        if (key == 320) return launch_dw_bwd_pair<T, 3, 2, 0, FD_ACT_RELU6_, FD_ACT_RELU6_, 0>(c, i, nblk);
        if (key == 510) return launch_dw_bwd_pair<T, 5, 1, 0, FD_ACT_RELU6_, FD_ACT_RELU6_, 0>(c, i, nblk);
    }
    if (a1 == FD_ACT_RELU6 && add && key == 320) return launch_dw_bwd_pair<T, 3, 2, 0, FD_ACT_RELU6_, FD_ACT_RELU6_, 1>(c, i, nblk);
    if (a1 == FD_ACT_RELU && key == 511) return launch_dw_bwd_pair<T, 5, 1, 1, FD_ACT_RELU_, FD_ACT_RELU6_, 0>(c, i, nblk);
    if (a1 == FD_ACT_RELU && a2 == FD_ACT_RELU6 && key == 512) return launch_dw_bwd_pair<T, 5, 1, 2, FD_ACT_RELU_, FD_ACT_RELU6_, 0>(c, i, nblk);
    if (a1 == FD_ACT_RELU && a2 == FD_ACT_RELU6 && key == 513) return launch_dw_bwd_pair<T, 5, 1, 3, FD_ACT_RELU_, FD_ACT_RELU6_, 0>(c, i, nblk);
    *paired = false;                                          // an unusual combination (sibling / custom plans): the two separate kernels cover it
    return FD_OK;
}

template <typename T, int ACT_IN>
int launch_pw_bwd_h16(BwdCtx &c, int i, int *nblk)
{
    TLayer &L = c.p->layers[i];
    TLayer &P = c.p->layers[L.d.src];
    const int M = (int)L.M, N = L.d.cout, K = L.d.cin;
    const T *Gsrc = twt<T>(c.p, L.g_off);
    T *G = L.dz_off ? twt<T>(c.p, L.dz_off) : twt<T>(c.p, L.g_off);     // dz: the operand of both GEMMs below
    int rc;
    // dz = BatchNorm-backward(G, z), in place over G: the operand of both GEMMs below
    {
        const long chunks = (long)M * N / 8;
        if (L.bwd_fin_rows) {                                 // the apply pass finalises the unit's BatchNorm backward itself
            const int gx = ceil_div(N, 64), gy = std::max(1, std::min(ceil_div(M, 32), 1024 / gx));
            const fd_bn_bwd_fin fa = bwd_fin_args(c, i, 0);
            FD_LAUNCH((fd_bn_bwd_apply_fin_h16<T>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, c.s, Gsrc, G, twt<T>(c.p, L.z_off), M, N, fa);
            if ((rc = check_launch("fd_bn_bwd_apply_fin_h16"))) return rc;
        } else {
            FD_LAUNCH((fd_bn_bwd_apply_h16<T>), dim3((unsigned)std::min<long>(4096, ceil_div(chunks, 256))), dim3(256), 0, c.s, Gsrc, G, twt<T>(c.p, L.z_off), tws(c.p, L.coef_off), chunks, N);
            if ((rc = check_launch("fd_bn_bwd_apply_h16"))) return rc;
        }
    }
    // weight gradient: output tiles x pixel splits
    const int n_tiles = ceil_div(N, 64), k_tiles_w = ceil_div(K, 64);
    int splits = std::max(1, std::min(ceil_div(FD_WGRAD_TARGET_WGS_H16, (long)n_tiles * k_tiles_w), ceil_div(M, 256)));
    const int rows = ceil_div(ceil_div(M, splits), 64) * 64;
    splits = ceil_div(M, rows);
    if ((size_t)splits * N * K > L.wp_elems) return fail(FD_ERR_STATE, "weight-gradient partial region too small");
    // backward data: 64 x 128 tiles of G_in when there are >= 128 input channels and that still leaves >= 200 workgroups: every dz fragment feeds two MFMAs
    const bool pair = !(c.p->flags & FD_PLAN_NO_BWD_PAIRING) && !(c.p->tune & FD_TUNE_NO_PW_PAIRING);
    int tn = (K >= 128 && (long)ceil_div(M, 64) * ceil_div(K, 128) >= 200) ? 2 : 1;
    if (pair && !(c.p->tune & FD_TUNE_PW_PAIR_TN2)) tn = 1;   // paired launch: 64 x 64 backward-data tiles (49 KB of LDS per workgroup instead of 74: the weight-gradient workgroups share it) -- measured 554 vs 583 us per bf16 step
    const int m_tiles = ceil_div(M, 64), k_tiles = ceil_div(K, 64 * tn);
    const size_t lds_d = FD_PW_DGRAD_H16_RING(L.n64, tn) + (size_t)4 * 64 * tn * 4;     // ring of min(3, N tiles) stages (>= the epilogue's fp32 tiles) + statistics
    const bool add = P.skip_consumer >= 0;
    const unsigned n_dgrad = (unsigned)((m_tiles + 7) / 8 * 8 * k_tiles);
    *nblk = m_tiles;
    if (pair) {
        const size_t lds = std::max(lds_d, FD_PW_WGRAD_H16_LDS(1));
        const int tiles_w = n_tiles * k_tiles_w;
#define FD_PWBWD_H16(ADDV, TNV)                                                                                                                    \
    do {                                                                                                                                           \
        (void)hipFuncSetAttribute((const void *)fd_pw_bwd_h16<T, ACT_IN, ADDV, TNV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        FD_LAUNCH((fd_pw_bwd_h16<T, ACT_IN, ADDV, TNV>), dim3(n_dgrad + (unsigned)(tiles_w * splits)), dim3(256), lds, c.s, G, twt<T>(c.p, L.wtt_off), twt<T>(c.p, P.z_off), \
                  tws(c.p, P.st_off), ADDV ? twt<T>(c.p, P.sg_off) : (const T *)nullptr, twt<T>(c.p, P.g_off), bwd_rows(c.p, L.d.src, m_tiles), tws(c.p, L.wp_off), M, N, K, L.n64, \
                  m_tiles, k_tiles, (int)n_dgrad, k_tiles_w, tiles_w, rows, (FD_PW_BWD_W_FIRST && (unsigned)(tiles_w * splits) < n_dgrad) ? tiles_w * splits : 0); \
    } while (0)
        if (add) { if (tn == 2) FD_PWBWD_H16(1, 2); else FD_PWBWD_H16(1, 1); }
        else { if (tn == 2) FD_PWBWD_H16(0, 2); else FD_PWBWD_H16(0, 1); }
#undef FD_PWBWD_H16
        if ((rc = check_launch("fd_pw_bwd_h16"))) return rc;
        return defer_weights(c, tws(c.p, L.wp_off), splits, N * K, 0, 0, c.grads[i].conv_weight);
    }
    {
        // (two k tiles per workgroup -- fd_pw_wgrad_h16<.., 2>, the staged dz tile feeding twice the MFMAs -- measured slower: 22.3 vs 20.9 us on the
        // 512 x 512 units, 3 instead of 5 workgroups per CU and half as many of them)
        FD_LAUNCH((fd_pw_wgrad_h16<T, ACT_IN, 1>), dim3(n_tiles * k_tiles_w, splits), dim3(256), FD_PW_WGRAD_H16_LDS(1), c.s, G, twt<T>(c.p, P.z_off), tws(c.p, P.st_off),
                  tws(c.p, L.wp_off), M, N, K, k_tiles_w, rows);
        if ((rc = check_launch("fd_pw_wgrad_h16"))) return rc;
        if ((rc = defer_weights(c, tws(c.p, L.wp_off), splits, N * K, 0, 0, c.grads[i].conv_weight))) return rc;
    }
    {
        dim3 grid(n_dgrad);
#define FD_DGRAD_H16(ADDV, TNV)                                                                                                                    \
    do {                                                                                                                                           \
        (void)hipFuncSetAttribute((const void *)fd_pw_dgrad_h16<T, ACT_IN, ADDV, TNV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d);    \
        FD_LAUNCH((fd_pw_dgrad_h16<T, ACT_IN, ADDV, TNV>), grid, dim3(256), lds_d, c.s, G, twt<T>(c.p, L.wtt_off), twt<T>(c.p, P.z_off), tws(c.p, P.st_off), \
                  ADDV ? twt<T>(c.p, P.sg_off) : (const T *)nullptr, twt<T>(c.p, P.g_off), bwd_rows(c.p, L.d.src, m_tiles), M, N, K, L.n64, m_tiles, k_tiles);          \
    } while (0)
        if (add) { if (tn == 2) FD_DGRAD_H16(1, 2); else FD_DGRAD_H16(1, 1); }
        else { if (tn == 2) FD_DGRAD_H16(0, 2); else FD_DGRAD_H16(0, 1); }
#undef FD_DGRAD_H16
        return check_launch("fd_pw_dgrad_h16");
    }
}

template <int ACT_IN>
int launch_pw_bwd(BwdCtx &c, int i, int *nblk)
{
    TLayer &L = c.p->layers[i];
    TLayer &P = c.p->layers[L.d.src];
    const int M = (int)L.M, N = L.d.cout, K = L.d.cin;
    const float *G = tws(c.p, L.g_off), *Z = tws(c.p, L.z_off), *coef = tws(c.p, L.coef_off);
    // weights: dW[N][K], reduction over M split across workgroups;  data: G_src[M][K], 64 x 64 tiles (both kernels tile K by 64)
    const int n_tiles = ceil_div(N, 64), k_tiles = ceil_div(K, 64), m_tiles = ceil_div(M, 64);
    int splits = std::max(1, std::min(ceil_div(FD_WGRAD_TARGET_WGS_F32, (long)n_tiles * k_tiles), ceil_div(M, 256)));
    const int rows = ceil_div(ceil_div(M, splits), 64) * 64;
    splits = ceil_div(M, rows);
    if ((size_t)splits * N * K > L.wp_elems) return fail(FD_ERR_STATE, "weight-gradient partial region too small");
    const size_t lds_w = (size_t)FD_BWD_STAGES * 3 * 32 * 64 * 4;
    const int N32 = (N + 31) / 32 * 32;
    const size_t lds_d = ((size_t)FD_BWD_STAGES * (2 * 64 * 32 + 32 * 64) + 4 * N32 + 256) * 4;
    const bool add = P.skip_consumer >= 0;
    const unsigned n_dgrad = (unsigned)((m_tiles + 7) / 8 * 8 * k_tiles);
    *nblk = m_tiles;
    int rc;
    if (!(c.p->flags & FD_PLAN_NO_BWD_PAIRING) && !(c.p->tune & FD_TUNE_NO_PW_PAIRING)) {
        const size_t lds = std::max(lds_w, lds_d);
        const int tiles_w = n_tiles * k_tiles;
        const dim3 grid(n_dgrad + (unsigned)(tiles_w * splits));
        const int n_w = (FD_PW_BWD_W_FIRST_F32 && (unsigned)(tiles_w * splits) < n_dgrad) ? tiles_w * splits : 0;
        if (add) {
            (void)hipFuncSetAttribute((const void *)fd_pw_bwd_f32<ACT_IN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            FD_LAUNCH((fd_pw_bwd_f32<ACT_IN, 1>), grid, dim3(256), lds, c.s, G, Z, coef, c.params[i].conv_weight, tws(c.p, P.z_off), tws(c.p, P.st_off),
                      tws(c.p, P.sg_off), tws(c.p, P.g_off), bwd_rows(c.p, L.d.src, m_tiles), tws(c.p, L.wp_off), M, N, K, m_tiles, k_tiles, (int)n_dgrad, tiles_w, rows, n_w);
        } else {
            (void)hipFuncSetAttribute((const void *)fd_pw_bwd_f32<ACT_IN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            FD_LAUNCH((fd_pw_bwd_f32<ACT_IN, 0>), grid, dim3(256), lds, c.s, G, Z, coef, c.params[i].conv_weight, tws(c.p, P.z_off), tws(c.p, P.st_off),
                      (const float *)nullptr, tws(c.p, P.g_off), bwd_rows(c.p, L.d.src, m_tiles), tws(c.p, L.wp_off), M, N, K, m_tiles, k_tiles, (int)n_dgrad, tiles_w, rows, n_w);
        }
        if ((rc = check_launch("fd_pw_bwd_f32"))) return rc;
        return defer_weights(c, tws(c.p, L.wp_off), splits, N * K, 0, 0, c.grads[i].conv_weight);
    }
    {
        (void)hipFuncSetAttribute((const void *)fd_pw_wgrad_f32<ACT_IN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);
            FD_LAUNCH((fd_pw_wgrad_f32<ACT_IN>), dim3(n_tiles * k_tiles, splits), dim3(256), lds_w, c.s, G, Z, coef, tws(c.p, P.z_off), tws(c.p, P.st_off),
                  tws(c.p, L.wp_off), M, N, K, k_tiles, rows);
        if ((rc = check_launch("fd_pw_wgrad_f32"))) return rc;
        if ((rc = defer_weights(c, tws(c.p, L.wp_off), splits, N * K, 0, 0, c.grads[i].conv_weight))) return rc;
    }
    {
        dim3 grid(n_dgrad);
        if (add) {
            (void)hipFuncSetAttribute((const void *)fd_pw_dgrad_f32<ACT_IN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d);
            FD_LAUNCH((fd_pw_dgrad_f32<ACT_IN, 1>), grid, dim3(256), lds_d, c.s, G, Z, coef, c.params[i].conv_weight, tws(c.p, P.z_off), tws(c.p, P.st_off),
                      tws(c.p, P.sg_off), tws(c.p, P.g_off), bwd_rows(c.p, L.d.src, m_tiles), M, N, K, m_tiles, k_tiles);
        } else {
            (void)hipFuncSetAttribute((const void *)fd_pw_dgrad_f32<ACT_IN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d);
            FD_LAUNCH((fd_pw_dgrad_f32<ACT_IN, 0>), grid, dim3(256), lds_d, c.s, G, Z, coef, c.params[i].conv_weight, tws(c.p, P.z_off), tws(c.p, P.st_off),
                      (const float *)nullptr, tws(c.p, P.g_off), bwd_rows(c.p, L.d.src, m_tiles), M, N, K, m_tiles, k_tiles);
        }
        return check_launch("fd_pw_dgrad_f32");
    }
}

template <typename T>
int train_backward_t(fd_train_plan *plan, const fd_layer_params *params, const fd_layer_grads *grads, int32_t n_layers,
                     const void *dy, int32_t from_layer, int32_t to_layer, void *stream)
{
    constexpr bool F32 = std::is_same<T, float>::value;
    BwdCtx c{plan, params, grads, static_cast<hipStream_t>(stream)};
    hipStream_t s = c.s;
    int rc;
    // ---- head
    const int hi = n_layers - 1;
    TLayer &Hd = plan->layers[hi];
    TLayer &Hp = plan->layers[Hd.d.src];
    if (from_layer == hi) {
        fd_hs().trace_layer = hi;
        if (!plan->bwd_stats_clean) {       // a second backward after one forward: the backward rows hold the previous pass's sums
            for (int i = 0; i < n_layers; ++i)
                if (hipMemsetAsync(plan->ws + plan->layers[i].sb_off, 0, stat_rows_bytes(plan->layers[i].nr_cap, plan->layers[i].d.cout), s) != hipSuccess) return fail(FD_ERR_HIP, "hipMemsetAsync(statistics rows) failed");
        }
        plan->bwd_stats_clean = false;
        const int nb = std::min(ceil_div(Hd.M, 256), 512);     // (grid-stride: one addition to the head's single statistics channel per workgroup)
        if (Hd.d.act == FD_ACT_RELU6) FD_LAUNCH((fd_head_bwd_reduce_f32<FD_ACT_RELU6_>), dim3(nb), dim3(256), 0, s, static_cast<const float *>(dy), tws(plan, Hd.z_off), tws(plan, Hd.st_off), tws(plan, Hd.g_off), bwd_rows(plan, hi, nb), Hd.M, Hd.out_h, Hd.out_w, Hd.d.upsample);
        else FD_LAUNCH((fd_head_bwd_reduce_f32<FD_ACT_RELU_>), dim3(nb), dim3(256), 0, s, static_cast<const float *>(dy), tws(plan, Hd.z_off), tws(plan, Hd.st_off), tws(plan, Hd.g_off), bwd_rows(plan, hi, nb), Hd.M, Hd.out_h, Hd.out_w, Hd.d.upsample);
        if ((rc = check_launch("fd_head_bwd_reduce_f32"))) return rc;
        if ((rc = bn_bwd_finalize(c, hi))) return rc;
        constexpr int PPB = 16;
        const int nb2 = ceil_div(Hd.M, 32 * PPB);
        const size_t lds = (size_t)32 * Hd.d.cin * 3 * 4;
        float *wpart = tws(plan, Hd.wp_off);
        if ((size_t)nb2 * Hd.d.cin > Hd.wp_elems) return fail(FD_ERR_STATE, "weight-gradient partial region too small for the head");
        if (Hp.d.act == FD_ACT_RELU6) FD_LAUNCH((fd_head_bwd<T, FD_ACT_RELU6_, PPB>), dim3(nb2), dim3(256), lds, s, tws(plan, Hd.g_off), tws(plan, Hd.z_off), tws(plan, Hd.coef_off), twt<T>(plan, Hp.z_off), tws(plan, Hp.st_off), params[hi].conv_weight, twt<T>(plan, Hp.g_off), bwd_rows(plan, Hd.d.src, nb2), wpart, Hd.M, Hd.d.cin);
        else FD_LAUNCH((fd_head_bwd<T, FD_ACT_RELU_, PPB>), dim3(nb2), dim3(256), lds, s, tws(plan, Hd.g_off), tws(plan, Hd.z_off), tws(plan, Hd.coef_off), twt<T>(plan, Hp.z_off), tws(plan, Hp.st_off), params[hi].conv_weight, twt<T>(plan, Hp.g_off), bwd_rows(plan, Hd.d.src, nb2), wpart, Hd.M, Hd.d.cin);
        if ((rc = check_launch("fd_head_bwd"))) return rc;
        if ((rc = defer_weights(c, wpart, nb2, Hd.d.cin, 0, 0, grads[hi].conv_weight))) return rc;
        // the BatchNorm partial sums of the head's producer are in its statistics rows
        if ((rc = finalize_or_defer(c, Hd.d.src))) return rc;
    }
    // ---- remaining units in reverse order; invariant: coef_i and the BN grads of unit i are final when unit i is processed
    for (int i = std::min(hi - 1, (int)from_layer); i >= to_layer; --i) {
        TLayer &L = plan->layers[i];
        const fd_layer_desc &d = L.d;
        int nblk = 0;
        fd_hs().trace_layer = i;
        switch (d.op) {
        case FD_OP_STEM: {
            const int nblocks_w = ceil_div((long)plan->B * L.out_h * L.out_w, 256);   // blocks of 256 pixels of the flat [B*Ho*Wo] order (they may straddle images)
            const int nb_w = std::min(nblocks_w, 512);       // workgroups walk them grid-stride
            if (d.cout > 64) return fail(FD_ERR_INVALID, "stem weight gradient supports at most 64 output channels");
            float *wpart = tws(plan, L.wp_off);
            if constexpr (!F32) {
                // 16-bit plans, the 32-channel stem: row-walking fp32 kernel (fd_kernels_dw5p_bwd.h: fd_stem_wgrad_rows), one partial row per workgroup
                if (L.dw3_groups) {
                    const int wgs = ceil_div((long)L.dw3_groups * ceil_div(L.out_h, L.dw3_bh), 4), rows_w = wgs * plan->B;
                    if ((size_t)rows_w * 27 * d.cout > L.wp_elems) return fail(FD_ERR_STATE, "stem weight-gradient partial region too small");
#define FD_STEMW(CL_)                                                                                                                                     \
    FD_LAUNCH((fd_stem_wgrad_rows<T, CL_, FD_STEMW_CPL>), dim3((unsigned)wgs, (unsigned)plan->B), dim3(256), 0, s, static_cast<const float *>(plan->x_saved), twt<T>(plan, L.g_off), \
              twt<T>(plan, L.z_off), tws(plan, L.coef_off), wpart, L.in_h, L.in_w, L.dw3_groups, L.dw3_bh)
                    if (d.cout == 32) FD_STEMW(32 / FD_STEMW_CPL); else if (d.cout == 16) FD_STEMW(16 / FD_STEMW_CPL); else FD_STEMW(8 / FD_STEMW_CPL);
#undef FD_STEMW
                    if ((rc = check_launch("fd_stem_wgrad_rows"))) return rc;
                    if ((rc = defer_weights(c, wpart, rows_w, 27 * d.cout, 0, 0, grads[i].conv_weight))) return rc;
                    break;
                }
            }
            FD_LAUNCH((fd_stem_wgrad<T>), dim3(nb_w), dim3(256), (size_t)(256 * 33 + 256 * (d.cout + 1)) * 4, s, static_cast<const float *>(plan->x_saved), twt<T>(plan, L.g_off), twt<T>(plan, L.z_off), tws(plan, L.coef_off), wpart, plan->B, L.in_h, L.in_w, d.cout, nblocks_w);
            if ((rc = check_launch("fd_stem_wgrad"))) return rc;
            if ((rc = defer_weights(c, wpart, nb_w, 27 * d.cout, 0, 0, grads[i].conv_weight))) return rc;
            break;
        }
        case FD_OP_DW: {
            if (!(plan->flags & FD_PLAN_NO_BWD_PAIRING)) {
                bool paired = false;
                if ((rc = dispatch_dw_bwd_pair<T>(c, i, &nblk, &paired))) return rc;
                if (paired) {
                    if ((rc = finalize_or_defer(c, d.src))) return rc;
                    break;
                }
            }
            if ((rc = launch_dw_wgrad<T>(c, i))) return rc;
            const TLayer &P = plan->layers[d.src];
            const bool add = P.skip_consumer >= 0 && L.mode == 0;
            if (P.d.act == FD_ACT_RELU6) rc = add ? dispatch_dw_dgrad<T, FD_ACT_RELU6_, 1>(c, i, &nblk) : dispatch_dw_dgrad<T, FD_ACT_RELU6_, 0>(c, i, &nblk);
            else rc = add ? dispatch_dw_dgrad<T, FD_ACT_RELU_, 1>(c, i, &nblk) : dispatch_dw_dgrad<T, FD_ACT_RELU_, 0>(c, i, &nblk);
            if (rc) return rc;
            if ((rc = finalize_or_defer(c, d.src))) return rc;
            break;
        }
        case FD_OP_PW: {
            const TLayer &P = plan->layers[d.src];
            if constexpr (F32) rc = P.d.act == FD_ACT_RELU6 ? launch_pw_bwd<FD_ACT_RELU6_>(c, i, &nblk) : launch_pw_bwd<FD_ACT_RELU_>(c, i, &nblk);
            else rc = P.d.act == FD_ACT_RELU6 ? launch_pw_bwd_h16<T, FD_ACT_RELU6_>(c, i, &nblk) : launch_pw_bwd_h16<T, FD_ACT_RELU_>(c, i, &nblk);
            if (rc) return rc;
            if ((rc = finalize_or_defer(c, d.src))) return rc;
            break;
        }
        }
    }
    fd_hs().trace_layer = -1;
    return flush_weights(c);                                  // one launch reduces the weight-gradient partials of every unit of this range
}

}  // namespace

extern "C" {

int fd_train_backward(fd_train_plan *plan, const fd_layer_params *params, const fd_layer_grads *grads, int32_t n_layers,
                      const void *dy, void *stream)
{
    return fd_train_backward_range(plan, params, grads, n_layers, dy, n_layers - 1, 0, stream);
}

int fd_train_backward_range(fd_train_plan *plan, const fd_layer_params *params, const fd_layer_grads *grads, int32_t n_layers,
                            const void *dy, int32_t from_layer, int32_t to_layer, void *stream)
{
    if (!plan || !params || !grads || !dy) return fail(FD_ERR_INVALID, "null argument");
    if (!plan->ws || !plan->forward_done) return fail(FD_ERR_STATE, "fd_train_backward needs a preceding fd_train_forward on this plan");
    if (n_layers != (int)plan->layers.size()) return fail(FD_ERR_INVALID, "expected %zu layers", plan->layers.size());
    if (from_layer < to_layer || from_layer >= n_layers || to_layer < 0) return fail(FD_ERR_INVALID, "bad layer range %d..%d", from_layer, to_layer);
    for (int i = 0; i < n_layers; ++i)
        if (!grads[i].conv_weight || !grads[i].bn_weight || !grads[i].bn_bias) return fail(FD_ERR_INVALID, "layer %d: null gradient pointer", i);
    return plan->dtype == FD_BF16 ? train_backward_t<fd_bf16>(plan, params, grads, n_layers, dy, from_layer, to_layer, stream)
                                  : train_backward_t<float>(plan, params, grads, n_layers, dy, from_layer, to_layer, stream);
}

int fd_cast_gradients(const void *src, void *dst, int64_t numel, int32_t to_bf16, void *stream);

/* ---- data-parallel gradient exchange issued by the library itself: RCCL all-reduce over xGMI (the reference's only multi-GPU
 * idiom is torch.nn.DataParallel, imagenet/mobilenet.py:68).  RCCL is bound at RUN time (dlopen "librccl.so.1": inside a torch process that is the
 * RCCL torch itself loaded, by SONAME; a torch-free host gets the system library), so the library has no link-time dependency on it and loads on
 * hosts without RCCL.  One communicator + one non-blocking HIP stream + one event per bucket; no host synchronisation anywhere.
 * The binding is injectable (fd_comm_bind_library, csrc/fd_tuning.h): the CPU test tier binds tests/rccl_stub -- the four nccl* entry points over POSIX
 * shared memory -- to the EMULATOR build, so that this very code (bucket tiling, the 16-bit cast -> sum -> cast back, the order of the hand-overs, the
 * mean folded into fd_sgd_step) runs at world size 2 without a GPU.  Without a bound library the emulator build offers a one-rank communicator whose
 * all-reduce is the identity. ---- */
#include <dlfcn.h>
#ifdef FD_EMU
#define hipEventReleaseToDevice 0
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
#endif
struct fd_comm {
    void *comm = nullptr;                 // ncclComm_t (null: the emulator's one-rank communicator, all-reduce = identity)
    hipStream_t stream = nullptr;         // the collectives' stream
    std::vector<hipEvent_t> bucket_done;  // recorded on the compute stream after a bucket's last backward kernel
    hipEvent_t all_done = nullptr, t0 = nullptr, t1 = nullptr, tb = nullptr;   // t0 / t1: first collective issued / last one finished; tb: backward finished on the compute stream (timing enabled)
    int rank = 0, world = 1;
    bool timed = false;                   // t0 / t1 hold a pair of the last fd_train_backward_allreduce
    bool elide = false;                   // measurement hook (fd_tuning.h: fd_comm_elide_collectives): everything but the ncclAllReduce calls themselves
};

extern "C++" {
namespace {
struct fd_nccl_uid { char internal[FD_COMM_ID_BYTES]; };
struct RcclApi {
    int (*GetUniqueId)(fd_nccl_uid *) = nullptr;
    int (*CommInitRank)(void **, int, fd_nccl_uid, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
std::string &rccl_path() { static std::string p; return p; }      // set by fd_comm_bind_library before the first use
RcclApi &rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        void *h = nullptr;
        if (!rccl_path().empty()) h = dlopen(rccl_path().c_str(), RTLD_NOW | RTLD_LOCAL);
#ifndef FD_EMU
        else for (const char *name : {"librccl.so.1", "librccl.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
#endif
        if (!h) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy;
        return a;
    }();
    return api;
}
int rccl_fail(const char *what, int rc)
{
    const RcclApi &a = rccl();
    return fail(FD_ERR_HIP, "%s: RCCL error %d (%s)", what, rc, a.GetErrorString ? a.GetErrorString(rc) : "?");
}
constexpr int kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9;    // rccl.h: ncclRedOp_t / ncclDataType_t
#ifdef FD_EMU
constexpr bool kOneRankWithoutLibrary = true;     // the emulator's fallback communicator
#else
constexpr bool kOneRankWithoutLibrary = false;
#endif
}  // namespace
}  // extern "C++"

int fd_comm_bind_library(const char *path)
{
    if (!path || !*path) return fail(FD_ERR_INVALID, "fd_comm_bind_library: empty path");
    if (!rccl_path().empty() && rccl_path() != path) return fail(FD_ERR_STATE, "fd_comm_bind_library: already bound to %s", rccl_path().c_str());
    if (rccl_path().empty()) {
        // the candidate is opened and checked BEFORE the path is committed: a library that cannot be loaded (or lacks an entry point) must not leave the
        // process bound to it for good -- the one-time binding below would then never try librccl.so.1 or another path again
        void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(FD_ERR_STATE, "fd_comm_bind_library: %s cannot be loaded (%s)", path, dlerror());
        for (const char *sym : {"ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy"})
            if (!dlsym(h, sym)) { dlclose(h); return fail(FD_ERR_STATE, "fd_comm_bind_library: %s does not provide %s", path, sym); }
    }
    rccl_path() = path;
    if (!rccl().ok) return fail(FD_ERR_STATE, "fd_comm_bind_library: %s does not provide ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy (or the binding was made before this call)", path);
    return FD_OK;
}

int fd_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(FD_ERR_INVALID, "null argument");
    if (!rccl().ok) {
        if (kOneRankWithoutLibrary) { memset(id_out, 0, FD_COMM_ID_BYTES); return FD_OK; }       // (the emulator's one-rank communicator: no rendezvous)
        return fail(FD_ERR_STATE, "librccl.so.1 could not be loaded (or lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy)");
    }
    fd_nccl_uid id{};
    const int rc = rccl().GetUniqueId(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, FD_COMM_ID_BYTES);
    return FD_OK;
}

int fd_comm_create(const void *id, int32_t rank, int32_t world, fd_comm **out)
{
    if (!id || !out || world <= 0 || rank < 0 || rank >= world) return fail(FD_ERR_INVALID, "fd_comm_create: bad argument (rank %d of %d)", rank, world);
    if (!rccl().ok && !(kOneRankWithoutLibrary && world == 1))
        return fail(FD_ERR_STATE, kOneRankWithoutLibrary ? "the CPU emulator offers a one-rank communicator only unless a collective library is bound (fd_comm_bind_library)" : "librccl.so.1 could not be loaded");
    fd_comm *c = new fd_comm();
    c->rank = rank; c->world = world;
    if (rccl().ok) {
        fd_nccl_uid uid{};
        memcpy(uid.internal, id, FD_COMM_ID_BYTES);
        const int rc = rccl().CommInitRank(&c->comm, world, uid, rank);          // (on the calling thread's current device, like every entry point here)
        if (rc) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    }
    // Every event here is consumed by another stream of the SAME device (or only read for its timestamp): a device-scope release.  A default HIP event
    // performs a SYSTEM-scope fence when it is recorded -- an L2 write-back + invalidate in the middle of backward; measured on one rank: the exchange
    // machinery cost +155 us per step with default events (6 records per step), with or without the ncclAllReduce calls themselves.
    // (a default-priority stream: with the highest priority the one-rank step measured 8.6 ms instead of 2.56 -- the hand-over barriers pre-empt the
    // compute queue)
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->all_done, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess ||
        hipEventCreateWithFlags(&c->t0, hipEventReleaseToDevice) != hipSuccess || hipEventCreateWithFlags(&c->t1, hipEventReleaseToDevice) != hipSuccess ||
        hipEventCreateWithFlags(&c->tb, hipEventReleaseToDevice) != hipSuccess) {
        fd_comm_destroy(c);
        return fail(FD_ERR_HIP, "fd_comm_create: stream / event creation failed");
    }
    *out = c;
    return FD_OK;
}

void fd_comm_destroy(fd_comm *c)
{
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    for (hipEvent_t e : c->bucket_done) (void)hipEventDestroy(e);
    for (hipEvent_t e : {c->all_done, c->t0, c->t1, c->tb}) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int fd_train_backward_allreduce(fd_train_plan *plan, const fd_layer_params *params, const fd_layer_grads *grads, int32_t n_layers, const void *dy,
                                fd_comm *comm, const fd_grad_bucket *buckets, int32_t n_buckets, void *stream)
{
    if (!comm || !buckets || n_buckets <= 0) return fail(FD_ERR_INVALID, "fd_train_backward_allreduce: null communicator / empty bucket list");
    // the buckets must tile the layers n-1 .. 0 in backward order
    int expect = n_layers - 1;
    for (int b = 0; b < n_buckets; ++b) {
        if (buckets[b].from_layer != expect || buckets[b].to_layer > buckets[b].from_layer || buckets[b].to_layer < 0 || !buckets[b].grad || buckets[b].numel <= 0)
            return fail(FD_ERR_INVALID, "bucket %d: layers %d..%d do not continue the backward order at %d (or null / empty gradient slice)", b, buckets[b].from_layer, buckets[b].to_layer, expect);
        expect = buckets[b].to_layer - 1;
    }
    if (expect != -1) return fail(FD_ERR_INVALID, "the buckets stop at layer %d: they must cover every layer", expect + 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    while ((int)comm->bucket_done.size() < n_buckets) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) return fail(FD_ERR_HIP, "hipEventCreate failed");
        comm->bucket_done.push_back(e);
    }
    comm->timed = false;
    // the summing all-reduce of `count` values in place on the communicator's stream (no library bound, emulator: one rank, the identity)
    auto all_reduce = [&](void *buf, size_t count, bool bf16) -> int {
        if (comm->elide || !comm->comm) return FD_OK;
        const int nrc = rccl().AllReduce(buf, buf, count, bf16 ? kNcclBfloat16 : kNcclFloat32, kNcclSum, comm->comm, comm->stream);
        return nrc ? rccl_fail("ncclAllReduce", nrc) : FD_OK;
    };
    // Whatever happens after the first hand-over, the caller's stream must not run ahead of collectives that are already enqueued on the gradient
    // buffers: every exit below that point -- error or not -- joins the communication stream back into `s`.
    bool handed_over = false;
    auto join = [&]() -> bool {
        return hipEventRecord(comm->all_done, comm->stream) == hipSuccess && hipStreamWaitEvent(s, comm->all_done, 0) == hipSuccess;
    };
    auto bail = [&](int rc) -> int {
        if (handed_over && !join()) (void)hipStreamSynchronize(comm->stream);      // (last resort: the host waits)
        return rc;
    };
    for (int b = 0; b < n_buckets; ++b) {
        const fd_grad_bucket &k = buckets[b];
        int rc = fd_train_backward_range(plan, params, grads, n_layers, dy, k.from_layer, k.to_layer, stream);
        if (rc) return bail(rc);
        // the bucket's gradients are complete on the compute stream: the collective's stream waits for exactly that point
        if (hipEventRecord(comm->bucket_done[b], s) != hipSuccess || hipStreamWaitEvent(comm->stream, comm->bucket_done[b], 0) != hipSuccess)
            return bail(fail(FD_ERR_HIP, "event hand-over to the communication stream failed"));
        handed_over = true;
        if (b == 0) (void)hipEventRecord(comm->t0, comm->stream);
        if (k.grad16) {                                          // 16-bit exchange: convert, all-reduce the bf16 copy, convert back -- all in stream order
            if ((rc = fd_cast_gradients(k.grad, k.grad16, k.numel, 1, comm->stream))) return bail(rc);
            if ((rc = all_reduce(k.grad16, (size_t)k.numel, true))) return bail(rc);
            if ((rc = fd_cast_gradients(k.grad16, k.grad, k.numel, 0, comm->stream))) return bail(rc);
        } else if ((rc = all_reduce(k.grad, (size_t)k.numel, false))) return bail(rc);
    }
    (void)hipEventRecord(comm->tb, s);
    (void)hipEventRecord(comm->t1, comm->stream);
    // whatever the caller enqueues next on its stream (fd_sgd_step) runs after the last collective
    if (!join()) { (void)hipStreamSynchronize(comm->stream); return fail(FD_ERR_HIP, "event hand-over from the communication stream failed"); }
    comm->timed = true;
    return FD_OK;
}

void fd_comm_elide_collectives(fd_comm *comm, int32_t on) { if (comm) comm->elide = on != 0; }

int fd_comm_last_exchange_ms(fd_comm *comm, float *ms_exchange, float *ms_exposed)
{
    if (!comm || !ms_exchange || !ms_exposed) return fail(FD_ERR_INVALID, "null argument");
    if (!comm->timed) return fail(FD_ERR_STATE, "no exchange has been issued on this communicator");
    if (hipEventSynchronize(comm->t1) != hipSuccess || hipEventSynchronize(comm->tb) != hipSuccess || hipEventElapsedTime(ms_exchange, comm->t0, comm->t1) != hipSuccess)
        return fail(FD_ERR_HIP, "event timing failed");
    // exposed: how long after the last backward kernel the last collective ended (an event pair in the "wrong" order is a fully hidden exchange)
    if (hipEventElapsedTime(ms_exposed, comm->tb, comm->t1) != hipSuccess || *ms_exposed < 0.0f) *ms_exposed = 0.0f;
    return FD_OK;
}

int fd_val_transform(const void *rgb_u8, const float *depth, int32_t n, int32_t height, int32_t width, int32_t out_h, int32_t out_w,
                     const int32_t *ymap_device, const int32_t *xmap_device, float *x_out, float *depth_out, void *stream)
{
    if (!rgb_u8 || !ymap_device || !xmap_device || !x_out || n <= 0 || height <= 0 || width <= 0 || out_h <= 0 || out_w <= 0)
        return fail(FD_ERR_INVALID, "fd_val_transform: null/empty argument");
    if ((depth == nullptr) != (depth_out == nullptr)) return fail(FD_ERR_INVALID, "fd_val_transform: depth and depth_out go together");
    const long total = (long)n * out_h * out_w;
    FD_LAUNCH(fd_val_transform_u8, dim3((unsigned)std::min<long>(4096, ceil_div(total, 256))), dim3(256), 0, static_cast<hipStream_t>(stream),
              static_cast<const unsigned char *>(rgb_u8), depth, ymap_device, xmap_device, x_out, depth_out, n, height, width, out_h, out_w);
    return check_launch("fd_val_transform_u8");
}

int fd_cast_gradients(const void *src, void *dst, int64_t numel, int32_t to_bf16, void *stream)
{
    if (!src || !dst || numel <= 0) return fail(FD_ERR_INVALID, "fd_cast_gradients: null/empty argument");
    // 4 values per work-item: 16-byte accesses on the fp32 side, 8-byte accesses on the bf16 side -- each side is tested against ITS access width
    // (bucket starts are multiples of 4 elements: a bf16 pointer at 8 mod 16 is fine and must not drop the whole bucket to one element per work-item)
    const uintptr_t p32 = reinterpret_cast<uintptr_t>(to_bf16 ? src : dst), p16 = reinterpret_cast<uintptr_t>(to_bf16 ? dst : src);
    const bool al = (p32 & 15) == 0 && (p16 & 7) == 0;
    const long n4 = al ? (long)(numel >> 2) : 0;
    const unsigned nb = (unsigned)std::min<long>(2048, ceil_div((long)std::max<int64_t>(numel / 4, 1), 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (to_bf16) FD_LAUNCH(fd_cast_f32_bf16, dim3(nb), dim3(256), 0, s, static_cast<const float *>(src), static_cast<fd_bf16 *>(dst), n4, (long)numel);
    else FD_LAUNCH(fd_cast_bf16_f32, dim3(nb), dim3(256), 0, s, static_cast<const fd_bf16 *>(src), static_cast<float *>(dst), n4, (long)numel);
    return check_launch("fd_cast_gradients");
}

size_t fd_depth_metrics_scratch_bytes(void) { return (size_t)1024 * 10 * sizeof(double); }

int fd_depth_metrics(const void *output, const void *target, int64_t numel, double *sums_device, void *scratch, void *stream)
{
    if (!output || !target || !sums_device || !scratch || numel <= 0) return fail(FD_ERR_INVALID, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = (int)std::min<int64_t>(1024, (numel + 255) / 256);
    FD_LAUNCH(fd_depth_metrics_f32, dim3(nb), dim3(256), 0, s, static_cast<const float *>(output), static_cast<const float *>(target), (long)numel, static_cast<double *>(scratch));
    int rc = check_launch("fd_depth_metrics_f32");
    if (rc) return rc;
    FD_LAUNCH(fd_depth_metrics_final_f32, dim3(1), dim3(64), 0, s, static_cast<const double *>(scratch), nb, sums_device);
    return check_launch("fd_depth_metrics_final_f32");
}

size_t fd_depth_metrics_frames_scratch_bytes(int32_t n_frames) { return (size_t)std::max(n_frames, 1) * 64 * 10 * sizeof(double); }

int fd_depth_metrics_frames(const void *output, const void *target, int32_t n_frames, int64_t frame_numel, double *sums_device, void *scratch, void *stream)
{
    if (!output || !target || !sums_device || !scratch || n_frames <= 0 || frame_numel <= 0) return fail(FD_ERR_INVALID, "null argument / empty batch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = (int)std::min<int64_t>(64, (frame_numel + 255) / 256);
    FD_LAUNCH(fd_depth_metrics_f32, dim3(nb, n_frames), dim3(256), 0, s, static_cast<const float *>(output), static_cast<const float *>(target), (long)frame_numel,
              static_cast<double *>(scratch));
    int rc = check_launch("fd_depth_metrics_f32");
    if (rc) return rc;
    FD_LAUNCH(fd_depth_metrics_final_f32, dim3(n_frames), dim3(64), 0, s, static_cast<const double *>(scratch), nb, sums_device);
    return check_launch("fd_depth_metrics_final_f32");
}

size_t fd_l1_loss_scratch_bytes(int64_t numel) { (void)numel; return 2 * 1024 * sizeof(float); }   // masked form: (sum, count) per workgroup

int fd_l1_loss(const void *pred, const void *target, void *dpred, float *loss_out, int64_t numel, void *scratch, void *stream)
{
    if (!pred || !target || !dpred || !loss_out || !scratch || numel <= 0) return fail(FD_ERR_INVALID, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = (int)std::min<int64_t>(1024, (numel + 255) / 256);
    const float inv = 1.0f / (float)numel;
    FD_LAUNCH(fd_l1_loss_f32, dim3(nb), dim3(256), 0, s, static_cast<const float *>(pred), static_cast<const float *>(target), static_cast<float *>(dpred),
              static_cast<float *>(scratch), (long)numel, inv);
    int rc = check_launch("fd_l1_loss_f32");
    if (rc) return rc;
    FD_LAUNCH(fd_l1_loss_final_f32, dim3(1), dim3(64), 0, s, static_cast<const float *>(scratch), nb, inv, loss_out);
    return check_launch("fd_l1_loss_final_f32");
}

int fd_l1_loss_masked(const void *pred, const void *target, void *dpred, float *loss_out, int64_t numel, void *scratch, void *stream)
{
    if (!pred || !target || !dpred || !loss_out || !scratch || numel <= 0) return fail(FD_ERR_INVALID, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = (int)std::min<int64_t>(1024, (numel + 255) / 256);
    FD_LAUNCH(fd_l1_masked_partial_f32, dim3(nb), dim3(256), 0, s, static_cast<const float *>(pred), static_cast<const float *>(target), static_cast<float *>(scratch), (long)numel);
    int rc = check_launch("fd_l1_masked_partial_f32");
    if (rc) return rc;
    FD_LAUNCH(fd_l1_masked_apply_f32, dim3(nb), dim3(256), 0, s, static_cast<const float *>(pred), static_cast<const float *>(target), static_cast<const float *>(scratch), nb,
              static_cast<float *>(dpred), loss_out, (long)numel);
    return check_launch("fd_l1_masked_apply_f32");
}

int fd_sgd_step(const fd_sgd_tensor *table_device, int32_t n_tensors, int64_t total_numel, float lr, float momentum, float weight_decay,
                float grad_scale, int32_t first_step, void *stream)
{
    if (!table_device || n_tensors <= 0) return fail(FD_ERR_INVALID, "null/empty tensor table");
    static_assert(sizeof(fd_sgd_tensor) == sizeof(fd_sgd_rec), "table record layout");
    (void)total_numel;
    FD_LAUNCH(fd_sgd_f32, dim3(64, n_tensors), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const fd_sgd_rec *>(table_device), n_tensors,
              lr, momentum, weight_decay, grad_scale, first_step);
    return check_launch("fd_sgd_f32");
}

}  // extern "C"
