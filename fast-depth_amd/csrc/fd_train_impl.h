// fd_train_impl.h -- train plan creation and the train-mode forward (translation unit fd_train_fwd.hip; records and helpers: fd_train_plan.h)
#pragma once
#include "fd_train_plan.h"

namespace {

template <typename T, int ACT1, int ACT2>
int launch_dw_train(const TLayer &L, const T *zin, const float *st1, const T *zskip, const float *st2, const float *w,
                    T *zout, fd_stat_rows part, hipStream_t s, int batch, const fd_bn_fin &fin)
{
    L.lds_rounding = (L.lds_rounding & ~(1 | 4)) | ((!L.rows_th && L.dw_n == 8) ? 1 : 0);
    if (L.dw5_groups) {                                       // 16-bit plans, 5x5 on up2 + skip: row-walking pixel-pair kernel (input AND taps rounded to the storage type)
        if constexpr (!std::is_same<T, float>::value) {
            L.lds_rounding |= 1 | 4;
            FD_LAUNCH((fd_dw5_rows_train<T, ACT1, ACT2>), L.grid, dim3(256), 0, s, zin, st1, zskip, st2, w, zout, part, L.in_h, L.in_w, L.d.cin, L.dw5_groups, L.dw5_bh, fin);
            return check_launch("fd_dw5_rows_train");
        }
    }
    if (L.dw3_cl) {                                           // 16-bit plans, 3x3 on a plain input: row-walking fp32-window kernel (nothing rounded but the stored output)
        if constexpr (!std::is_same<T, float>::value) {
            const int key3 = L.d.stride * 100 + L.dw3_cl;
#define FD_DW3F(S_, CL_)                                                                                                                                   \
    case S_ * 100 + CL_:                                                                                                                                  \
        FD_LAUNCH((fd_dw3_rows_fwd<T, S_, ACT1, CL_>), L.grid, dim3(256), 0, s, zin, st1, w, zout, part, L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.dw3_groups, L.dw3_bh, fin); \
        break;
            switch (key3) {
                FD_DW3F(1, 16) FD_DW3F(1, 32) FD_DW3F(2, 16) FD_DW3F(2, 32)
            default: return fail(FD_ERR_STATE, "train: fd_dw3_rows_fwd has no instance for stride %d, %d channel lanes", L.d.stride, L.dw3_cl);
            }
#undef FD_DW3F
            return check_launch("fd_dw3_rows_fwd");
        }
    }
    if (L.rows_th) {                                          // register-window kernel (3x3, plain input, large maps)
        if (L.d.stride == 1) FD_LAUNCH((fd_dw3_rows_train<T, 1, ACT1>), L.grid, dim3(256), 0, s, zin, st1, w, zout, part, L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.rows_th, fin);
        else FD_LAUNCH((fd_dw3_rows_train<T, 2, ACT1>), L.grid, dim3(256), 0, s, zin, st1, w, zout, part, L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.rows_th, fin);
        return check_launch("fd_dw3_rows_train");
    }
    const int key = L.d.ksize * 100 + L.d.stride * 10 + L.mode;
#define FD_DWT(K_, S_, M_)                                                                                                   \
    case K_ * 100 + S_ * 10 + M_:                                                                                            \
        fd_by_lane_width<T>(L.dw_n, [&](auto nt) {                                                                          \
            constexpr int NL = decltype(nt)::value;                                                                         \
            if (L.lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)fd_dwconv_train<T, K_, S_, M_, ACT1, ACT2, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds); \
            FD_LAUNCH((fd_dwconv_train<T, K_, S_, M_, ACT1, ACT2, NL>), L.grid, dim3(256), L.lds, s, zin, st1, zskip, st2, w, zout, part, \
                      L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.cbq, L.th, L.tw, L.tiles_x, L.csplit, L.pstr, fin);       \
        });                                                                                                                  \
        break;
    switch (key) {
        FD_DWT(3, 1, 0) FD_DWT(3, 2, 0) FD_DWT(5, 1, 0) FD_DWT(5, 1, 1) FD_DWT(5, 1, 2) FD_DWT(5, 1, 3)
    default: return fail(FD_ERR_INVALID, "train: depthwise k=%d stride=%d mode=%d has no kernel", L.d.ksize, L.d.stride, L.mode);
    }
#undef FD_DWT
    return check_launch("fd_dwconv_train");
}

// activation of the producer(s) decides the template instance: encoder = ReLU6, decoder = ReLU (both appear as act1; act2 is
// the skip tensor's activation, always an encoder unit)
template <typename T>
int dispatch_dw_train(const TLayer &L, int act1, int act2, const T *zin, const float *st1, const T *zskip, const float *st2,
                      const float *w, T *zout, fd_stat_rows part, hipStream_t s, int batch, const fd_bn_fin &fin)
{
    if (act1 == FD_ACT_RELU6 && act2 == FD_ACT_RELU6) return launch_dw_train<T, FD_ACT_RELU6_, FD_ACT_RELU6_>(L, zin, st1, zskip, st2, w, zout, part, s, batch, fin);
    if (act1 == FD_ACT_RELU && act2 == FD_ACT_RELU6) return launch_dw_train<T, FD_ACT_RELU_, FD_ACT_RELU6_>(L, zin, st1, zskip, st2, w, zout, part, s, batch, fin);
    if (act1 == FD_ACT_RELU && act2 == FD_ACT_RELU) return launch_dw_train<T, FD_ACT_RELU_, FD_ACT_RELU_>(L, zin, st1, zskip, st2, w, zout, part, s, batch, fin);
    if (act1 == FD_ACT_RELU6 && act2 == FD_ACT_RELU) return launch_dw_train<T, FD_ACT_RELU6_, FD_ACT_RELU_>(L, zin, st1, zskip, st2, w, zout, part, s, batch, fin);
    return fail(FD_ERR_INVALID, "train: unsupported producer activations %d/%d", act1, act2);
}

template <typename T>
int train_forward_t(fd_train_plan *plan, const fd_layer_params *params, int32_t n_layers, float bn_eps, float bn_momentum,
                    const void *x_nchw, void *y, void *stream)
{
    constexpr bool F32 = std::is_same<T, float>::value;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float *x = static_cast<const float *>(x_nchw);
    plan->eps = bn_eps;
    plan->x_saved = x_nchw;
    if constexpr (!F32) {
        // 16-bit operand copies (W as [N][K64], W^T as [K][N64]) of the live fp32 master weights of every pointwise unit, FD_PACK_MAX
        // units per launch; read by this step's forward and backward
        fd_pack_table<T> tab;
        int cnt = 0;
        auto flush = [&]() -> int {
            if (!cnt) return FD_OK;
            FD_LAUNCH((fd_pack_train_w_h16<T>), dim3(512, (unsigned)cnt), dim3(256), 0, s, tab);
            cnt = 0;
            return check_launch("fd_pack_train_w_h16");
        };
        for (int i = 0; i < n_layers; ++i) {
            const TLayer &L = plan->layers[i];
            if (L.d.op != FD_OP_PW || L.head) continue;
            if (!params[i].conv_weight) return fail(FD_ERR_INVALID, "layer %d: null parameter pointer", i);
            tab.rec[cnt++] = fd_pack_rec<T>{params[i].conv_weight, twt<T>(plan, L.wt_off), twt<T>(plan, L.wtt_off), L.d.cout, L.d.cin, L.k64, L.n64};
            if (cnt == FD_PACK_MAX) { int rc = flush(); if (rc) return rc; }
        }
        int rc = flush();
        if (rc) return rc;
    }
    // the statistics rows of every unit, both directions, start the step at zero: one memset instead of 57 finalisation launches
    if (hipMemsetAsync(plan->ws + plan->stat_off, 0, plan->stat_bytes, s) != hipSuccess) return fail(FD_ERR_HIP, "hipMemsetAsync(statistics rows) failed");
    plan->bwd_stats_clean = true;
    // the finalisation of unit u as its consumer's kernel (or fd_bn_finalize_rows_f32) performs it
    auto fin_of = [&](int u) {
        const TLayer &U = plan->layers[u];
        const fd_layer_params &pq = params[u];
        return fd_bn_fin{stat_ptr(plan, U.sf_off), U.nr_f, stat_pitch(U.d.cout), U.n_stat, U.n_unbiased, 1.0 / U.n_stat, bn_eps, bn_momentum, pq.bn_weight, pq.bn_bias,
                         const_cast<float *>(pq.bn_mean), const_cast<float *>(pq.bn_var), tws(plan, U.st_off), reinterpret_cast<long long *>(pq.bn_num_batches_tracked)};
    };
    for (int i = 0; i < n_layers; ++i)
        if (!params[i].conv_weight || !params[i].bn_weight || !params[i].bn_bias || !params[i].bn_mean || !params[i].bn_var) return fail(FD_ERR_INVALID, "layer %d: null parameter pointer", i);
    for (int i = 0; i < n_layers; ++i) {
        const TLayer &L = plan->layers[i];
        const fd_layer_desc &d = L.d;
        const fd_layer_params &q = params[i];
        fd_hs().trace_layer = i;
        T *z = twt<T>(plan, L.z_off);
        const TLayer *P = d.src >= 0 ? &plan->layers[d.src] : nullptr;
        const T *zin = P ? twt<T>(plan, P->z_off) : nullptr;
        const float *st1 = P ? tws(plan, P->st_off) : nullptr;
        int rc = FD_OK;
        const fd_stat_rows part = fwd_rows(plan, L);
        const fd_bn_fin fin = (P && P->fin_by_consumer) ? fin_of(d.src) : fd_bn_fin{};       // the producer's finalisation runs in this unit's kernel
        switch (d.op) {
        case FD_OP_STEM:
            FD_LAUNCH((fd_stem_train<T>), L.grid, dim3(256), L.lds, s, x, q.conv_weight, z, part, L.in_h, L.in_w, d.cout, (int)(L.lds / 4) - 4 * 2 * 32);
            rc = check_launch("fd_stem_train");
            break;
        case FD_OP_DW: {
            const TLayer *K = d.skip >= 0 ? &plan->layers[d.skip] : nullptr;
            rc = dispatch_dw_train<T>(L, P->d.act, K ? K->d.act : FD_ACT_RELU6, zin, st1, K ? twt<T>(plan, K->z_off) : (const T *)nullptr,
                                      K ? tws(plan, K->st_off) : nullptr, q.conv_weight, z, part, s, plan->B, fin);
            break;
        }
        case FD_OP_PW:
            if (L.head) {
                const long npix = L.M;
                float *zl = tws(plan, L.z_off);
                if (P->d.act == FD_ACT_RELU6) FD_LAUNCH((fd_head_train<T, FD_ACT_RELU6_>), L.grid, dim3(256), 0, s, zin, st1, q.conv_weight, zl, part, npix, d.cin, fin);
                else FD_LAUNCH((fd_head_train<T, FD_ACT_RELU_>), L.grid, dim3(256), 0, s, zin, st1, q.conv_weight, zl, part, npix, d.cin, fin);
                rc = check_launch("fd_head_train");
            } else {
                if constexpr (F32) {
                    if (L.pw16_tm) {
                        const fd_g16_train tr{st1, part, fin};
#define FD_PW16T(TMV, ACTV)                                                                                                                          \
    do {                                                                                                                                              \
        (void)hipFuncSetAttribute((const void *)fd_pw_gemm16_f32<TMV, FD_G16_STAGES, 0, 0, 0, ACTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds); \
        FD_LAUNCH((fd_pw_gemm16_f32<TMV, FD_G16_STAGES, 0, 0, 0, ACTV>), L.grid, dim3(512), L.lds, s, zin, q.conv_weight, (const float *)nullptr, z, (int)L.M, d.cout, d.cin, \
                  d.cin, L.pw16_stride, L.m_tiles, L.n_tiles, fd_dwfuse{}, tr);                                                                       \
    } while (0)
#define FD_PW16T_CASE(TMV) case TMV: if (P->d.act == FD_ACT_RELU6) FD_PW16T(TMV, FD_ACT_RELU6_); else FD_PW16T(TMV, FD_ACT_RELU_); break;
                        switch (L.pw16_tm) {
                            FD_PW16T_CASE(13) FD_PW16T_CASE(7) FD_PW16T_CASE(4)
                        default: return fail(FD_ERR_INVALID, "no gemm16 instance for TM=%d", L.pw16_tm);
                        }
#undef FD_PW16T_CASE
#undef FD_PW16T
                        rc = check_launch("fd_pw_gemm16_f32");
                        break;
                    }
                    if (P->d.act == FD_ACT_RELU6) FD_LAUNCH((fd_pw_gemm_train_f32<FD_ACT_RELU6_>), L.grid, dim3(256), L.lds, s, zin, st1, q.conv_weight, z, part, (int)L.M, d.cout, d.cin, L.m_tiles, L.n_tiles, fin);
                    else FD_LAUNCH((fd_pw_gemm_train_f32<FD_ACT_RELU_>), L.grid, dim3(256), L.lds, s, zin, st1, q.conv_weight, z, part, (int)L.M, d.cout, d.cin, L.m_tiles, L.n_tiles, fin);
                    rc = check_launch("fd_pw_gemm_train_f32");
                } else {
                    T *wt = twt<T>(plan, L.wt_off);      // 16-bit operand copies of the master weights: made for all units at the start of the step
#define FD_PWT_H16(ACTV, TNV)                                                                                                                     \
    do {                                                                                                                                          \
        (void)hipFuncSetAttribute((const void *)fd_pw_gemm_train_h16<T, ACTV, TNV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds);     \
        FD_LAUNCH((fd_pw_gemm_train_h16<T, ACTV, TNV>), L.grid, dim3(256), L.lds, s, zin, st1, wt, z, part, (int)L.M, d.cout, d.cin, L.k64, L.m_tiles, L.n_tiles, fin); \
    } while (0)
                    if (P->d.act == FD_ACT_RELU6) { if (L.pw_tn == 2) FD_PWT_H16(FD_ACT_RELU6_, 2); else FD_PWT_H16(FD_ACT_RELU6_, 1); }
                    else { if (L.pw_tn == 2) FD_PWT_H16(FD_ACT_RELU_, 2); else FD_PWT_H16(FD_ACT_RELU_, 1); }
#undef FD_PWT_H16
                    rc = check_launch("fd_pw_gemm_train_h16");
                }
            }
            break;
        }
        if (rc) return rc;
        if (L.fin_by_consumer) continue;
        const fd_bn_fin own = fin_of(i);
        FD_LAUNCH(fd_bn_finalize_rows_f32, dim3((unsigned)ceil_div(d.cout, 16)), dim3(256), 0, s, own, d.cout);
        if ((rc = check_launch("fd_bn_finalize_rows_f32"))) return rc;
    }
    const TLayer &Hd = plan->layers.back();
    const fd_bn_fin hfin = Hd.fin_by_consumer ? fin_of(n_layers - 1) : fd_bn_fin{};     // the head's own BatchNorm: finalised by the kernel that writes the prediction
    if (Hd.d.act == FD_ACT_RELU6)
        FD_LAUNCH((fd_head_apply_f32<FD_ACT_RELU6_>), dim3(ceil_div(Hd.M, 256)), dim3(256), 0, s, tws(plan, Hd.z_off), tws(plan, Hd.st_off), static_cast<float *>(y), Hd.M, Hd.out_h, Hd.out_w, Hd.d.upsample, hfin);
    else
        FD_LAUNCH((fd_head_apply_f32<FD_ACT_RELU_>), dim3(ceil_div(Hd.M, 256)), dim3(256), 0, s, tws(plan, Hd.z_off), tws(plan, Hd.st_off), static_cast<float *>(y), Hd.M, Hd.out_h, Hd.out_w, Hd.d.upsample, hfin);
    int rc = check_launch("fd_head_apply_f32");
    fd_hs().trace_layer = -1;
    if (rc) return rc;
    plan->forward_done = true;
    return FD_OK;
}

}  // namespace

extern "C" {

int fd_train_plan_create(const fd_layer_desc *layers, int32_t n_layers, int32_t batch, int32_t height, int32_t width,
                         int32_t dtype, uint32_t flags, fd_train_plan **out_plan)
{
    const uint32_t tune = fd_take_tuning();                  // (consumed even when the creation fails)
    if (!layers || !out_plan || n_layers <= 0) return fail(FD_ERR_INVALID, "null/empty layer list");
    if (tune & ~FD_TUNE_ALL) return fail(FD_ERR_INVALID, "unknown tuning bits 0x%x", tune & ~FD_TUNE_ALL);
    if (batch <= 0 || height <= 0 || width <= 0 || height % 32 || width % 32)
        return fail(FD_ERR_INVALID, "batch must be > 0 and height/width positive multiples of 32 (got %d, %dx%d)", batch, height, width);
    if (dtype != FD_F32 && dtype != FD_BF16)
        return fail(FD_ERR_INVALID, "train plan: dtype %d not supported (fp32 or bf16; fp16 gradients would need loss scaling)", dtype);
    if (flags & ~FD_PLAN_ALL_FLAGS) return fail(FD_ERR_INVALID, "unknown plan flag bits 0x%x", flags & ~FD_PLAN_ALL_FLAGS);
    fd_train_plan *p = new fd_train_plan();
    p->B = batch; p->H = height; p->W = width; p->dtype = dtype; p->flags = flags;
    p->tune = tune;
    const bool h16 = dtype != FD_F32;
    const size_t esz = h16 ? 2 : 4;
    p->esz = esz;
    p->layers.resize(n_layers);
    size_t off = 0, max_g = 0;
#define FD_BAD(...) do { int rc_ = fail(FD_ERR_INVALID, __VA_ARGS__); delete p; return rc_; } while (0)
    for (int i = 0; i < n_layers; ++i) {
        TLayer &L = p->layers[i];
        L.d = layers[i];
        const fd_layer_desc &d = L.d;
        if (d.src >= i || d.skip >= i) FD_BAD("layer %d: src/skip must reference earlier layers", i);
        if (d.act != FD_ACT_RELU && d.act != FD_ACT_RELU6) FD_BAD("layer %d: train mode needs ReLU or ReLU6", i);
        int src_h, src_w, src_c;
        if (d.src < 0) { src_h = height; src_w = width; src_c = 3; }
        else { const TLayer &S = p->layers[d.src]; src_h = S.out_h; src_w = S.out_w; src_c = S.d.cout; }
        const bool concat = d.concat != 0;
        if (concat) {
            if (d.skip < 0 || d.op != FD_OP_DW || !d.upsample) FD_BAD("layer %d: concat needs an upsampled depthwise consumer with a skip tensor", i);
            L.csplit = src_c;
            if (src_c + p->layers[d.skip].d.cout != d.cin || src_c % 32) FD_BAD("layer %d: concat of %d + %d channels does not give cin %d (the first part must be a multiple of 32)", i, src_c, p->layers[d.skip].d.cout, d.cin);
        } else if (src_c != d.cin) FD_BAD("layer %d: cin %d != producer channels %d", i, d.cin, src_c);
        if (h16 && (d.cout % 8 || (d.op != FD_OP_STEM && d.cin % 8)) && !(d.op == FD_OP_PW && d.cout == 1))
            FD_BAD("layer %d: the 16-bit train plan needs channel counts that are multiples of 8", i);
        L.in_h = d.upsample ? 2 * src_h : src_h;
        L.in_w = d.upsample ? 2 * src_w : src_w;
        if (d.src >= 0) { if (p->layers[d.src].consumer >= 0) FD_BAD("layer %d: producer %d already has a consumer", i, d.src); p->layers[d.src].consumer = i; }
        if (d.skip >= 0) {
            const TLayer &S = p->layers[d.skip];
            if (!d.upsample || S.out_h != L.in_h || S.out_w != L.in_w || (!concat && S.d.cout != d.cin)) FD_BAD("layer %d: bad skip", i);
            if (p->layers[d.skip].skip_consumer >= 0) FD_BAD("layer %d: skip source %d used twice", i, d.skip);
            p->layers[d.skip].skip_consumer = i;
        }
        switch (d.op) {
        case FD_OP_STEM:
            if (d.src != -1 || d.cin != 3 || d.ksize != 3 || d.stride != 2 || d.upsample || d.skip >= 0 || d.cout % 8) FD_BAD("layer %d: bad stem", i);
            L.out_h = L.in_h / 2; L.out_w = L.in_w / 2;
            L.chunk = d.cout % 32 == 0 ? 32 : (d.cout % 16 == 0 ? 16 : 8);
            if (d.cout > 64) FD_BAD("layer %d: the stem supports at most 64 output channels", i);
            {   // blocks of 256 consecutive output pixels of ONE image; LDS: the zero-padded band of input rows under a block (3 planes; reused as
                // the four waves' [64][36] output tiles) + [4][2][32] statistics
                const int bpi = ceil_div((long)L.out_h * L.out_w, 256), nrows = 2 * ceil_div(255, L.out_w) + 3;
                L.stem_band = std::max(3 * nrows * (L.in_w + 8), 4 * 64 * 36);
                L.lds = (size_t)(L.stem_band + 4 * 2 * 32) * 4;
                L.grid = dim3(bpi, batch);
                L.nblk = bpi * batch;
            }
            L.wp_elems = (size_t)std::min(L.nblk, 512) * 27 * d.cout;
            // 16-bit plans, 32 / 16 / 8 output channels: the backward-weights pass runs on the row-walking kernel fd_stem_wgrad_rows (bands of ~7 output rows, 16 output
            // columns per wave); dw3_groups / dw3_bh carry its column groups per row and rows per band
            if (h16 && (d.cout == 32 || d.cout == 16 || d.cout == 8) && L.out_w % 4 == 0 && L.in_h == 2 * L.out_h && L.in_w == 2 * L.out_w && !(tune & FD_TUNE_NO_DW5_ROWS)) {
                L.dw3_groups = ceil_div(L.out_w, 4 * (64 / (d.cout / FD_STEMW_CPL)));     // a wave: cout / FD_STEMW_CPL channel lanes, the rest of its 64 lanes column groups of 4 output columns
                L.dw3_bh = ceil_div(L.out_h, std::max(1, (L.out_h + FD_STEMW_BAND / 2) / FD_STEMW_BAND));
                const long rows_w = (long)ceil_div((long)L.dw3_groups * ceil_div(L.out_h, L.dw3_bh), 4) * batch;
                L.wp_elems = std::max(L.wp_elems, (size_t)rows_w * 27 * d.cout);
            }
            break;
        case FD_OP_DW: {
            if (d.src < 0 || d.cin != d.cout || (d.ksize != 3 && d.ksize != 5) || (d.stride != 1 && d.stride != 2) || d.cin % 4) FD_BAD("layer %d: bad depthwise", i);
            L.mode = d.upsample ? (d.skip >= 0 ? (concat ? 3 : 2) : 1) : 0;
            L.out_h = L.in_h / d.stride; L.out_w = L.in_w / d.stride;
            // bf16 plans CAN run these kernels with 8 channels (16 bytes) per work-item and LDS patches kept in bf16 (fd_lane<T, 8>: a 64-channel block
            // has the LDS footprint and the per-workgroup instruction count of the 32-channel fp32-patch block) -- FD_TUNE_FORCE_DW_H8.  Measured at batch 32
            // (round 4, us per launch, 8- vs 4-channel form): paired backward conv1.0 95.6 vs 51.3, conv3.0 83.6 vs 54.4, conv5.0 45.5 vs 32.0, 14x14 units
            // 25.1 vs 19.3, decode_conv5.0 148.4 vs 151.6, decode_conv4.0 79.6 vs 81.8; forward 208.6 vs 203.7 over the family: the kernels are VALU-issue
            // bound (PMC: 61 % VALU busy at 23 % parked), not residency bound, and the wider lanes cost registers.  Default: 4.
            L.dw_n = (h16 && (tune & FD_TUNE_FORCE_DW_H8) && !(tune & FD_TUNE_NO_DW_H8)) ? 8 : 4;
            const int cb_max = (tune & FD_TUNE_DW_CB16) ? 4 * L.dw_n : 8 * L.dw_n;
            int cb = L.dw_n;                                      // the largest power-of-two block <= cin, at most cb_max
            while (cb * 2 <= cb_max && cb * 2 <= d.cin) cb *= 2;
            if (L.dw_n == 8 && cb == 64 && ceil_div(d.cin, 64) * 64 > ceil_div(d.cin, 32) * 32) cb = 32;     // (pruned widths: the block size that pads the channel count least)
            L.cbq = ilog2(cb / L.dw_n);
            const int le = L.dw_n == 8 ? 2 : 4;                   // bytes per LDS patch element
            const bool k5 = d.ksize == 5;
            L.tw = std::min((L.out_w + 3) / 4 * 4, d.stride == 2 ? 8 : (k5 ? FD_T_DW5_FTW : 16));
            L.th = (tune & FD_TUNE_DW_TH8) ? std::min(L.out_h, 8) : ceil_div(L.out_h, ceil_div(L.out_h, k5 ? FD_T_DW5_FTH : 8));   // balanced rows: 14 -> 7 + 7 instead of 8 + 6 (both tiles full, smaller patches: one more workgroup per CU)
            // The backward kernels keep these tiles (L.bth / L.btw).  The FORWARD kernel takes larger ones on stride-1 units: a workgroup's life is dominated
            // by fixed costs (tap / table loads ~2 us, barriers, the reduction: 4 us even with no patch loads or stores at all --
            // tools/microbench/dwtrain.hip), so fewer, fatter workgroups win until the patch staging takes too many load rounds:
            // measured 56x56x128: 7x16 39.4 us, 14x28 30.4; 28x28x256: 20.3 -> 14.6; 112x112x32: 8x16 34.9 -> 16x16 32.1; 14x14: 7x16 10.6 -> 14x16 9.1
            L.bth = L.th; L.btw = L.tw;
            if (k5) { L.btw = std::min((L.out_w + 3) / 4 * 4, FD_T_DW5_WTW); L.bth = ceil_div(L.out_h, ceil_div(L.out_h, FD_T_DW5_WTH)); }
            // (5x5 weight-gradient tiles of 4 rows -- 32 KB of LDS instead of 53 -- measured slower: decode_conv5 156 -> 169 us)
            if (d.ksize == 5 && (tune & FD_TUNE_DW_WGRAD_TH4)) L.bth = ceil_div(L.out_h, ceil_div(L.out_h, 4));
            // (3x3 stride-1 weight-gradient tiles of 7 rows -- 39.5 KB instead of 44.4: four resident workgroups per CU -- measured neutral in the paired launch)
            // (3x3 units only: the 5x5 decoder units LOSE with larger tiles -- 34 -> 50 us at 56x56, their patch staging takes too many load rounds)
            if (d.stride == 1 && d.ksize == 3 && !(tune & FD_TUNE_DW_SMALL_TILES)) {
                L.th = L.out_h <= 14 ? L.out_h : (L.out_h <= 56 ? 14 : 16);
                L.tw = L.out_w <= 16 ? (L.out_w + 3) / 4 * 4 : (L.out_w <= 56 ? 28 : 16);
            }
            L.tiles_x = ceil_div(L.out_w, L.tw); L.tiles_y = ceil_div(L.out_h, L.th);
            const int th_in = (L.th - 1) * d.stride + d.ksize, tw_in = (L.tw - 1) * d.stride + d.ksize;
            // pitch in LDS elements: the block's channels + 4 dwords (FD_TUNE_DW_PITCH4 / 8: +4 / +8 / +12 dwords more)
            L.pstr = L.bpstr = cb + (4 + 4 * (int)((tune / FD_TUNE_DW_PITCH4) & 3)) * (4 / le);
            L.lds = lds_patch_bytes((long)th_in * tw_in, L.pstr, le) + (size_t)d.ksize * d.ksize * cb * 4;
            L.grid = dim3(L.tiles_x * L.tiles_y, ceil_div(d.cin, cb), batch);
            L.nblk = L.tiles_x * L.tiles_y * batch;
            {   // FORWARD of the 3x3 units on plain inputs whose channel-group count C/4 is a power of two in 8 ... 64 (32 ... 256 channels: the
                // large maps; the plan takes it up to 128 channels): the register-window kernel (fd_dw3_rows_train, no LDS staging); row strips as high as still leave >= ~1024 workgroups
                const int cg = d.cin / 4;
                const bool rows_ok = d.ksize == 3 && L.mode == 0 && d.cin % 4 == 0 && cg >= 8 && cg <= 64 && (cg & (cg - 1)) == 0;
                // measured (bf16, us, rows vs tiled): conv1.0 19.2 / 22.9, conv2.0 15.1 / 29.2, conv3.0 19.0 / 21.1, conv4.0 10.5 / 15.1, conv5.0 (256 channels) 13.8 / 11.7
                if (rows_ok && !(tune & FD_TUNE_DW_NO_ROWS) && (((long)L.out_h * L.out_w >= 28 * 28 && cg <= 32) || (tune & FD_TUNE_DW_FORCE_ROWS))) {
                    const int gx = ceil_div((long)L.out_w * cg, 256);
                    int th = L.out_h;
                    while (th > 4 && (long)gx * ceil_div(L.out_h, th) * batch < 1024) th = (th + 1) / 2;
                    L.rows_th = th;
                    L.grid = dim3(gx, ceil_div(L.out_h, th), batch);
                    L.nblk = gx * ceil_div(L.out_h, th) * batch;
                    L.lds = 0;
                }
            }
            // 16-bit plans, 5x5 on up2 + skip (decode_conv3 / 4 / 5 .0): the row-walking pixel-pair kernel (fd_kernels_dw5p_bwd.h: fd_dw5_rows_train); bands of ~14 rows
            if (h16 && k5 && d.stride == 1 && L.mode == 2 && L.out_w % 4 == 0 && L.in_h % 2 == 0 && d.cin % 8 == 0 && (double)L.in_h * L.in_w * d.cin * 2.0 < 2147483648.0 &&
                !(tune & (FD_TUNE_NO_DW5_ROWS | FD_TUNE_FORCE_DW_H8))) {
                L.dw5_groups = ceil_div(L.out_w, 8);
                const int bands = std::max(1, (L.out_h + 7) / 14);
                L.dw5_bh = ceil_div(ceil_div(L.out_h, bands), 2) * 2;
                const int wgs = ceil_div((long)L.dw5_groups * ceil_div(L.out_h, L.dw5_bh), 4);
                L.grid = dim3(wgs, ceil_div(d.cin, 64), batch);
                L.nblk = wgs * batch;
                L.lds = 0;
            }
            // 16-bit plans, 3x3 on a plain input (every encoder unit): the row-walking fp32-window kernel (fd_kernels_dw5p_bwd.h: fd_dw3_rows_fwd) -- bands of ~14 output
            // rows at stride 1, ~7 at stride 2 (14 input rows); the 14x14 maps take 7, the 7x7 maps 4 (twice the waves: those launches are latency-bound)
            if (h16 && d.ksize == 3 && L.mode == 0 && d.cin % 8 == 0 && (d.stride == 1 || (L.in_h % 2 == 0 && L.in_w % 2 == 0)) &&
                (double)L.in_h * L.in_w * d.cin * 2.0 < 2147483648.0 &&
                !(tune & (FD_TUNE_NO_DW5_ROWS | FD_TUNE_FORCE_DW_H8 | FD_TUNE_DW_NO_ROWS | FD_TUNE_DW_FORCE_ROWS))) {
                L.rows_th = 0;
                L.dw3_cl = d.cin <= 32 ? 16 : 32;                // (a 32-channel unit would leave half of every wave idle at 32 lanes per strip)
                L.dw3_groups = ceil_div(L.out_w, 4 * (64 / L.dw3_cl));
                const int tall = L.out_h <= 7 ? 4 : (L.out_h <= 14 ? 7 : (d.stride == 1 ? 14 : 7));
                const int bands = std::max(1, (L.out_h + tall / 2) / tall);
                L.dw3_bh = ceil_div(L.out_h, bands);
                const int wgs = ceil_div((long)L.dw3_groups * ceil_div(L.out_h, L.dw3_bh), 4);
                L.grid = dim3(wgs, ceil_div(d.cin, 2 * L.dw3_cl), batch);
                L.nblk = wgs * batch;
                L.lds = 0;
            }
            if (L.mode != 3 && !(tune & FD_TUNE_NO_CONSUMER_FINALIZE) && !(L.rows_th && d.cin > 256) &&
                p->layers[d.src].nr_f <= (L.rows_th ? FD_STAT_FIN_MAX_ROWS_ALL : ((L.dw3_cl || L.dw5_groups) ? FD_STAT_FIN_MAX_ROWS_ROWK : FD_STAT_FIN_MAX_ROWS_BLOCK))) {
                // the producer's BatchNorm is finalised by this kernel's workgroups from the producer's statistics rows (fd_stat_table_block in the LDS-tiled
                // kernel, which keeps the block's (scale, shift) behind its tap table; the register-window kernel holds all C <= 256 channels in its static LDS)
                p->layers[d.src].fin_by_consumer = true;
                if (!L.rows_th && !L.dw5_groups && !L.dw3_cl) L.lds += (size_t)2 * cb * 4;
            }
            const int fwd_tiles = ceil_div(L.out_w, L.btw) * ceil_div(L.out_h, L.bth) * batch;      // (tiles of the separate backward-weights kernel)
            {   // weight-gradient partial rows: one per forward tile (separate kernels) or one per INPUT-space backward tile (fd_dw_bwd1: 16 columns x
                // up to 8 rows) -- sized for the larger count
                const long bwd_tiles = (long)ceil_div(L.in_w, 16) * ceil_div(L.in_h, 6) * batch;
                // (the register-window weight-gradient kernels: one row per 256 (column, channel group) pairs x strip of >= 4 rows)
                const long rows_tiles = (long)ceil_div((long)L.out_w * (d.cin / 4), 256) * ceil_div(L.out_h, 4) * batch;
                L.wp_elems = (size_t)std::max<long>(std::max<long>(fwd_tiles, bwd_tiles), rows_tiles) * d.ksize * d.ksize * d.cin;
            }
            break;
        }
        case FD_OP_PW:
            if (d.src < 0 || d.ksize != 1 || d.stride != 1 || d.cin % 4) FD_BAD("layer %d: bad pointwise", i);
            if (d.cout == 1) {
                if (d.skip >= 0) FD_BAD("layer %d: head with skip", i);
                L.head = true;
                L.out_h = d.upsample ? L.in_h / 2 : L.in_h;      // stored at the LOW resolution
                L.out_w = d.upsample ? L.in_w / 2 : L.in_w;
                L.grid = dim3(std::min(ceil_div((long)batch * L.out_h * L.out_w * 8, 256), 512));     // (grid-stride: one addition to the head's single statistics channel per workgroup)
                L.nblk = (int)L.grid.x;
                L.wp_elems = (size_t)ceil_div((long)batch * L.out_h * L.out_w, 32 * 16) * d.cin;
                if (!(tune & FD_TUNE_NO_CONSUMER_FINALIZE)) {
                    if (d.cin <= FD_HEAD_FIN_MAX && p->layers[d.src].nr_f <= FD_STAT_FIN_MAX_ROWS_ALL) p->layers[d.src].fin_by_consumer = true;     // its producer: finalised in fd_head_train
                    L.fin_by_consumer = FD_STAT_MAX_ROWS <= FD_STAT_FIN_MAX_ROWS_ALL;            // the head's own 1-channel BatchNorm (all 16 rows: one line each) could run in fd_head_apply_f32; with that many rows it keeps its launch
                }
            } else {
                if (d.upsample || d.skip >= 0) FD_BAD("layer %d: pointwise after upsample only as the 1-channel head", i);
                L.out_h = L.in_h; L.out_w = L.in_w;
                const long M = (long)batch * L.out_h * L.out_w;
                // 16-bit forward GEMM: 64 x 128 tiles when there are >= 128 output channels and that still leaves >= 200 workgroups
                L.pw_tn = (h16 && d.cout >= 128 && (long)ceil_div(M, 64) * ceil_div(d.cout, 128) >= 200) ? 2 : 1;
                const int bn = 64 * L.pw_tn;
                L.m_tiles = ceil_div(M, 64); L.n_tiles = ceil_div(d.cout, bn);
                L.grid = dim3((unsigned)((L.m_tiles + 7) / 8 * 8 * L.n_tiles));
                L.k64 = (d.cin + 63) / 64 * 64; L.n64 = (d.cout + 63) / 64 * 64;
                if (L.k64 > 1024 || d.cout > 1024) FD_BAD("layer %d: the train GEMMs hold per-channel tables of at most 1024 input / output channels (cin = %d, cout = %d)", i, d.cin, d.cout);
                // (the LDS-DMA ring holds min(3, K tiles) stages: the short-K units keep more workgroups resident)
                L.lds = h16 ? (size_t)std::min(FD_H16_STAGES, L.k64 / 64) * (64 + bn) * 128 + ((size_t)2 * L.k64 + 4 * bn) * 4
                            : (size_t)(std::min(3, ceil_div(d.cin, 32)) * 128 * 32 + 2 * ((d.cin + 31) / 32 * 32) + 256) * 4;
                L.nblk = L.m_tiles;
                if (!h16 && !(flags & FD_PLAN_NO_GEMM16) && d.cin % 32 == 0) {
                    // fp32 plans: the second-generation GEMM (fd_kernels_gemm16_f32.h, train mode) where ONE round of 256 workgroups covers the unit -- the same
                    // rule as the inference plan (choose_pw16): the 14 x 14 / 7 x 7 units at batch 32; its tiles are whole strides of
                    // up to 208 rows, so the unit leaves 32 ... 128 partial rows instead of 98 ... 392
                    // measured (B = 32, us, gemm16 train vs the 32x32x2 kernel): conv7.3 ... conv11.3 42.5 / 50.0, conv6.3 27.2 / 28.5, conv13.3 44.0 / 48.4, decode_conv1.1 31.0 / 33.3,
                    // conv12.3 26.6 / 26.5; with fewer than 512 output channels (<= 4 column tiles: decode_conv2.1 27.9 / 26.8, decode_conv3.1 29.4 / 28.9) it loses.  The BatchNorm +
                    // activation of the A fragments costs 6 of the 42.5 us (every fragment is read -- and transformed -- by the 4 column waves of its k-half; ablation: 36.6 without)
                    const bool force16 = (tune & FD_TUNE_FORCE_GEMM16) != 0;
                    const Pw16Cfg c16 = (d.cout >= 512 || force16) ? choose_pw16(M, d.cout, d.cin, force16) : Pw16Cfg();
                    if (c16.tm) {
                        L.pw16_tm = c16.tm; L.pw16_stride = c16.stride;
                        L.m_tiles = ceil_div(M, c16.stride); L.n_tiles = ceil_div(d.cout, 64);
                        L.grid = dim3((unsigned)((L.m_tiles + 7) / 8 * 8 * L.n_tiles));
                        L.lds = (size_t)FD_G16_STAGES * (c16.tm * 16 + 64) * 32 * 4 + (size_t)2 * d.cin * 4;      // ring + the producer's (scale, shift) table
                        L.nblk = L.m_tiles;
                    }
                }
                // the producer's BatchNorm is finalised by this GEMM's workgroups (fd_stat_table_all into the [2][K] table they keep in LDS anyway)
                if (!(tune & FD_TUNE_NO_CONSUMER_FINALIZE) && p->layers[d.src].nr_f <= FD_STAT_FIN_MAX_ROWS_ALL) p->layers[d.src].fin_by_consumer = true;
                {   // weight-gradient partials: splits x N x K (same split rule as launch_pw_bwd)
                    const int nt = ceil_div(d.cout, 64), kt = ceil_div(d.cin, 64);
                    int splits = std::max(1, std::min(ceil_div(h16 ? FD_WGRAD_TARGET_WGS_H16 : FD_WGRAD_TARGET_WGS_F32, (long)nt * kt), ceil_div(M, 256)));
                    const int rows = ceil_div(ceil_div(M, splits), 64) * 64;
                    splits = ceil_div(M, rows);
                    L.wp_elems = (size_t)splits * d.cout * d.cin;
                }
            }
            break;
        default: FD_BAD("layer %d: unknown op", i);
        }
        // (the 16-bit pointwise forward kernel raises its dynamic-LDS limit itself; the other train kernels stay within the default 64 KiB)
        if (L.lds > (((h16 && d.op == FD_OP_PW && !L.head) || d.op == FD_OP_DW || L.pw16_tm) ? 160 : 64) * 1024) FD_BAD("layer %d: LDS request %zu exceeds the limit", i, L.lds);
        // fd_nhwc (fd_device.h): within-image element offsets are 32-bit, formed with 24 x 24 bit multiplications
        if ((long)L.in_h * L.in_w >= (1L << 24) || d.cin >= (1 << 24) || d.cout >= (1 << 24) || (double)L.in_h * L.in_w * std::max(d.cin, d.cout) >= 4294967296.0)
            FD_BAD("layer %d: a %dx%d map with %d channels exceeds the kernels' 32-bit within-image addressing", i, L.in_h, L.in_w, std::max(d.cin, d.cout));
        L.M = (long)batch * L.out_h * L.out_w;
        // statistics rows (fd_device.h): a 64-bit bin holds 2^14 partials of < 2^48 each with room to spare and a unit adds at most one partial per 64 stored
        // pixels and channel -- larger batches would let the cross-row total wrap silently: rejected instead
        if (ceil_div(L.M, 64) > (1L << 14) * FD_STAT_MAX_ROWS) FD_BAD("layer %d: %ld stored pixels per channel exceed the statistics rows' headroom (%d rows x 2^14 partials of 64 pixels)", i, L.M, FD_STAT_MAX_ROWS);
        L.z_elems = (size_t)L.M * d.cout;
        L.n_stat = (double)L.M;
        L.n_unbiased = (L.head && d.upsample) ? 4.0 * (double)L.M : (double)L.M;
        L.z_off = off; off += align_up(L.z_elems * ((L.head) ? 4 : esz), 256);       // the 1-channel head stays fp32
        if (h16 && d.op == FD_OP_PW && !L.head) {
            L.wt_off = off; off += align_up((size_t)d.cout * L.k64 * esz, 256);
            L.wtt_off = off; off += align_up((size_t)d.cin * L.n64 * esz, 256);
        }
        L.st_off = off; off += align_up((size_t)4 * d.cout * 4, 256);
        L.coef_off = off; off += align_up((size_t)4 * d.cout * 4, 256);
        // statistics rows: reserved for the row count a producer of M / 64 workgroups per channel would take (the pointwise GEMMs' 64-row tiles: no kernel
        // of either direction has more workgroups per channel), used with the count the actual producer's workgroup number asks for
        L.nr_cap = L.head ? FD_STAT_MAX_ROWS : stat_nr(ceil_div(L.M, 64));       // (the head's single channel: one line per row, 512 workgroups)
        L.nr_f = L.head ? L.nr_cap : std::min(L.nr_cap, stat_nr(L.nblk));
        max_g = std::max(max_g, L.z_elems);
        L.wp_off = off; off += align_up(std::max(L.wp_elems, (size_t)1) * 4, 256);
    }
    TLayer &last = p->layers.back();
    if (!last.head || (last.d.upsample ? 2 * last.out_h : last.out_h) != height) FD_BAD("the last layer must be the 1-channel head producing [B,1,%d,%d]", height, width);
#undef FD_BAD
    // backward buffers: the gradient of unit i is consumed by unit i's own backward kernels right after unit i+1's, so two
    // ping-pong buffers suffice; skip sources get a private buffer for the decoder's contribution
    const size_t g_bytes = std::max(max_g * esz, p->layers.back().z_elems * 4);   // the head's gradient is fp32 in every plan
    size_t g0 = off; off += align_up(g_bytes, 256);
    size_t g1 = off; off += align_up(g_bytes, 256);
    for (int i = 0; i < n_layers; ++i) {
        TLayer &L = p->layers[i];
        L.g_off = (i & 1) ? g1 : g0;
        if (flags & FD_PLAN_KEEP_ACTIVATIONS) {   // private gradient buffers (layer-wise tests)
            L.g_off = off; off += align_up(L.z_elems * (L.head ? 4 : esz), 256);
            if (h16 && L.d.op == FD_OP_PW && !L.head) { L.dz_off = off; off += align_up(L.z_elems * esz, 256); }
        }
        if (L.skip_consumer >= 0) { L.sg_off = off; off += align_up(L.z_elems * esz, 256); }
    }
    // the statistics rows of every unit, forward then backward, in ONE region (zeroed by a single memset per step)
    p->stat_off = off;
    for (int i = 0; i < n_layers; ++i) {
        TLayer &L = p->layers[i];
        L.sf_off = off; off += align_up(stat_rows_bytes(L.nr_cap, L.d.cout), 256);
        L.sb_off = off; off += align_up(stat_rows_bytes(L.nr_cap, L.d.cout), 256);
    }
    p->stat_bytes = off - p->stat_off;
    // a unit whose BatchNorm is finalised inside its src-consumer's kernel publishes its table only when that kernel runs: a SKIP consumer that comes earlier in the
    // layer list would read the table before it is written (FastDepth's graph never does; the C ABI accepts arbitrary descriptors -- ADVICE r05)
    for (int i = 0; i < n_layers; ++i) {
        TLayer &U = p->layers[i];
        if (U.fin_by_consumer && U.skip_consumer >= 0 && U.skip_consumer < U.consumer) U.fin_by_consumer = false;
    }
    for (int i = 0; i < n_layers; ++i) p->layers[i].bwd_fin = bwd_fin_candidate(p, i);
    p->ws_bytes = off;
    *out_plan = p;
    return FD_OK;
}

void fd_train_plan_destroy(fd_train_plan *plan) { delete plan; }
size_t fd_train_plan_workspace_bytes(const fd_train_plan *plan) { return plan ? plan->ws_bytes : 0; }

int fd_train_plan_bind_workspace(fd_train_plan *plan, void *device_ptr, size_t bytes)
{
    if (!plan || !device_ptr) return fail(FD_ERR_INVALID, "null plan/workspace");
    if (bytes < plan->ws_bytes) return fail(FD_ERR_INVALID, "workspace too small: %zu < %zu", bytes, plan->ws_bytes);
    if (reinterpret_cast<uintptr_t>(device_ptr) % 256) return fail(FD_ERR_INVALID, "workspace must be 256-byte aligned");
    plan->ws = static_cast<unsigned char *>(device_ptr);
    plan->forward_done = false;
    return FD_OK;
}

int fd_train_forward(fd_train_plan *plan, const fd_layer_params *params, int32_t n_layers, float bn_eps, float bn_momentum,
                     const void *x_nchw, void *y, void *stream)
{
    if (!plan || !params || !x_nchw || !y) return fail(FD_ERR_INVALID, "null argument");
    if (!plan->ws) return fail(FD_ERR_STATE, "bind a workspace first");
    if (n_layers != (int)plan->layers.size()) return fail(FD_ERR_INVALID, "expected %zu layer parameter sets", plan->layers.size());
    return plan->dtype == FD_BF16 ? train_forward_t<fd_bf16>(plan, params, n_layers, bn_eps, bn_momentum, x_nchw, y, stream)
                                  : train_forward_t<float>(plan, params, n_layers, bn_eps, bn_momentum, x_nchw, y, stream);
}

int fd_train_plan_lds_rounding(const fd_train_plan *plan, int32_t layer)
{
    if (!plan || layer < 0 || layer >= (int)plan->layers.size()) return -1;
    return plan->layers[layer].lds_rounding;
}

int fd_train_plan_unit_kernels(const fd_train_plan *plan, int32_t layer)
{
    if (!plan || layer < 0 || layer >= (int)plan->layers.size()) return -1;
    const TLayer &L = plan->layers[layer];
    return (L.pw16_tm ? 1 : 0) | (L.fin_by_consumer ? 2 : 0) | (L.bwd_fin_rows > 0 ? 4 : 0) | (L.bwd_rows ? 8 : 0) | (L.dw5_groups ? 16 : 0) | (L.dw3_cl ? 32 : 0);
}

int fd_train_layer_tensor(const fd_train_plan *plan, int32_t layer, int32_t which, const void **device_ptr, int32_t *n, int32_t *h,
                          int32_t *w, int32_t *c)
{
    if (!plan || layer < 0 || layer >= (int)plan->layers.size() || which < 0 || which > 4) return fail(FD_ERR_INVALID, "bad layer index / selector");
    if (!plan->ws) return fail(FD_ERR_STATE, "no workspace bound");
    const TLayer &L = plan->layers[layer];
    if (which == 2) {   // the BatchNorm table [4][C]: scale, shift, mean, invstd
        if (device_ptr) *device_ptr = plan->ws + L.st_off;
        if (n) *n = 1;
        if (h) *h = 4;
        if (w) *w = 1;
        if (c) *c = L.d.cout;
        return FD_OK;
    }
    if (which == 3 && L.skip_consumer < 0) return fail(FD_ERR_INVALID, "layer %d is not a skip source", layer);
    if (which == 4 && !L.dz_off) return fail(FD_ERR_INVALID, "layer %d keeps no separate dz (16-bit pointwise units of a KEEP_ACTIVATIONS plan only)", layer);
    if (device_ptr) *device_ptr = plan->ws + (which == 0 ? L.z_off : which == 1 ? L.g_off : which == 3 ? L.sg_off : L.dz_off);
    if (n) *n = plan->B;
    if (h) *h = L.out_h;
    if (w) *w = L.out_w;
    if (c) *c = L.d.cout;
    return FD_OK;
}

}  // extern "C"

