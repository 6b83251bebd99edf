// fd_infer_launch.h -- kernel launches of the inference plan: one function per kernel family, template instance picked from the plan's per-layer record
// (translation unit fd_api.hip; split out of it in round 4 -- the plan code was a 1 160-line monolith)
#pragma once
namespace {

// ---- launches ------------------------------------------------------------------------------------
template <typename T, int ACT>
int launch_stem(const Layer &L, const float *x, const float *wp, const float *bias, T *y, int B, hipStream_t s)
{
    switch (L.chunk) {
    case 32: FD_LAUNCH((fd_stem3x3s2<T, ACT, 32>), L.grid, dim3(256), L.lds, s, x, wp, bias, y, B, L.in_h, L.in_w, L.d.cout); break;
    case 16: FD_LAUNCH((fd_stem3x3s2<T, ACT, 16>), L.grid, dim3(256), L.lds, s, x, wp, bias, y, B, L.in_h, L.in_w, L.d.cout); break;
    default: FD_LAUNCH((fd_stem3x3s2<T, ACT, 8>), L.grid, dim3(256), L.lds, s, x, wp, bias, y, B, L.in_h, L.in_w, L.d.cout); break;
    }
    return check_launch("fd_stem3x3s2");
}

template <typename T, int K, int S, int MODE, int ACT>
int launch_dw_inst(const Layer &L, const T *in, const T *skip, const float *wp, const float *bias, T *out, hipStream_t s)
{
    if constexpr (!std::is_same<T, float>::value) {
        if (L.dw_n == 8) {                                   // storage-typed LDS patches, 8 channels (16 bytes) per work-item
            FD_LAUNCH((fd_dwconv<T, K, S, MODE, ACT, 8>), L.grid, dim3(256), L.lds, s, in, skip, wp, bias, out,
                      L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.cbq, L.th, L.tw, L.tiles_x, L.csplit, L.pstr);
            return check_launch("fd_dwconv");
        }
    }
    FD_LAUNCH((fd_dwconv<T, K, S, MODE, ACT, 4>), L.grid, dim3(256), L.lds, s, in, skip, wp, bias, out,
                       L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.cbq, L.th, L.tw, L.tiles_x, L.csplit, L.pstr);
    return check_launch("fd_dwconv");
}

template <typename T, int ACT>
int launch_dw(const Layer &L, const T *in, const T *skip, const float *wp, const float *bias, T *out, hipStream_t s, const unsigned *wpk = nullptr)
{
    if (L.dw5_cl) {                                          // 16-bit plans: 5x5 on up2(low) + skip, row-walking pixel-pair kernel
        if constexpr (!std::is_same<T, float>::value) {
            if (L.dw5_cl == 64)
                FD_LAUNCH((fd_dw5_rows<T, ACT, 64>), L.grid, dim3(FD_DW5R_BLOCK), 0, s, in, skip, wpk, bias, out, L.in_h, L.in_w, L.d.cin, L.dw5_cbs, L.dw5_groups, L.dw5_bh);
            else
                FD_LAUNCH((fd_dw5_rows<T, ACT, 32>), L.grid, dim3(FD_DW5R_BLOCK), 0, s, in, skip, wpk, bias, out, L.in_h, L.in_w, L.d.cin, L.dw5_cbs, L.dw5_groups, L.dw5_bh);
            return check_launch("fd_dw5_rows");
        }
        return fail(FD_ERR_INVALID, "fd_dw5_rows is a 16-bit kernel");
    }
    if (L.dw_rows && L.dw_rows8) {
        if constexpr (!std::is_same<T, float>::value) {
            if (L.d.stride == 1)
                FD_LAUNCH((fd_dw3_rows8<T, 1, ACT>), L.grid, dim3(256), 0, s, in, wp, bias, out, L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.th);
            else
                FD_LAUNCH((fd_dw3_rows8<T, 2, ACT>), L.grid, dim3(256), 0, s, in, wp, bias, out, L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.th);
            return check_launch("fd_dw3_rows8");
        }
    }
    if (L.dw_rows) {
        if (L.d.stride == 1)
            FD_LAUNCH((fd_dw3_rows<T, 1, ACT>), L.grid, dim3(256), 0, s, in, wp, bias, out, L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.th);
        else
            FD_LAUNCH((fd_dw3_rows<T, 2, ACT>), L.grid, dim3(256), 0, s, in, wp, bias, out, L.in_h, L.in_w, L.out_h, L.out_w, L.d.cin, L.th);
        return check_launch("fd_dw3_rows");
    }
    const int key = L.d.ksize * 100 + L.d.stride * 10 + L.mode;
    switch (key) {
    case 310: return launch_dw_inst<T, 3, 1, 0, ACT>(L, in, skip, wp, bias, out, s);
    case 320: return launch_dw_inst<T, 3, 2, 0, ACT>(L, in, skip, wp, bias, out, s);
    case 510: return launch_dw_inst<T, 5, 1, 0, ACT>(L, in, skip, wp, bias, out, s);
    case 511: return launch_dw_inst<T, 5, 1, 1, ACT>(L, in, skip, wp, bias, out, s);
    case 512: return launch_dw_inst<T, 5, 1, 2, ACT>(L, in, skip, wp, bias, out, s);
    case 311: return launch_dw_inst<T, 3, 1, 1, ACT>(L, in, skip, wp, bias, out, s);
    case 312: return launch_dw_inst<T, 3, 1, 2, ACT>(L, in, skip, wp, bias, out, s);
    case 513: return launch_dw_inst<T, 5, 1, 3, ACT>(L, in, skip, wp, bias, out, s);
    }
    return fail(FD_ERR_INVALID, "depthwise k=%d stride=%d mode=%d has no kernel", L.d.ksize, L.d.stride, L.mode);
}

template <int ACT>
int launch_pw(const fd_plan *plan, const Layer &L, const float *A, const float *wp, const float *bias, float *out, long M, hipStream_t s)
{
    const int N = L.d.cout, K = L.d.cin;
    if (L.pw16_tm) {
        fd_dwfuse fz{};
        int fdw = 0;
        if (L.fuse_next_dw >= 0) {                            // the consuming depthwise layer runs in this kernel's epilogue
            const Layer &D = plan->layers[L.fuse_next_dw];
            fz.w = reinterpret_cast<const float *>(plan->ws + D.w_off); fz.b = reinterpret_cast<const float *>(plan->ws + D.b_off);
            fz.out = reinterpret_cast<float *>(plan->ws + D.out_off);
            fz.H = L.out_h; fz.W = L.out_w; fz.S = D.d.stride; fz.up = D.d.upsample;
            fz.hi = D.d.act == FD_ACT_RELU6 ? 6.0f : __builtin_inff();
            fz.store_pw = (plan->flags & FD_PLAN_KEEP_ACTIVATIONS) ? 1 : 0;
            fdw = D.d.ksize;
        }
#define FD_PW16_LAUNCH(TMV, FD_) \
        do { (void)hipFuncSetAttribute((const void *)fd_pw_gemm16_f32<TMV, FD_G16_STAGES, ACT, 0, FD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds); \
             FD_LAUNCH((fd_pw_gemm16_f32<TMV, FD_G16_STAGES, ACT, 0, FD_>), L.grid, dim3(512), L.lds, s, A, wp, bias, out, (int)M, N, K, L.w_pitch, L.pw16_stride, L.m_tiles, L.n_tiles, fz, fd_g16_train{}); } while (0)
#define FD_PW16_CASE(TMV) \
    case TMV: if (fdw == 3) FD_PW16_LAUNCH(TMV, 3); else if (fdw == 5) FD_PW16_LAUNCH(TMV, 5); else FD_PW16_LAUNCH(TMV, 0); break;
        switch (L.pw16_tm) {
            FD_PW16_CASE(13)
            FD_PW16_CASE(7)
            FD_PW16_CASE(4)
        default: return fail(FD_ERR_INVALID, "no gemm16 instance for TM=%d", L.pw16_tm);
        }
#undef FD_PW16_CASE
#undef FD_PW16_LAUNCH
        return check_launch("fd_pw_gemm16_f32");
    }
    const int key = L.pw.wgm * 1000 + L.pw.wgn * 100 + L.pw.tm * 10 + L.pw.tn;
#define FD_PW_CASE(a, b, c, d) \
    case a * 1000 + b * 100 + c * 10 + d: \
        if (L.lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)fd_pw_gemm_f32<a, b, c, d, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds); \
        FD_LAUNCH((fd_pw_gemm_f32<a, b, c, d, ACT>), L.grid, dim3(256), L.lds, s, A, wp, bias, out, (int)M, N, K, L.w_pitch, L.m_tiles, L.n_tiles); break;
    switch (key) {
        FD_PW_CASE(2, 2, 2, 2)
        FD_PW_CASE(2, 2, 2, 1)
        FD_PW_CASE(2, 2, 1, 2)
        FD_PW_CASE(2, 2, 1, 1)
        FD_PW_CASE(4, 1, 1, 1)
    default: return fail(FD_ERR_INVALID, "no pointwise tile %d", key);
    }
#undef FD_PW_CASE
    return check_launch("fd_pw_gemm_f32");
}



// depthwise + pointwise unit of a large map as one kernel (fd_kernels_dwpw_f32.h)
template <int ACT>
int launch_dwpw(const fd_plan *p, const Layer &L, float *out, float *y, hipStream_t s)
{
    const Layer &D = p->layers[L.fused_dw];
    const float *din = reinterpret_cast<const float *>(p->ws + p->layers[D.d.src].out_off);
    const float *dskip = D.d.skip >= 0 ? reinterpret_cast<const float *>(p->ws + p->layers[D.d.skip].out_off) : nullptr;
    const float *wdw = reinterpret_cast<const float *>(p->ws + D.w_off), *bdw = reinterpret_cast<const float *>(p->ws + D.b_off);
    const float *wp = reinterpret_cast<const float *>(p->ws + L.w_off), *bias = reinterpret_cast<const float *>(p->ws + L.b_off);
    fd_dwpw_head hd{};
    if (L.fuse_head >= 0) {
        const Layer &H = p->layers[L.fuse_head];
        hd.w = reinterpret_cast<const float *>(p->ws + H.w_off); hd.b = reinterpret_cast<const float *>(p->ws + H.b_off);
        hd.y = y; hd.act = H.d.act == FD_ACT_RELU6 ? 2 : (H.d.act == FD_ACT_RELU ? 1 : 0); hd.up = H.d.upsample;
    }
    const int key = D.d.ksize * 1000 + D.d.stride * 100 + D.mode * 10 + L.dp_nt + (L.fuse_head >= 0 ? 10000 : 0);
#define FD_DWPW_CASE(KSV, SV, MODEV, WMV, NTV, NLDV, HEADV)                                                                            \
    case KSV * 1000 + SV * 100 + MODEV * 10 + NTV + HEADV * 10000:                                                                     \
        (void)hipFuncSetAttribute((const void *)fd_dwpw_f32<KSV, SV, MODEV, ACT, WMV, NTV, NLDV, HEADV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds); \
        FD_LAUNCH((fd_dwpw_f32<KSV, SV, MODEV, ACT, WMV, NTV, NLDV, HEADV>), L.grid, dim3(512), L.lds, s, din, dskip, wdw, bdw, wp, bias, out, p->B, D.in_h, D.in_w, \
                  D.out_h, D.out_w, D.d.cin, L.w_pitch, L.d.cout, L.dp_th, L.dp_tw, L.dp_tiles_x, L.dp_tiles_x * ceil_div(D.out_h, L.dp_th), L.dp_xcd, L.pstr, hd);  \
        break;
    switch (key) {
        FD_DWPW_CASE(3, 1, 0, 4, 1, 6, 0)
        FD_DWPW_CASE(3, 1, 0, 4, 2, 6, 0)
        FD_DWPW_CASE(3, 1, 0, 4, 4, 6, 0)
        FD_DWPW_CASE(5, 1, 2, 4, 1, 8, 0)
        FD_DWPW_CASE(5, 1, 2, 4, 1, 8, 1)
        FD_DWPW_CASE(5, 1, 2, 4, 2, 8, 0)
        FD_DWPW_CASE(5, 1, 2, 4, 4, 8, 0)
        FD_DWPW_CASE(3, 2, 0, 2, 2, 10, 0)
        FD_DWPW_CASE(3, 2, 0, 2, 4, 10, 0)
    default: return fail(FD_ERR_INVALID, "no fd_dwpw_f32 instance %d", key);
    }
#undef FD_DWPW_CASE
    return check_launch("fd_dwpw_f32");
}

template <int ACT>
int launch_pw_t(const fd_plan *plan, const Layer &L, const float *A, const void *wp, const float *bias, float *out, long M, hipStream_t s, float * = nullptr)
{
    return launch_pw<ACT>(plan, L, A, static_cast<const float *>(wp), bias, out, M, s);
}
template <int ACT, typename T>
int launch_pw_t(const fd_plan *plan, const Layer &L, const T *A, const void *wp, const float *bias, T *out, long M, hipStream_t s, float *y = nullptr)
{
    const int K = L.d.cin;
    if (L.pw16_tm) {                                         // fd_pw_gemm16_h16: whole frames per workgroup, optionally with the consuming depthwise layer
        fd_dwfuse fz{};
        int fdw = 0;
        if (L.fuse_next_dw >= 0) {
            const Layer &D = plan->layers[L.fuse_next_dw];
            fz.w = reinterpret_cast<const float *>(plan->ws + D.w_off); fz.b = reinterpret_cast<const float *>(plan->ws + D.b_off);
            fz.out = reinterpret_cast<float *>(plan->ws + D.out_off);        // (T-typed: the kernel casts)
            fz.H = L.out_h; fz.W = L.out_w; fz.S = D.d.stride; fz.up = D.d.upsample;
            fz.hi = D.d.act == FD_ACT_RELU6 ? 6.0f : __builtin_inff();
            fz.store_pw = (plan->flags & FD_PLAN_KEEP_ACTIVATIONS) ? 1 : 0;
            fdw = D.d.ksize;
        }
#define FD_PW16H_LAUNCH(TMV, FD_) \
        do { (void)hipFuncSetAttribute((const void *)fd_pw_gemm16_h16<T, TMV, 4, ACT, FD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds); \
             FD_LAUNCH((fd_pw_gemm16_h16<T, TMV, 4, ACT, FD_>), L.grid, dim3(512), L.lds, s, A, static_cast<const T *>(wp), bias, out, (int)M, L.d.cout, K, L.w_pitch, L.pw16_stride, L.m_tiles, L.n_tiles, fz, 0 /* ablation bits: tools/microbench only */); } while (0)
#define FD_PW16H_CASE(TMV) \
    case TMV: if (fdw == 3) FD_PW16H_LAUNCH(TMV, 3); else if (fdw == 5) FD_PW16H_LAUNCH(TMV, 5); else FD_PW16H_LAUNCH(TMV, 0); break;
        switch (L.pw16_tm) {
            FD_PW16H_CASE(13)
            FD_PW16H_CASE(7)
            FD_PW16H_CASE(4)
        default: return fail(FD_ERR_INVALID, "no 16-bit gemm16 instance for TM=%d", L.pw16_tm);
        }
#undef FD_PW16H_CASE
#undef FD_PW16H_LAUNCH
        return check_launch("fd_pw_gemm16_h16");
    }
    if (L.fuse_head >= 0) {                                  // the network head on this GEMM's output tile (cout <= 32): one launch, no intermediate tensor
        const Layer &H = plan->layers[L.fuse_head];
        fd_pw_head hd{};
        hd.w = reinterpret_cast<const float *>(plan->ws + H.w_off); hd.b = reinterpret_cast<const float *>(plan->ws + H.b_off);
        hd.y = y; hd.act = H.d.act == FD_ACT_RELU6 ? 2 : (H.d.act == FD_ACT_RELU ? 1 : 0); hd.up = H.d.upsample; hd.h = L.out_h; hd.w_ = L.out_w;
        FD_LAUNCH((fd_pw_gemm_head_h16<T, ACT>), L.grid, dim3(256), L.lds, s, A, static_cast<const T *>(wp), bias, (int)M, L.d.cout, K, (K + 63) / 64 * 64,
                  L.m_tiles, L.n_tiles, hd);
        return check_launch("fd_pw_gemm_head_h16");
    }
    FD_LAUNCH((fd_pw_gemm_h16<T, ACT>), L.grid, dim3(256), L.lds, s, A, static_cast<const T *>(wp), bias, out, (int)M, L.d.cout, K, (K + 63) / 64 * 64,
              L.m_tiles, L.n_tiles);
    return check_launch("fd_pw_gemm_h16");
}

template <typename T, int ACT>
int launch_layer(const fd_plan *p, const Layer &L, const float *x, float *y, hipStream_t s)
{
    const void *wp = p->ws + L.w_off;
    const float *wpf = reinterpret_cast<const float *>(p->ws + L.w_off);
    const float *bias = reinterpret_cast<const float *>(p->ws + L.b_off);
    T *out = reinterpret_cast<T *>(p->ws + L.out_off);
    const T *in = L.d.src < 0 ? nullptr : reinterpret_cast<const T *>(p->ws + p->layers[L.d.src].out_off);
    const T *skip = L.d.skip >= 0 ? reinterpret_cast<const T *>(p->ws + p->layers[L.d.skip].out_off) : nullptr;
    switch (L.d.op) {
    case FD_OP_STEM: return launch_stem<T, ACT>(L, x, wpf, bias, out, p->B, s);
    case FD_OP_DW: return launch_dw<T, ACT>(L, in, skip, wpf, bias, out, s, L.dw5_cl ? reinterpret_cast<const unsigned *>(p->ws + L.wpk_off) : nullptr);
    case FD_OP_PW:
        if (L.head) {
            const int h = L.d.upsample ? L.in_h / 2 : L.in_h, w = L.d.upsample ? L.in_w / 2 : L.in_w;
            const long npix = (long)p->B * h * w;
            FD_LAUNCH((fd_head_pw1<T, ACT>), L.grid, dim3(256), 0, s, in, wpf, bias, y, npix, h, w, L.d.cin, L.d.upsample);
            return check_launch("fd_head_pw1");
        }
        if (L.dwpw) {
            if constexpr (std::is_same<T, float>::value) return launch_dwpw<ACT>(p, L, out, y, s);
            else return fail(FD_ERR_INVALID, "fused units are fp32 only");
        }
        return launch_pw_t<ACT>(p, L, in, wp, bias, out, (long)p->B * L.out_h * L.out_w, s, y);
    }
    return fail(FD_ERR_INVALID, "bad op");
}

template <typename T>
int run_layer_t(fd_plan *plan, const Layer &L, const float *x, float *out, hipStream_t s)
{
    switch (L.d.act) {
    case FD_ACT_RELU: return launch_layer<T, FD_ACT_RELU_>(plan, L, x, out, s);
    case FD_ACT_RELU6: return launch_layer<T, FD_ACT_RELU6_>(plan, L, x, out, s);
    default: return launch_layer<T, FD_ACT_NONE_>(plan, L, x, out, s);
    }
}
int run_layer(fd_plan *plan, const Layer &L, const float *x, float *out, hipStream_t s)
{
    switch (plan->dtype) {
    case FD_F16: return run_layer_t<fd_half>(plan, L, x, out, s);
    case FD_BF16: return run_layer_t<fd_bf16>(plan, L, x, out, s);
    default: return run_layer_t<float>(plan, L, x, out, s);
    }
}

}  // namespace
