// fd_plan_build.h -- the passes of fd_plan_create: per-layer geometry and validation, the three fusion passes, the activation arena, bookkeeping / descriptions
// (translation unit fd_api.hip; split out of it in round 4 -- the plan code was a 1 160-line monolith)
#pragma once
namespace {

// pass 1: validates every layer against its producer, picks kernel / tiling / LDS per layer, lays out the packed weights (woff_out: their total bytes)
int plan_layers(fd_plan *p, const fd_layer_desc *layers, size_t *woff_out)
{
    const size_t esz = p->dtype == FD_F32 ? 4 : 2;   // activation / pointwise-weight element size
    const int n_layers = (int)p->layers.size(), batch = p->B, height = p->H, width = p->W, dtype = p->dtype;
    const uint32_t flags = p->flags, tune = p->tune;
    (void)n_layers; (void)batch; (void)height; (void)width; (void)dtype; (void)flags; (void)tune;
    size_t woff = 0;
    for (int i = 0; i < n_layers; ++i) {
        Layer &L = p->layers[i];
        L.d = layers[i];
        const fd_layer_desc &d = L.d;
#define FD_BAD(...) do { return fail(FD_ERR_INVALID, __VA_ARGS__); } while (0)
        if (d.src >= i || d.skip >= i) FD_BAD("layer %d: src/skip must reference earlier layers", i);
        if (d.act < FD_ACT_NONE || d.act > FD_ACT_RELU6) FD_BAD("layer %d: bad activation", i);
        if (d.cin <= 0 || d.cout <= 0) FD_BAD("layer %d: bad channel counts", i);
        int src_h, src_w, src_c;
        if (d.src < 0) { src_h = height; src_w = width; src_c = 3; }
        else { const Layer &S = p->layers[d.src]; src_h = S.out_h; src_w = S.out_w; src_c = S.d.cout; }
        const bool concat = d.concat != 0;
        if (concat && (d.skip < 0 || d.op != FD_OP_DW || !d.upsample)) FD_BAD("layer %d: concat needs an upsampled depthwise consumer with a skip tensor", i);
        if (concat) {
            L.csplit = src_c;
            if (src_c + p->layers[d.skip].d.cout != d.cin || src_c % 4) FD_BAD("layer %d: concat of %d + %d channels does not give cin %d", i, src_c, p->layers[d.skip].d.cout, d.cin);
        } else if (src_c != d.cin) FD_BAD("layer %d: cin %d != producer channels %d", i, d.cin, src_c);
        L.in_h = d.upsample ? 2 * src_h : src_h;
        L.in_w = d.upsample ? 2 * src_w : src_w;
        if (d.skip >= 0) {
            const Layer &S = p->layers[d.skip];
            if (!d.upsample) FD_BAD("layer %d: skip without upsample is not part of this path", i);
            if (S.out_h != L.in_h || S.out_w != L.in_w || (!concat && S.d.cout != d.cin))
                FD_BAD("layer %d: skip tensor %dx%dx%d does not match input %dx%dx%d", i, S.out_h, S.out_w, S.d.cout, L.in_h, L.in_w, d.cin);
        }
        switch (d.op) {
        case FD_OP_STEM:
            if (d.src != -1 || d.cin != 3 || d.ksize != 3 || d.stride != 2 || d.upsample || d.skip >= 0 || d.cout % 8)
                FD_BAD("layer %d: stem must be 3->8k channels, 3x3 stride 2 on the network input", i);
            L.out_h = L.in_h / 2; L.out_w = L.in_w / 2;
            L.chunk = d.cout % 32 == 0 ? 32 : (d.cout % 16 == 0 ? 16 : 8);
            if (d.cout > 64) FD_BAD("layer %d: the stem supports at most 64 output channels", i);
            {   // LDS: the zero-padded band of input rows under 256 consecutive output pixels (3 planes), later reused as the output staging tiles
                const int nrows = 2 * ceil_div(255, L.out_w) + 3;
                L.lds = std::max((size_t)3 * nrows * (L.in_w + 8) * 4, (size_t)4 * 64 * 36 * 4);
            }
            L.grid = dim3(ceil_div((long)L.out_h * L.out_w, 256), batch);
            L.w_bytes = (size_t)27 * d.cout * 4; L.w_elems = (size_t)27 * d.cout;
            break;
        case FD_OP_DW: {
            if (d.src < 0 || d.cin != d.cout || (d.ksize != 3 && d.ksize != 5) || (d.stride != 1 && d.stride != 2) || d.cin % 4)
                FD_BAD("layer %d: depthwise needs cin==cout (multiple of 4), k in {3,5}, stride in {1,2}", i);
            if (d.stride == 2 && (L.in_h % 2 || L.in_w % 2)) FD_BAD("layer %d: stride-2 depthwise on odd input", i);
            L.mode = d.upsample ? (d.skip >= 0 ? (concat ? 3 : 2) : 1) : 0;
            L.out_h = L.in_h / d.stride; L.out_w = L.in_w / d.stride;
            if (d.ksize == 3 && L.mode == 0) {
                // register-window kernel: pick the row-strip height so that the grid has >= ~4 workgroups per CU when it can
                L.dw_rows = true;
                L.dw_rows8 = dtype != FD_F32 && d.cin % 8 == 0 && d.stride == 1 && !(flags & FD_PLAN_NO_ROWS8);   // 16-bit storage, stride 1: eight channels (16 bytes) per work-item (measured: conv3.0 15.2 -> 13.9 us, pruned conv7-11 -1 ... -2 us each; the stride-2 layers lose: 20.5 -> 23.8)
                const int gx = ceil_div((long)L.out_w * (d.cin / (L.dw_rows8 ? 8 : 4)), 256);
                int th = L.out_h;
                while (th > 4 && (long)gx * ceil_div(L.out_h, th) * batch < 1024) th = (th + 1) / 2;
                L.th = th;
                L.grid = dim3(gx, ceil_div(L.out_h, th), batch);
                L.lds = 0;
                L.w_bytes = (size_t)9 * d.cin * 4; L.w_elems = (size_t)9 * d.cin;
                break;
            }
            // 16-bit plans, 5x5 on up2(low) + skip (decode_conv3 / 4 / 5): the row-walking pixel-pair kernel (fd_kernels_dw5p.h).  Measured at batch 32, fp16,
            // us (tools/microbench/dw5pairs.hip): decode_conv5.0 42.6 -> 34, decode_conv4.0 26.4 -> 21, decode_conv3.0 15.3 -> 13; pruned widths at batch 64:
            // 81 -> 56, 49 -> 33, 26 -> 20.  Bands of 14 rows (balanced, even) measured best or equal on all three maps; the 128-channel wave form only
            // where the strip count is odd (7 strips of a 28-pixel row: no half-idle wave)
            if (dtype != FD_F32 && d.ksize == 5 && d.stride == 1 && L.mode == 2 && d.cin % 8 == 0 && L.out_w % 4 == 0 && L.in_h % 2 == 0 &&
                (double)L.in_h * L.in_w * d.cin * 2.0 < 2147483648.0 && !(tune & FD_TUNE_NO_DW5_ROWS)) {
                const int strips = L.out_w / 4;
                L.dw5_cl = (strips % 2 == 1 && d.cin >= 128) ? 64 : 32;
                const int cblocks = ceil_div(d.cin, 2 * L.dw5_cl);
                L.dw5_cbs = ceil_div(ceil_div(d.cin, cblocks), 8) * 8;
                L.dw5_groups = ceil_div(strips, 64 / L.dw5_cl);
                const int bands = std::max(1, (L.out_h + 7) / 14);
                L.dw5_bh = ceil_div(ceil_div(L.out_h, bands), 2) * 2;
                L.grid = dim3(ceil_div((long)L.dw5_groups * ceil_div(L.out_h, L.dw5_bh), FD_DW5R_BLOCK / 64), cblocks, batch);
                L.lds = 0;
                L.w_elems = (size_t)25 * d.cin;
                L.wpk_off = align_up((size_t)25 * d.cin * 4, 256);      // (relative to w_off; made absolute below)
                L.w_bytes = L.wpk_off + (size_t)30 * d.cin * 4;
                break;
            }
            // 16-bit plans: 8 channels (16 bytes) per work-item and patches kept in the storage type -- a 64-channel block has the LDS footprint
            // (and the instruction count) of the 32-channel fp32 block; FD_TUNE_NO_DW_H8 keeps the 4-channel / fp32-patch form for A/B runs
            // Measured (fp16, batch 32, us, 8-channel vs 4-channel form): decode_conv5.0 46.5 vs 49.4, decode_conv4.0 28.3 vs 29.4 -- but decode_conv3.0
            // 18.4 vs 16.7, decode_conv1.0 11.0 vs 7.8: half the workgroups only pays where many rounds of them remain, so the plan takes it for
            // the 5x5 units on maps of >= 56 x 56 with whole 64-channel blocks (FD_TUNE_FORCE_DW_H8: wherever eligible -- tests)
            const bool h8_ok = dtype != FD_F32 && d.cin % 8 == 0 && (!concat || L.csplit % 8 == 0) && !(tune & FD_TUNE_NO_DW_H8);
            const bool h8 = h8_ok && ((tune & FD_TUNE_FORCE_DW_H8) || (d.ksize == 5 && d.cin % 64 == 0 && (long)L.out_h * L.out_w >= 56 * 56));
            L.dw_n = h8 ? 8 : 4;
            int cb = d.cin >= 32 ? 32 : (d.cin >= 16 ? 16 : (d.cin >= 8 ? 8 : 4));
            if (h8 && d.cin >= 64 && ceil_div(d.cin, 64) * 64 <= ceil_div(d.cin, 32) * 32) cb = 64;   // (pruned widths: the block size that pads the channel count least)
            L.cbq = ilog2(cb / L.dw_n);
            const int tmax_w = d.stride == 2 ? 8 : 16, tmax_h = d.ksize == 5 ? 7 : 8;   // 8x16 (5x5: 7x16, conflict-free pitch 40) outputs x 32 channels: < 40 KB LDS -> 4 workgroups per CU
            L.tw = std::min((L.out_w + 3) / 4 * 4, tmax_w);
            L.th = std::min(L.out_h, tmax_h);
            L.tiles_x = ceil_div(L.out_w, L.tw); L.tiles_y = ceil_div(L.out_h, L.th);
            const int th_in = (L.th - 1) * d.stride + d.ksize, tw_in = (L.tw - 1) * d.stride + d.ksize;
            // (h8: a lane's 8 channels are 4 dwords, so the bank replay is that of cb / 2 fp32 channels; the pitch comes back in dwords)
            L.pstr = h8 ? 2 * pick_patch_pitch(cb / 2, L.tw, tw_in, d.stride) : pick_patch_pitch(cb, L.tw, tw_in, d.stride);
            L.lds = align_up((size_t)th_in * tw_in * L.pstr * (h8 ? 2 : 4), 16) + ((size_t)d.ksize * d.ksize * cb + cb) * 4;
            L.grid = dim3(L.tiles_x * L.tiles_y, ceil_div(d.cin, cb), batch);
            L.w_bytes = (size_t)d.ksize * d.ksize * d.cin * 4; L.w_elems = (size_t)d.ksize * d.ksize * d.cin;
            break;
        }
        case FD_OP_PW:
            if (d.src < 0 || d.ksize != 1 || d.stride != 1 || d.cin % 4) FD_BAD("layer %d: pointwise needs k=1 stride=1 cin%%4==0", i);
            L.out_h = L.in_h; L.out_w = L.in_w;
            L.w_bytes = (size_t)d.cin * d.cout * 4;     // the 1-channel head keeps fp32 weights
            L.w_elems = (size_t)d.cin * d.cout;
            if (d.cout == 1) {
                if (d.skip >= 0) FD_BAD("layer %d: head with skip is not part of this path", i);
                L.head = true;
                const long npix = (long)batch * (L.in_h >> (d.upsample ? 1 : 0)) * (L.in_w >> (d.upsample ? 1 : 0));
                L.grid = dim3(ceil_div(npix * 8, 256));
            } else {
                if (d.upsample || d.skip >= 0) FD_BAD("layer %d: pointwise after upsample is only supported for the 1-channel head", i);
                const long M = (long)batch * L.out_h * L.out_w;
                if (dtype != FD_F32 && d.cin % 8) FD_BAD("layer %d: 16-bit pointwise needs cin %% 8 == 0", i);
                L.w_pitch = dtype == FD_F32 ? (d.cin + 31) / 32 * 32 : (d.cin + 63) / 64 * 64;   // rows zero-padded to a multiple of BK
                L.w_bytes = (size_t)d.cout * L.w_pitch * esz;
                L.pw = dtype == FD_F32 ? choose_pw(M, d.cout) : PwCfg{2, 2, 1, 1};
                L.lds = pw_lds_bytes(L.pw);
                // 16-bit kernel: a reduction of one or two K tiles never touches the ring's later stages -- not requested, so that more workgroups
                // of the short-K units (conv1.3, conv2.3, decode_conv5.1: all head and tail) are resident per CU (its epilogue tile needs 10 KiB)
                if (dtype != FD_F32) L.lds = (size_t)std::min(FD_H16_STAGES, ceil_div(d.cin, 64)) * 128 * 128;
                L.m_tiles = ceil_div(M, L.pw.wgm * L.pw.tm * 32);
                L.n_tiles = ceil_div(d.cout, L.pw.wgn * L.pw.tn * 32);
                L.grid = dim3((unsigned)((L.m_tiles + 7) / 8 * 8 * L.n_tiles));   // 1-D, XCD-aware mapping inside the kernel
                if ((dtype == FD_F32 || (tune & FD_TUNE_FORCE_GEMM16)) && !(flags & FD_PLAN_NO_GEMM16)) {
                    // (16-bit plans take fd_pw_gemm16_h16 where a depthwise consumer fuses behind it -- decided in the fusion pass below --
                    // or, with FD_TUNE_FORCE_GEMM16, everywhere: tests)
                    const Pw16Cfg c16 = choose_pw16(M, d.cout, d.cin, (tune & FD_TUNE_FORCE_GEMM16) != 0);
                    if (c16.tm) {
                        L.pw16_tm = c16.tm; L.pw16_stride = c16.stride;
                        L.lds = (size_t)(dtype == FD_F32 ? FD_G16_STAGES : 4) * (c16.tm * 16 + 64) * 32 * 4;   // (128-byte rows in both kernels; the 16-bit one runs a 4-stage ring)
                        L.m_tiles = ceil_div(M, c16.stride); L.n_tiles = ceil_div(d.cout, 64);
                        L.grid = dim3((unsigned)((L.m_tiles + 7) / 8 * 8 * L.n_tiles));
                    }
                }
            }
            break;
        default: FD_BAD("layer %d: unknown op %d", i, d.op);
        }
        if (L.lds > 160 * 1024) FD_BAD("layer %d: LDS request %zu exceeds 160 KiB", i, L.lds);
        // fd_nhwc (fd_device.h): within-image element offsets are 32-bit, formed with 24 x 24 bit multiplications
        if ((long)L.in_h * L.in_w >= (1L << 24) || d.cin >= (1 << 24) || d.cout >= (1 << 24) || (double)L.in_h * L.in_w * std::max(d.cin, d.cout) >= 4294967296.0)
            FD_BAD("layer %d: a %dx%d map with %d channels exceeds the kernels' 32-bit within-image addressing", i, L.in_h, L.in_w, std::max(d.cin, d.cout));
        L.w_off = woff; woff += align_up(L.w_bytes, 256);
        if (L.dw5_cl) L.wpk_off += L.w_off;
        L.b_off = woff; woff += align_up((size_t)d.cout * 4, 256);
        L.out_bytes = align_up((size_t)batch * L.out_h * L.out_w * d.cout * esz, 256);
        L.pw_packed_t = (d.op == FD_OP_PW && !L.head && dtype != FD_F32);
    }
    Layer &last = p->layers.back();
    if (last.d.cout != 1 || last.out_h != height || last.out_w != width)
        FD_BAD("the last layer must produce the [B,1,%d,%d] network output (got %dx%dx%d)", height, width, last.out_h, last.out_w, last.d.cout);
#undef FD_BAD
    last.to_output = true;
    p->weights_bytes = woff;

    *woff_out = woff;
    return FD_OK;
}

// pass 2: depthwise consumers evaluated in the epilogue of a whole-frame pointwise GEMM
void plan_fuse_epilogues(fd_plan *p)
{
    const int n_layers = (int)p->layers.size(), batch = p->B, height = p->H, width = p->W, dtype = p->dtype;
    const uint32_t flags = p->flags, tune = p->tune;
    (void)n_layers; (void)batch; (void)height; (void)width; (void)dtype; (void)flags; (void)tune;
    // ---- fusion: a depthwise layer whose producer is a gemm16 pointwise layer with WHOLE frames per workgroup is evaluated in that
    // kernel's epilogue (fd_pw_gemm16_f32<..., FDW>): depthwise convolution is per channel, so a workgroup that holds 64 channels of a
    // few complete frames holds everything the consumer needs for those channels and frames.
    if (!(flags & FD_PLAN_NO_EPILOGUE_FUSION)) {
        std::vector<int> readers(n_layers, 0);
        for (int i = 0; i < n_layers; ++i) {
            if (p->layers[i].d.src >= 0) ++readers[p->layers[i].d.src];
            if (p->layers[i].d.skip >= 0) ++readers[p->layers[i].d.skip];
        }
        for (int j = 1; j < n_layers; ++j) {
            Layer &D = p->layers[j];
            if (D.d.op != FD_OP_DW || D.d.src < 0 || D.d.skip >= 0 || D.d.concat) continue;
            if (D.d.act == FD_ACT_NONE) continue;                      // the epilogue's depthwise stage always clamps at 0 (ReLU / ReLU6)
            Layer &Pw = p->layers[D.d.src];
            if (Pw.d.op != FD_OP_PW || Pw.head || readers[D.d.src] != 1) continue;       // the pointwise output must have no other reader (skip sources keep their tensor)
            const int hw = Pw.out_h * Pw.out_w;
            int tm = Pw.pw16_tm, stride = Pw.pw16_stride, m_tiles = Pw.m_tiles, n_tiles = Pw.n_tiles;
            size_t lds = Pw.lds;
            // (measured at batch 32 / 64, fp16: the fused launch takes 11-12.6 us where the pointwise GEMM + the depthwise launch took 18 on the 14x14
            // maps; on the 7x7 maps (6.6 + 5.4 us unfused) and where the grid needs a second round of workgroups (pruned plan at batch 64) it is
            // no faster, so those keep the first-generation kernels unless FD_TUNE_FORCE_EPILOGUE_FUSION asks for every eligible pair: tests)
            const bool want_all = (tune & FD_TUNE_FORCE_EPILOGUE_FUSION) != 0;
            const bool h16_pick = dtype != FD_F32 && !(flags & FD_PLAN_NO_GEMM16) && !(tune & FD_TUNE_FORCE_GEMM16) && hw <= 208 &&
                                  (want_all || (hw >= 128 && (long)batch * ceil_div(Pw.d.cout, 64) <= 272));
            if (h16_pick) {
                // 16-bit plans: a pointwise layer of a small map (a frame is at most 13 row tiles) followed by a fusable depthwise layer moves to
                // fd_pw_gemm16_h16 with WHOLE frames per workgroup -- as many (4, 2, 1) as still leave a full round of workgroups
                n_tiles = ceil_div(Pw.d.cout, 64);
                int f = 1;
                for (int cand : {4, 2}) if (cand * hw <= 208 && (long)ceil_div(batch, cand) * n_tiles >= 256) { f = cand; break; }
                stride = f * hw; tm = stride <= 64 ? 4 : (stride <= 112 ? 7 : 13);
                lds = (size_t)4 * (tm * 16 + 64) * 128;
                m_tiles = ceil_div((long)batch * hw, stride);
            }
            if (!tm) continue;
            if (stride % hw || D.d.cin % 4) continue;                 // whole frames per workgroup
            if (D.d.upsample && D.d.stride != 1) continue;
            {   // the zero-bordered frame image (+ one dump row) must fit the kernel's LDS ring
                const int P = D.d.upsample ? (D.d.ksize / 2 + 1) / 2 : D.d.ksize / 2;
                const long img_rows = (long)(stride / hw) * (Pw.out_h + 2 * P) * (Pw.out_w + 2 * P) + 1;
                if ((size_t)img_rows * 68 * 4 > lds) continue;
            }
            if (h16_pick) {
                Pw.pw16_tm = tm; Pw.pw16_stride = stride; Pw.lds = lds; Pw.m_tiles = m_tiles; Pw.n_tiles = n_tiles;
                Pw.grid = dim3((unsigned)((m_tiles + 7) / 8 * 8 * n_tiles));
            }
            Pw.fuse_next_dw = j;
            D.fused_into = D.d.src;
        }
    }

    // ---- fusion: depthwise -> pointwise units of the LARGE maps become one kernel (fd_dwpw_f32): the depthwise output (up to 103 MB at
}

// pass 3: depthwise + pointwise units of the large maps as one persistent kernel (fd_dwpw_f32), optionally with the network head on its accumulators
void plan_fuse_units(fd_plan *p)
{
    const int n_layers = (int)p->layers.size(), batch = p->B, height = p->H, width = p->W, dtype = p->dtype;
    const uint32_t flags = p->flags, tune = p->tune;
    (void)n_layers; (void)batch; (void)height; (void)width; (void)dtype; (void)flags; (void)tune;
    // batch 32) never makes its HBM round trip.  Applies where a workgroup can own a pixel tile with ALL output channels (N <= 128, or
    // <= 256 behind a stride-2 depthwise) -- on the small maps the GEMM is the cost and the opposite fusion (above) is used.
    if (dtype == FD_F32 && !(flags & FD_PLAN_NO_UNIT_FUSION) && (!(flags & FD_PLAN_KEEP_ACTIVATIONS) || (tune & FD_TUNE_FORCE_UNIT_FUSION))) {
        std::vector<int> readers(n_layers, 0);
        for (int i = 0; i < n_layers; ++i) {
            if (p->layers[i].d.src >= 0) ++readers[p->layers[i].d.src];
            if (p->layers[i].d.skip >= 0) ++readers[p->layers[i].d.skip];
        }
        for (int i = 0; i + 1 < n_layers; ++i) {
            Layer &D = p->layers[i], &Pw = p->layers[i + 1];
            if (D.d.op != FD_OP_DW || D.fused_into >= 0 || D.skipped || D.d.src < 0 || readers[i] != 1) continue;
            if (Pw.d.op != FD_OP_PW || Pw.head || Pw.d.src != i || Pw.d.upsample || Pw.fuse_next_dw >= 0 || Pw.to_output) continue;
            const int C = D.d.cin, N = Pw.d.cout, KS = D.d.ksize, S = D.d.stride;
            if (D.d.act != Pw.d.act || D.d.act == FD_ACT_NONE || C % 32 || C > 256 || N % 32) continue;
            // the kernel addresses its tensors with 32-bit element / byte offsets
            if ((double)batch * D.in_h * D.in_w * C >= 2147483648.0 || (double)batch * D.out_h * D.out_w * N * 4.0 >= 4294967296.0) continue;
            int wm = 0, nld = 0;
            if (KS == 3 && S == 1 && D.mode == 0) { wm = 4; nld = 6; }
            else if (KS == 5 && S == 1 && D.mode == 2) { wm = 4; nld = 8; }
            else if (KS == 3 && S == 2 && D.mode == 0) { wm = 2; nld = 10; }
            else continue;
            const int wn = 4 / wm, nt = N / 32 / wn;
            if (nt * wn * 32 != N || !(nt == 1 || nt == 2 || nt == 4) || (wm == 2 && nt == 1)) continue;
            // Where it pays (measured in the batch-32 plan): units with <= 64 depthwise channels on maps of >= 28x28
            // pixels (conv1: 52.7 -> 39 us, conv2: 50.4 -> 42 us, decode_conv5: 87 -> 80 us).  The 128-channel units are bound by the fp32
            // MFMAs (conv3) or the 5x5 taps' LDS reads (decode_conv4) and lose 7 us each; small maps are launch-bound and use the
            // GEMM-epilogue fusion above.
            if (!(tune & FD_TUNE_FORCE_UNIT_FUSION) && (C > 64 || D.out_h * D.out_w < 28 * 28)) continue;
            // the whole weight matrix, the taps and two A tiles stay in LDS next to the patch
            const size_t lds = ((size_t)nld * 32 * 36 + 2 * 32 * wm * 32 + (size_t)N * C + (size_t)KS * KS * C + C) * 4;
            if ((C / 32) & (C / 32 - 1) || lds > 160 * 1024) continue;
            // pixel tile: TH x TW <= 32*wm outputs, TW a power of two >= 4, patch <= 32*nld pixels; fewest staged patch pixels + MFMA rows wins
            long best = -1; int bth = 0, btw = 0;
            for (int tws = 2; tws <= 5; ++tws) {
                const int tw = 1 << tws;
                if (tw > 32 * wm || (tw > 4 && tw >= 2 * D.out_w)) continue;
                for (int th = 1; th * tw <= 32 * wm && th <= D.out_h; ++th) {
                    const int ph = (th - 1) * S + KS, pw = (tw - 1) * S + KS;
                    if (ph * pw > 32 * nld) continue;
                    const long tiles = (long)ceil_div(D.out_h, th) * ceil_div(D.out_w, tw);
                    const long cost = tiles * (ph * pw + 32 * wm);
                    if (best < 0 || cost < best) { best = cost; bth = th; btw = tws; }
                }
            }
            if (best < 0) continue;
            D.skipped = true;
            Pw.fused_dw = i; Pw.dwpw = true; Pw.pw16_tm = 0;
            Pw.dp_th = bth; Pw.dp_tw = btw; Pw.dp_tiles_x = ceil_div(D.out_w, 1 << btw); Pw.dp_wm = wm; Pw.dp_nt = nt; Pw.dp_nld = nld;
            const long tiles = (long)Pw.dp_tiles_x * ceil_div(D.out_h, bth) * batch;
            Pw.dp_xcd = batch >= 8 ? 1 : 0;                    // images dealt to XCDs (b mod 8); small batches: tiles dealt round-robin
            Pw.grid = dim3((unsigned)(Pw.dp_xcd ? 256 : std::min<long>(256, tiles)));
            Pw.pstr = pick_patch_pitch(32, 1 << btw, ((1 << btw) - 1) * S + KS, S, S == 2 ? 2 : 4);
            Pw.lds = lds + (size_t)nld * 32 * (Pw.pstr - 36) * 4;
            // the network head (32 -> 1 pointwise on the up2 of this unit's output) as the only reader: evaluated on the accumulators
            if (i + 2 < n_layers && !(flags & FD_PLAN_KEEP_ACTIVATIONS) && KS == 5 && D.mode == 2 && N == 32 && nt == 1 && wm == 4) {
                Layer &H = p->layers[i + 2];
                if (H.head && H.d.src == i + 1 && H.d.skip < 0 && readers[i + 1] == 1 && H.d.cin == 32) { Pw.fuse_head = i + 2; H.fused_into = i + 1; }
            }
        }
    }

    // 16-bit plans: the network head behind a pointwise layer of <= 32 channels (decode_conv5.1 -> decode_conv6) rides on that GEMM's output tile
}

// pass 4: 16-bit plans -- the network head on the output tile of the last pointwise GEMM
void plan_fuse_head_h16(fd_plan *p)
{
    const int n_layers = (int)p->layers.size(), batch = p->B, height = p->H, width = p->W, dtype = p->dtype;
    const uint32_t flags = p->flags, tune = p->tune;
    (void)n_layers; (void)batch; (void)height; (void)width; (void)dtype; (void)flags; (void)tune;
    // (fd_pw_gemm_head_h16): the 112x112xC tensor is neither written nor re-read and the head's launch disappears
    if (dtype != FD_F32 && !(flags & (FD_PLAN_KEEP_ACTIVATIONS | FD_PLAN_NO_EPILOGUE_FUSION))) {
        std::vector<int> rd(n_layers, 0);
        for (int i = 0; i < n_layers; ++i) {
            if (p->layers[i].d.src >= 0) ++rd[p->layers[i].d.src];
            if (p->layers[i].d.skip >= 0) ++rd[p->layers[i].d.skip];
        }
        for (int i = 0; i + 1 < n_layers; ++i) {
            Layer &Pw = p->layers[i], &H = p->layers[i + 1];
            if (Pw.d.op != FD_OP_PW || Pw.head || Pw.pw16_tm || Pw.dwpw || Pw.fuse_next_dw >= 0 || Pw.skipped || Pw.fused_into >= 0 || Pw.to_output) continue;
            if (!H.head || H.d.src != i || H.d.skip >= 0 || rd[i] != 1 || H.d.cin != Pw.d.cout || Pw.d.cout > 32 || Pw.d.cout % 8 || Pw.n_tiles != 1) continue;
            Pw.fuse_head = i + 1; H.fused_into = i;
        }
    }

    // activation arena
}

// pass 5: lifetime-based activation arena behind the packed weights (woff bytes)
void plan_arena(fd_plan *p, size_t woff)
{
    const int n_layers = (int)p->layers.size(), batch = p->B, height = p->H, width = p->W, dtype = p->dtype;
    const uint32_t flags = p->flags, tune = p->tune;
    (void)n_layers; (void)batch; (void)height; (void)width; (void)dtype; (void)flags; (void)tune;
    std::vector<int> last_use(n_layers, -1);
    for (int i = 0; i < n_layers; ++i) {
        if (p->layers[i].d.src >= 0) last_use[p->layers[i].d.src] = i;
        if (p->layers[i].d.skip >= 0) last_use[p->layers[i].d.skip] = i;
        if (p->layers[i].fused_dw >= 0) {                     // the fused kernel reads the depthwise layer's inputs
            const fd_layer_desc &dd = p->layers[p->layers[i].fused_dw].d;
            last_use[dd.src] = i;
            if (dd.skip >= 0) last_use[dd.skip] = i;
        }
    }
    FreeList fl;
    std::vector<char> released(n_layers, 0);
    for (int i = 0; i < n_layers; ++i) {
        Layer &L = p->layers[i];
        // a buffer whose last reader is layer i-1 (or earlier: layers that run inside another kernel are passed over below) is free from
        // layer i on; readers of layer i keep theirs
        if (!(flags & FD_PLAN_KEEP_ACTIVATIONS))
            for (int j = 0; j < i; ++j)
                if (!released[j] && last_use[j] >= 0 && last_use[j] <= i - 1 && !p->layers[j].to_output && !p->layers[j].skipped) {
                    fl.release(p->layers[j].out_off - woff, p->layers[j].out_bytes);
                    released[j] = 1;
                }
        if (L.to_output || L.skipped) continue;
        if (L.fused_into >= 0) continue;                     // allocated together with its producer (below)
        L.out_off = woff + fl.alloc(L.out_bytes);
        // a depthwise layer evaluated in this layer's epilogue is WRITTEN by this layer's kernel: its buffer must be live now, while this
        // kernel's own inputs are still being read (it must not reuse a buffer that becomes free only after this layer)
        if (L.fuse_next_dw >= 0) p->layers[L.fuse_next_dw].out_off = woff + fl.alloc(p->layers[L.fuse_next_dw].out_bytes);
    }
    p->ws_bytes = woff + fl.top;

    // bookkeeping: algorithmic traffic and descriptions (SURVEY.md 8(d) convention)
}

// pass 6: algorithmic bytes / flops, human-readable kernel descriptions, kernel symbols as rocprofv3 prints them
void plan_describe(fd_plan *p)
{
    const int n_layers = (int)p->layers.size(), batch = p->B, height = p->H, width = p->W, dtype = p->dtype;
    const uint32_t flags = p->flags, tune = p->tune;
    (void)n_layers; (void)batch; (void)height; (void)width; (void)dtype; (void)flags; (void)tune;
    const size_t esz = dtype == FD_F32 ? 4 : 2;
    for (int i = 0; i < n_layers; ++i) {
        Layer &L = p->layers[i];
        const fd_layer_desc &d = L.d;
        const int c_src = L.csplit ? L.csplit : d.cin, c_skip = L.csplit ? d.cin - L.csplit : d.cin;
        const double src_elems = (double)batch * (d.upsample ? (L.in_h / 2) * (L.in_w / 2) : L.in_h * L.in_w) * c_src;
        const double skip_elems = d.skip >= 0 ? (double)batch * L.in_h * L.in_w * c_skip : 0.0;
        const double out_elems = (double)batch * L.out_h * L.out_w * d.cout;
        const double in_esz = d.src < 0 ? 4.0 : (double)esz, out_esz = L.to_output ? 4.0 : (double)esz;   // network input / output stay fp32
        L.alg_bytes = (src_elems + skip_elems) * in_esz + out_elems * out_esz + (double)L.w_elems * (L.pw_packed_t ? esz : 4) + 2.0 * d.cout * 4;
        p->alg_bytes += L.alg_bytes;
        const double taps = d.op == FD_OP_STEM ? 27.0 : (d.op == FD_OP_DW ? (double)d.ksize * d.ksize : (double)d.cin);
        const double mac_px = L.head && d.upsample ? (double)L.out_h * L.out_w : (double)L.out_h * L.out_w;
        L.alg_flops = 2.0 * batch * mac_px * d.cout * taps;
        p->alg_flops += L.alg_flops;
        char buf[256];
        if (L.skipped)
            snprintf(buf, sizeof buf, "(fused into layer %d)", i + 1);
        else if (L.fused_into >= 0 && L.head)
            snprintf(buf, sizeof buf, "(pointwise head evaluated on the accumulators of layer %d's %s kernel)", L.fused_into, p->layers[L.fused_into].dwpw ? "dwpw" : "pw_gemm");
        else if (L.fused_into >= 0)
            snprintf(buf, sizeof buf, "(dw k%d s%d%s evaluated in the epilogue of layer %d's pw_gemm16)", d.ksize, d.stride, d.upsample ? " on up2" : "", L.fused_into);
        else if (L.dwpw)
            snprintf(buf, sizeof buf, "dwpw<dw k%d s%d mode%d + pw> persistent, 4 producer + 4 consumer waves; tile %dx%d px x all %d channels, C=%d in %d chunks, weights in LDS, grid=%u lds=%zu%s", p->layers[L.fused_dw].d.ksize,
                     p->layers[L.fused_dw].d.stride, p->layers[L.fused_dw].mode, L.dp_th, 1 << L.dp_tw, d.cout, d.cin, d.cin / 32, L.grid.x, L.lds,
                     L.fuse_head >= 0 ? " + the 32->1 head on the accumulators" : "");
        else if (d.op == FD_OP_STEM)
            snprintf(buf, sizeof buf, "stem3x3s2<mfma 32x32x2, LDS-staged rows, 256 px per workgroup> grid=%ux%u lds=%zu", L.grid.x, L.grid.y, L.lds);
        else if (d.op == FD_OP_DW && L.dw5_cl)
            snprintf(buf, sizeof buf, "dw5_rows<k5 s1 mode2, pixel pairs + dot2, %d channel lanes per strip> bands of %d rows, %d strip groups, %d channels per block, grid=%ux%ux%u, no LDS",
                     L.dw5_cl, L.dw5_bh, L.dw5_groups, L.dw5_cbs, L.grid.x, L.grid.y, L.grid.z);
        else if (d.op == FD_OP_DW && L.dw_rows)
            snprintf(buf, sizeof buf, "dw3_rows%s<s%d> rows/item %d grid=%ux%ux%u", L.dw_rows8 ? "8" : "", d.stride, L.th, L.grid.x, L.grid.y, L.grid.z);
        else if (d.op == FD_OP_DW)
            snprintf(buf, sizeof buf, "dwconv<k%d s%d mode%d> tile %dx%dx%d pitch %d grid=%ux%ux%u lds=%zu, %d channels per work-item", d.ksize, d.stride, L.mode,
                     L.th, L.tw, L.dw_n << L.cbq, L.pstr, L.grid.x, L.grid.y, L.grid.z, L.lds, L.dw_n);
        else if (L.head)
            snprintf(buf, sizeof buf, "head_pw1 up=%d grid=%u", d.upsample, L.grid.x);
        else
            if (L.pw16_tm)
                snprintf(buf, sizeof buf, "pw_gemm16<TM=%d: %dx64 tile, stride %d> M=%ld N=%d K=%d tiles=%dx%d (%.2f per CU) lds=%zu", L.pw16_tm, L.pw16_tm * 16, L.pw16_stride,
                         (long)batch * L.out_h * L.out_w, d.cout, d.cin, L.m_tiles, L.n_tiles, L.m_tiles * L.n_tiles / 256.0, L.lds),
                L.fuse_next_dw >= 0 ? (void)snprintf(buf + strlen(buf), sizeof buf - strlen(buf), " + fused dw k%d of layer %d", p->layers[L.fuse_next_dw].d.ksize, L.fuse_next_dw) : (void)0;
            else
            snprintf(buf, sizeof buf, "pw_gemm<%dx%d> M=%ld N=%d K=%d tiles=%dx%d lds=%zu", L.pw.wgm * L.pw.tm * 32,
                     L.pw.wgn * L.pw.tn * 32, (long)batch * L.out_h * L.out_w, d.cout, d.cin, L.m_tiles, L.n_tiles, L.lds),
            (L.fuse_head >= 0 && !L.dwpw) ? (void)snprintf(buf + strlen(buf), sizeof buf - strlen(buf), " + the %d->1 head on its output tile", d.cout) : (void)0;
        L.info = buf;
        const char *tn = dtype == FD_F32 ? "float" : (dtype == FD_F16 ? "_Float16" : "fd_bf16");
        if (L.skipped || L.fused_into >= 0) buf[0] = 0;
        else if (L.dwpw) snprintf(buf, sizeof buf, "fd_dwpw_f32<%d, %d, %d, %d, %d, %d, %d, %d, 0>", p->layers[L.fused_dw].d.ksize, p->layers[L.fused_dw].d.stride, p->layers[L.fused_dw].mode, d.act, L.dp_wm, L.dp_nt, L.dp_nld, L.fuse_head >= 0 ? 1 : 0);
        else if (d.op == FD_OP_STEM) snprintf(buf, sizeof buf, "fd_stem3x3s2<%s, %d, %d>", tn, d.act, L.chunk);
        else if (d.op == FD_OP_DW && L.dw5_cl) snprintf(buf, sizeof buf, "fd_dw5_rows<%s, %d, %d>", tn, d.act, L.dw5_cl);
        else if (d.op == FD_OP_DW && L.dw_rows) snprintf(buf, sizeof buf, "fd_dw3_rows%s<%s, %d, %d>", L.dw_rows8 ? "8" : "", tn, d.stride, d.act);
        else if (d.op == FD_OP_DW) snprintf(buf, sizeof buf, "fd_dwconv<%s, %d, %d, %d, %d, %d>", tn, d.ksize, d.stride, L.mode, d.act, L.dw_n);
        else if (L.head) snprintf(buf, sizeof buf, "fd_head_pw1<%s, %d>", tn, d.act);
        else if (L.pw16_tm && dtype != FD_F32) snprintf(buf, sizeof buf, "fd_pw_gemm16_h16<%s, %d, 4, %d, %d, 1>", tn, L.pw16_tm, d.act, L.fuse_next_dw >= 0 ? p->layers[L.fuse_next_dw].d.ksize : 0);
        else if (L.pw16_tm) snprintf(buf, sizeof buf, "fd_pw_gemm16_f32<%d, %d, %d, 0, %d, 0>", L.pw16_tm, FD_G16_STAGES, d.act, L.fuse_next_dw >= 0 ? p->layers[L.fuse_next_dw].d.ksize : 0);
        else if (dtype == FD_F32) snprintf(buf, sizeof buf, "fd_pw_gemm_f32<%d, %d, %d, %d, %d>", L.pw.wgm, L.pw.wgn, L.pw.tm, L.pw.tn, d.act);
        else if (L.fuse_head >= 0) snprintf(buf, sizeof buf, "fd_pw_gemm_head_h16<%s, %d>", tn, d.act);
        else snprintf(buf, sizeof buf, "fd_pw_gemm_h16<%s, %d>", tn, d.act);
        L.sym = buf;
    }
}

}  // namespace
