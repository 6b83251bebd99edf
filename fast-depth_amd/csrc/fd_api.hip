// fd_api.hip -- host side of libfastdepth_hip.so: plan construction, workspace layout, kernel
// selection/launch, and the C ABI declared in include/fastdepth_hip.h.
//
// The plan is the MI355X-native replacement for walking an nn.Sequential tree per call
// (reference models.py:706-732): the network is analysed once into a flat list of fused kernels with
// fixed grids, tile shapes and buffer addresses, so a forward is ~38 back-to-back launches on the
// caller's stream with no host-side decisions, allocations or synchronisation in between.
#include "fd_kernels_f32.h"
#include "fd_kernels_gemm16_f32.h"
#include "fd_kernels_h16.h"
#include "fd_kernels_gemm16_h16.h"
#include "fd_kernels_dwpw_f32.h"
#include "fd_kernels_dw5p.h"
#include "../../include/fastdepth_hip.h"
#include "fd_tuning.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#ifndef FD_EMU
#include <hip/hip_ext.h>
#endif

#define FD_DEFINE_HOST_STATE     // this unit defines the thread-local state the three units share (fd_host_common.h)
#include "fd_host_common.h"
#include "fd_plan_select.h"
#include "fd_infer_launch.h"
#include "fd_plan_build.h"   // (uses FD_G16_STAGES of fd_infer_launch.h)


// ==================================================================================================
extern "C" {

void fd_tuning_next(uint32_t mask) { fd_hs().tune_next = mask; }

const char *fd_last_error(void) { return fd_hs().err.c_str(); }
// FD_SOURCE_HASH: the first 16 hex digits of the SHA-256 over csrc/ and include/fastdepth_hip.h that fast-depth_amd/build.py computed when it compiled THIS
// binary (build.py's source_hash()); __graft_entry__.smoke() prints it next to the hash of the sources in the tree, so a reused library is visible
#ifndef FD_SOURCE_HASH
#define FD_SOURCE_HASH "unstamped"
#endif
const char *fd_version(void) { return "fastdepth_hip 0.5 (gfx950; inference f32/f16/bf16, train step f32/bf16; sources " FD_SOURCE_HASH ")"; }

int fd_plan_create(const fd_layer_desc *layers, int32_t n_layers, int32_t batch, int32_t height, int32_t width,
                   int32_t dtype, uint32_t flags, fd_plan **out_plan)
{
    const uint32_t tune = fd_take_tuning();                  // (consumed first: also when the creation fails)
    if (!layers || !out_plan || n_layers <= 0) return fail(FD_ERR_INVALID, "null/empty layer list");
    if (batch <= 0 || height <= 0 || width <= 0 || height % 32 || width % 32)
        return fail(FD_ERR_INVALID, "batch must be > 0 and height/width positive multiples of 32 (got %d, %dx%d)", batch, height, width);
    if (dtype != FD_F32 && dtype != FD_F16 && dtype != FD_BF16) return fail(FD_ERR_INVALID, "unknown dtype %d", dtype);
    if (flags & ~FD_PLAN_ALL_FLAGS) return fail(FD_ERR_INVALID, "unknown plan flag bits 0x%x", flags & ~FD_PLAN_ALL_FLAGS);
    if (tune & ~FD_TUNE_ALL) return fail(FD_ERR_INVALID, "unknown tuning bits 0x%x", tune & ~FD_TUNE_ALL);
    fd_plan *p = new fd_plan();
    p->B = batch; p->H = height; p->W = width; p->dtype = dtype; p->flags = flags; p->tune = tune;
    p->layers.resize(n_layers);
    size_t woff = 0;
    int rc = plan_layers(p, layers, &woff);
    if (rc) { delete p; return rc; }
    p->weights_bytes = woff;
    plan_fuse_epilogues(p);
    plan_fuse_units(p);
    plan_fuse_head_h16(p);
    plan_arena(p, woff);
    plan_describe(p);
    *out_plan = p;
    return FD_OK;
}

void fd_plan_destroy(fd_plan *plan) { delete plan; }

size_t fd_plan_workspace_bytes(const fd_plan *plan) { return plan ? plan->ws_bytes : 0; }

int fd_plan_bind_workspace(fd_plan *plan, void *device_ptr, size_t bytes)
{
    if (!plan || !device_ptr) return fail(FD_ERR_INVALID, "null plan/workspace");
    if (bytes < plan->ws_bytes) return fail(FD_ERR_INVALID, "workspace too small: %zu < %zu", bytes, plan->ws_bytes);
    if (reinterpret_cast<uintptr_t>(device_ptr) % 256) return fail(FD_ERR_INVALID, "workspace must be 256-byte aligned");
    plan->ws = static_cast<unsigned char *>(device_ptr);
    plan->packed = false;
    return FD_OK;
}

int fd_plan_pack_weights(fd_plan *plan, const fd_layer_params *params, int32_t n_layers, float bn_eps, void *stream)
{
    if (!plan || !params) return fail(FD_ERR_INVALID, "null plan/params");
    if (!plan->ws) return fail(FD_ERR_STATE, "bind a workspace before packing weights");
    if (n_layers != (int)plan->layers.size()) return fail(FD_ERR_INVALID, "expected %zu layer parameter sets", plan->layers.size());
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n_layers; ++i) {
        const Layer &L = plan->layers[i];
        const fd_layer_params &q = params[i];
        if (!q.conv_weight || !q.bn_weight || !q.bn_bias || !q.bn_mean || !q.bn_var) return fail(FD_ERR_INVALID, "layer %d: null parameter pointer", i);
        const int inner = (int)(L.w_elems / L.d.cout);
        const int transpose = (L.d.op == FD_OP_PW) ? 0 : 1;          // stem/dw kernels want tap-major [inner][cout]
        const int pitch = transpose ? inner : (L.w_pitch ? L.w_pitch : inner);   // pointwise rows are zero-padded to the GEMM's BK
        const long total = std::max<long>((long)L.d.cout * std::max(inner, pitch), L.d.cout);
        float *bptr = reinterpret_cast<float *>(plan->ws + L.b_off);
        if (L.pw_packed_t && plan->dtype == FD_F16)
            hipLaunchKernelGGL((fd_pack_fold<fd_half>), dim3(ceil_div(total, 256)), dim3(256), 0, s, q.conv_weight, q.bn_weight, q.bn_bias, q.bn_mean, q.bn_var, bn_eps,
                               reinterpret_cast<fd_half *>(plan->ws + L.w_off), bptr, L.d.cout, inner, transpose, pitch);
        else if (L.pw_packed_t)
            hipLaunchKernelGGL((fd_pack_fold<fd_bf16>), dim3(ceil_div(total, 256)), dim3(256), 0, s, q.conv_weight, q.bn_weight, q.bn_bias, q.bn_mean, q.bn_var, bn_eps,
                               reinterpret_cast<fd_bf16 *>(plan->ws + L.w_off), bptr, L.d.cout, inner, transpose, pitch);
        else
            hipLaunchKernelGGL((fd_pack_fold<float>), dim3(ceil_div(total, 256)), dim3(256), 0, s, q.conv_weight, q.bn_weight, q.bn_bias, q.bn_mean, q.bn_var, bn_eps,
                               reinterpret_cast<float *>(plan->ws + L.w_off), bptr, L.d.cout, inner, transpose, pitch);
        int rc = check_launch("fd_pack_fold");
        if (rc) return rc;
        if (L.dw5_cl) {                                      // the row-walking 5x5 kernel reads its folded taps as 16-bit pairs
            const float *wf = reinterpret_cast<const float *>(plan->ws + L.w_off);
            unsigned *wpk = reinterpret_cast<unsigned *>(plan->ws + L.wpk_off);
            if (plan->dtype == FD_F16) hipLaunchKernelGGL((fd_pack_dw5_pairs<fd_half>), dim3(ceil_div(30L * L.d.cin, 256)), dim3(256), 0, s, wf, wpk, L.d.cin);
            else hipLaunchKernelGGL((fd_pack_dw5_pairs<fd_bf16>), dim3(ceil_div(30L * L.d.cin, 256)), dim3(256), 0, s, wf, wpk, L.d.cin);
            rc = check_launch("fd_pack_dw5_pairs");
            if (rc) return rc;
        }
    }
    plan->packed = true;
    return FD_OK;
}

int fd_forward(fd_plan *plan, const void *x_nchw, void *y, void *stream)
{
    if (!plan || !x_nchw || !y) return fail(FD_ERR_INVALID, "null argument");
    if (!plan->ws || !plan->packed) return fail(FD_ERR_STATE, "plan needs a bound workspace and packed weights");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (size_t i = 0; i < plan->layers.size(); ++i) {
        const Layer &L = plan->layers[i];
        if (L.skipped || L.fused_into >= 0) continue;       // (a fused depthwise layer is produced by its producer's kernel)
        fd_hs().trace_layer = (int)i;
        int rc = run_layer(plan, L, static_cast<const float *>(x_nchw), static_cast<float *>(y), s);
        if (rc) return rc;
    }
    fd_hs().trace_layer = -1;
    return FD_OK;
}

#include "fd_bundle_impl.h"

int fd_plan_shape(const fd_plan *plan, int32_t *batch, int32_t *height, int32_t *width, int32_t *dtype)
{
    if (!plan) return fail(FD_ERR_INVALID, "null plan");
    if (batch) *batch = plan->B;
    if (height) *height = plan->H;
    if (width) *width = plan->W;
    if (dtype) *dtype = plan->dtype;
    return FD_OK;
}

#include "fd_trace_impl.h"

int fd_layer_output(const fd_plan *plan, int32_t layer, const void **device_ptr, int32_t *n, int32_t *h, int32_t *w, int32_t *c)
{
    if (!plan || layer < 0 || layer >= (int)plan->layers.size()) return fail(FD_ERR_INVALID, "bad layer index");
    if (!(plan->flags & FD_PLAN_KEEP_ACTIVATIONS)) return fail(FD_ERR_STATE, "plan was not created with FD_PLAN_KEEP_ACTIVATIONS");
    const Layer &L = plan->layers[layer];
    if (L.to_output) return fail(FD_ERR_STATE, "the last layer writes the caller's output buffer");
    if (L.skipped) return fail(FD_ERR_STATE, "layer %d is fused into its consumer and has no stored output", layer);
    if (!plan->ws) return fail(FD_ERR_STATE, "no workspace bound");
    if (device_ptr) *device_ptr = plan->ws + L.out_off;
    if (n) *n = plan->B;
    if (h) *h = L.out_h;
    if (w) *w = L.out_w;
    if (c) *c = L.d.cout;
    return FD_OK;
}

int32_t fd_plan_num_kernels(const fd_plan *plan) { return plan ? (int32_t)plan->layers.size() : 0; }   /* = number of layers; fused-away layers report an empty symbol */

const char *fd_plan_kernel_info(const fd_plan *plan, int32_t layer)
{
    if (!plan || layer < 0 || layer >= (int)plan->layers.size()) return "";
    return plan->layers[layer].info.c_str();
}

const char *fd_plan_kernel_symbol(const fd_plan *plan, int32_t layer)
{
    if (!plan || layer < 0 || layer >= (int)plan->layers.size()) return "";
    return plan->layers[layer].sym.c_str();
}

double fd_plan_algorithmic_bytes(const fd_plan *plan) { return plan ? plan->alg_bytes : 0.0; }
double fd_plan_algorithmic_flops(const fd_plan *plan) { return plan ? plan->alg_flops : 0.0; }

int fd_plan_layer_stats(const fd_plan *plan, int32_t layer, double *algorithmic_bytes, double *algorithmic_flops)
{
    if (!plan || layer < 0 || layer >= (int)plan->layers.size()) return fail(FD_ERR_INVALID, "bad layer index");
    // the per-unit convention of SURVEY.md 8(d) is kept: a fused launch is credited with both of its units' algorithmic work
    const Layer &L = plan->layers[layer];
    double b = (L.skipped || L.fused_into >= 0) ? 0.0 : L.alg_bytes, f = (L.skipped || L.fused_into >= 0) ? 0.0 : L.alg_flops;
    if (L.fused_dw >= 0) { b += plan->layers[L.fused_dw].alg_bytes; f += plan->layers[L.fused_dw].alg_flops; }
    if (L.fuse_next_dw >= 0) { b += plan->layers[L.fuse_next_dw].alg_bytes; f += plan->layers[L.fuse_next_dw].alg_flops; }
    if (L.fuse_head >= 0) { b += plan->layers[L.fuse_head].alg_bytes; f += plan->layers[L.fuse_head].alg_flops; }
    if (algorithmic_bytes) *algorithmic_bytes = b;
    if (algorithmic_flops) *algorithmic_flops = f;
    return FD_OK;
}

int fd_plan_layer_traffic(const fd_plan *plan, int32_t layer, double *needed_bytes)
{
    if (!plan || layer < 0 || layer >= (int)plan->layers.size() || !needed_bytes) return fail(FD_ERR_INVALID, "bad layer index / null output");
    // what the launch of this layer must move: the stored inputs it reads + the outputs it writes + its weights.  An intermediate tensor that a
    // fused launch keeps on chip (the pointwise output consumed by a depthwise epilogue, the depthwise output inside fd_dwpw_f32, decode_conv5's
    // output under the head) is neither written nor read: both passes of it leave the sum of the units' algorithmic bytes.
    const Layer &L = plan->layers[layer];
    const double esz = plan->dtype == FD_F32 ? 4.0 : 2.0;
    auto out_b = [&](const Layer &X) { return (double)plan->B * X.out_h * X.out_w * X.d.cout * (X.to_output ? 4.0 : esz); };
    double b = (L.skipped || L.fused_into >= 0) ? 0.0 : L.alg_bytes;
    if (L.fused_dw >= 0) { const Layer &D = plan->layers[L.fused_dw]; b += D.alg_bytes - 2.0 * out_b(D); }
    if (L.fuse_next_dw >= 0) {
        const Layer &D = plan->layers[L.fuse_next_dw];
        b += D.alg_bytes - ((plan->flags & FD_PLAN_KEEP_ACTIVATIONS) ? 1.0 : 2.0) * out_b(L);     // (KEEP_ACTIVATIONS plans still store the pointwise output)
    }
    if (L.fuse_head >= 0) { const Layer &H = plan->layers[L.fuse_head]; b += H.alg_bytes - 2.0 * out_b(L); }
    *needed_bytes = b;
    return FD_OK;
}

}  // extern "C"

// (the train plan lives in its own translation units: fd_train_fwd.hip, fd_train_bwd.hip)

