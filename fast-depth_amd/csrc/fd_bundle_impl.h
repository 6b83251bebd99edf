// fd_bundle_impl.h -- deploy bundle export / import (included inside fd_api.hip's extern "C" block)
// (translation unit fd_api.hip; split out of it in round 4 -- the plan code was a 1 160-line monolith)
#pragma once
/* ---- deploy bundle: layer descriptions + packed (BatchNorm-folded) weights, self-describing, loadable with no Python.  The analogue of the
 * reference's TVM artefacts deploy_graph.json + deploy_param.params (deploy/tx2_run_tvm.py:13-20). ---- */
namespace {
struct BundleHeader {
    char magic[8];            // "FDPLAN2\0" (FDPLAN1: rounds 2-3, whose flag word used bit values that have since been retired)
    uint32_t header_bytes, n_layers;
    int32_t batch, height, width, dtype;
    uint32_t flags, desc_bytes;
    uint64_t weights_bytes;   // the packed-weight region of the workspace, bit for bit
};
const char kBundleMagic[8] = {'F', 'D', 'P', 'L', 'A', 'N', '2', 0};
}  // namespace

size_t fd_plan_export_bytes(const fd_plan *plan)
{
    return plan ? sizeof(BundleHeader) + plan->layers.size() * sizeof(fd_layer_desc) + plan->weights_bytes : 0;
}

int fd_plan_export(const fd_plan *plan, void *host_buffer, size_t bytes, void *stream)
{
    if (!plan || !host_buffer) return fail(FD_ERR_INVALID, "null argument");
    if (!plan->ws || !plan->packed) return fail(FD_ERR_STATE, "export needs a bound workspace with packed weights");
    if (bytes < fd_plan_export_bytes(plan)) return fail(FD_ERR_INVALID, "export buffer too small: %zu < %zu", bytes, fd_plan_export_bytes(plan));
    BundleHeader h{};
    memcpy(h.magic, kBundleMagic, 8);
    h.header_bytes = sizeof(BundleHeader); h.n_layers = (uint32_t)plan->layers.size();
    h.batch = plan->B; h.height = plan->H; h.width = plan->W; h.dtype = plan->dtype;
    h.flags = plan->flags & ~FD_PLAN_KEEP_ACTIVATIONS; h.desc_bytes = sizeof(fd_layer_desc); h.weights_bytes = plan->weights_bytes;
    unsigned char *o = static_cast<unsigned char *>(host_buffer);
    memcpy(o, &h, sizeof h); o += sizeof h;
    for (const Layer &L : plan->layers) { memcpy(o, &L.d, sizeof(fd_layer_desc)); o += sizeof(fd_layer_desc); }
#ifdef FD_EMU
    (void)stream;
    memcpy(o, plan->ws, plan->weights_bytes);
#else
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(o, plan->ws, plan->weights_bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return fail(FD_ERR_HIP, "copying the packed weights to the host failed");
#endif
    return FD_OK;
}

int fd_plan_import(const void *host_buffer, size_t bytes, int32_t batch_override, fd_plan **out_plan)
{
    (void)fd_take_tuning();            // a deploy bundle runs the product's kernel selection: a tuning mask left in this thread's slot does not apply
    if (!host_buffer || !out_plan) return fail(FD_ERR_INVALID, "null argument");
    BundleHeader h{};
    if (bytes < sizeof h) return fail(FD_ERR_INVALID, "not a deploy bundle (too short)");
    memcpy(&h, host_buffer, sizeof h);
    if (memcmp(h.magic, kBundleMagic, 8) || h.header_bytes != sizeof h || h.desc_bytes != sizeof(fd_layer_desc))
        return fail(FD_ERR_INVALID, "not a deploy bundle of this library version");
    // every size in the header is file-controlled: each term is checked against what is left of the buffer (no sum that could wrap)
    const size_t rest = bytes - sizeof h;
    if (h.n_layers == 0 || h.n_layers > 4096 || h.n_layers > rest / sizeof(fd_layer_desc)) return fail(FD_ERR_INVALID, "truncated deploy bundle (layer table)");
    if (h.weights_bytes > rest - (size_t)h.n_layers * sizeof(fd_layer_desc)) return fail(FD_ERR_INVALID, "truncated deploy bundle (weights)");
    fd_plan *p = nullptr;
    int rc;
    try {
        std::vector<fd_layer_desc> descs(h.n_layers);
        memcpy(descs.data(), static_cast<const unsigned char *>(host_buffer) + sizeof h, (size_t)h.n_layers * sizeof(fd_layer_desc));
        // the packed weights do not depend on the batch size: a bundle exported at one batch serves any other
        // (descriptors and flags go through fd_plan_create's own validation, like a caller's)
        // (unknown flag bits are refused there; the private tuning mask is not part of a bundle)
        rc = fd_plan_create(descs.data(), (int32_t)h.n_layers, batch_override > 0 ? batch_override : h.batch, h.height, h.width, h.dtype, h.flags, &p);
    } catch (const std::exception &e) {
        return fail(FD_ERR_INVALID, "deploy bundle rejected: %s", e.what());          // no C++ exception crosses the C ABI
    }
    if (rc) return rc;
    if (p->weights_bytes != h.weights_bytes) { fd_plan_destroy(p); return fail(FD_ERR_INVALID, "bundle weight layout (%llu bytes) does not match this library (%zu)", (unsigned long long)h.weights_bytes, p->weights_bytes); }
    *out_plan = p;
    return FD_OK;
}

int fd_plan_import_weights(fd_plan *plan, const void *host_buffer, size_t bytes, void *stream)
{
    if (!plan || !host_buffer) return fail(FD_ERR_INVALID, "null argument");
    if (!plan->ws) return fail(FD_ERR_STATE, "bind a workspace before loading the bundle's weights");
    BundleHeader h{};
    if (bytes < sizeof h) return fail(FD_ERR_INVALID, "not a deploy bundle (too short)");
    memcpy(&h, host_buffer, sizeof h);
    if (memcmp(h.magic, kBundleMagic, 8) || h.weights_bytes != plan->weights_bytes || h.n_layers != plan->layers.size())
        return fail(FD_ERR_INVALID, "bundle does not belong to this plan");
    const size_t rest = bytes - sizeof h;
    if (h.n_layers > rest / sizeof(fd_layer_desc) || h.weights_bytes > rest - (size_t)h.n_layers * sizeof(fd_layer_desc)) return fail(FD_ERR_INVALID, "truncated deploy bundle");
    const unsigned char *w = static_cast<const unsigned char *>(host_buffer) + sizeof h + (size_t)h.n_layers * sizeof(fd_layer_desc);
#ifdef FD_EMU
    (void)stream;
    memcpy(plan->ws, w, h.weights_bytes);
#else
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(plan->ws, w, h.weights_bytes, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return fail(FD_ERR_HIP, "copying the packed weights to the device failed");
#endif
    plan->packed = true;
    return FD_OK;
}

