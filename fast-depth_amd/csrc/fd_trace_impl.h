// fd_trace_impl.h -- measurement aids: per-launch tracing and the per-layer timed forward (included inside fd_api.hip's extern "C" block)
// (translation unit fd_api.hip; split out of it in round 4 -- the plan code was a 1 160-line monolith)
#pragma once
int fd_trace_begin(void)
{
#ifdef FD_EMU
    return fail(FD_ERR_STATE, "kernel tracing needs the HIP build");
#else
    for (auto &t : fd_hs().trace) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    fd_hs().trace.clear();
    fd_hs().trace_on = true;
    return FD_OK;
#endif
}

int fd_trace_end(void *stream, fd_trace_record *records, int32_t max_records, int32_t *n_records)
{
#ifdef FD_EMU
    (void)stream; (void)records; (void)max_records; (void)n_records;
    return fail(FD_ERR_STATE, "kernel tracing needs the HIP build");
#else
    if (!fd_hs().trace_on) return fail(FD_ERR_STATE, "fd_trace_end without fd_trace_begin");
    fd_hs().trace_on = false;
    int rc = FD_OK;
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = fail(FD_ERR_HIP, "synchronisation failed");
    const int n = (int)fd_hs().trace.size();
    if (n_records) *n_records = n;
    for (int i = 0; i < n; ++i) {
        float ms = 0.0f;
        if (rc == FD_OK && hipEventElapsedTime(&ms, fd_hs().trace[i].e0, fd_hs().trace[i].e1) != hipSuccess) rc = fail(FD_ERR_HIP, "hipEventElapsedTime failed");
        if (records && i < max_records) { records[i].kernel = fd_hs().trace[i].name; records[i].layer = fd_hs().trace[i].layer; records[i].ms = ms; }
        (void)hipEventDestroy(fd_hs().trace[i].e0); (void)hipEventDestroy(fd_hs().trace[i].e1);
    }
    fd_hs().trace.clear();
    return rc;
#endif
}

int fd_forward_timed(fd_plan *plan, const void *x_nchw, void *y, void *stream, float *ms_per_layer, int32_t n_layers)
{
    if (!plan || !x_nchw || !y || !ms_per_layer) return fail(FD_ERR_INVALID, "null argument");
    if (!plan->ws || !plan->packed) return fail(FD_ERR_STATE, "plan needs a bound workspace and packed weights");
    if (n_layers != (int)plan->layers.size()) return fail(FD_ERR_INVALID, "expected room for %zu layer timings", plan->layers.size());
#ifdef FD_EMU
    for (int i = 0; i < n_layers; ++i) ms_per_layer[i] = 0.0f;
    return fd_forward(plan, x_nchw, y, stream);
#else
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<hipEvent_t> ev(2 * n_layers);
    for (auto &e : ev) if (hipEventCreate(&e) != hipSuccess) return fail(FD_ERR_HIP, "hipEventCreate failed");
    int rc = FD_OK;
    for (int i = 0; i < n_layers && rc == FD_OK; ++i) {
        if (plan->layers[i].skipped || plan->layers[i].fused_into >= 0) continue;
        fd_hs().ev_start = ev[2 * i]; fd_hs().ev_stop = ev[2 * i + 1];
        rc = run_layer(plan, plan->layers[i], static_cast<const float *>(x_nchw), static_cast<float *>(y), s);
    }
    fd_hs().ev_start = fd_hs().ev_stop = nullptr;
    if (rc == FD_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(FD_ERR_HIP, "hipStreamSynchronize failed");
    for (int i = 0; i < n_layers && rc == FD_OK; ++i) {
        if (plan->layers[i].skipped || plan->layers[i].fused_into >= 0) { ms_per_layer[i] = 0.0f; continue; }
        if (hipEventElapsedTime(&ms_per_layer[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = fail(FD_ERR_HIP, "hipEventElapsedTime failed");
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
#endif
}

