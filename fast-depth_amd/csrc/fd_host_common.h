// fd_host_common.h -- host-side basics of libfastdepth_hip.so: thread-local error text, the private tuning hand-over, the launch / trace macro, small helpers.
// Included by every translation unit of the library (fd_api.hip: inference + the definitions of the shared state below; fd_train_fwd.hip; fd_train_bwd.hip).
#pragma once

// ---- state shared by the translation units: ONE thread-local record, reached through fd_hs() (defined in fd_api.hip).  (Plain `extern thread_local`
// variables with hidden visibility do not work here: for every such variable the compiler references a weak "TLS init function" that does not exist for
// trivially initialised types, and a hidden weak-undefined symbol in PIC code resolves to the library's load address instead of null -- a jump to +0.)
#ifndef FD_EMU
struct TraceRec { const char *name; int layer; hipEvent_t e0, e1; };
#endif
struct fd_host_state {
    std::string err;                 // fd_last_error
    uint32_t tune_next = 0;          // private tuning mask handed over by fd_tuning_next (fd_tuning.h): consumed by the next plan creation of this thread
    // fd_forward_timed sets these so that the next launch records the kernel's own begin/end timestamps (hipExtLaunchKernelGGL start/stop events ==
    // what rocprofv3's kernel trace reports), without the launch-gap and event-record overhead that bracketing with hipEventRecord would add
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // fd_trace_begin / fd_trace_end (measurement aid): every launch between the two carries its own begin/end events and is recorded with the source
    // name of its kernel and the layer it belongs to (trace_layer, set by the layer loops; -1 outside them)
    int trace_layer = -1;
#ifndef FD_EMU
    bool trace_on = false;
    std::vector<TraceRec> trace;
#endif
};
__attribute__((visibility("hidden"))) fd_host_state &fd_hs();
#ifdef FD_DEFINE_HOST_STATE
fd_host_state &fd_hs() { static thread_local fd_host_state s; return s; }
#endif

namespace {

inline uint32_t fd_take_tuning() { const uint32_t t = fd_hs().tune_next; fd_hs().tune_next = 0; return t; }

#ifdef FD_EMU
#define FD_LAUNCH(kernel, grid, block, lds, stream, ...) hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__)
#else
#define FD_LAUNCH(kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                        \
        if (fd_hs().trace_on) {                                                                                       \
            TraceRec tr_{#kernel, fd_hs().trace_layer, nullptr, nullptr};                                             \
            (void)hipEventCreate(&tr_.e0); (void)hipEventCreate(&tr_.e1);                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, tr_.e0, tr_.e1, 0, __VA_ARGS__);           \
            fd_hs().trace.push_back(tr_);                                                                             \
        } else if (fd_hs().ev_start) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, fd_hs().ev_start, fd_hs().ev_stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                 \
    } while (0)
#endif

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fd_hs().err = buf;
    return code;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }


int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return FD_OK;
}
}  // namespace
