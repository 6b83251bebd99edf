// fd_host_common.h -- host-side basics of libfastdepth_hip.so: thread-local error text, the private tuning hand-over, the launch / trace macro, small helpers
// (one translation unit: included by fd_api.hip; split out of it in round 4 -- the plan code was a 1 160-line monolith)
#pragma once
namespace {

thread_local std::string g_err;

// private tuning mask handed over by fd_tuning_next (fd_tuning.h): consumed by the next plan creation of this thread
thread_local uint32_t g_tune_next = 0;
inline uint32_t fd_take_tuning() { const uint32_t t = g_tune_next; g_tune_next = 0; return t; }

// fd_forward_timed sets these so that the next launch records the kernel's own begin/end timestamps
// (hipExtLaunchKernelGGL start/stop events == what rocprofv3's kernel trace reports), without the
// launch-gap and event-record overhead that bracketing with hipEventRecord would add.
thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

// fd_trace_begin / fd_trace_end (measurement aid): every launch between the two carries its own begin/end events and is recorded with
// the source name of its kernel and the layer it belongs to (g_trace_layer, set by the layer loops; -1 outside them).
thread_local int g_trace_layer = -1;
#ifdef FD_EMU
#define FD_LAUNCH(kernel, grid, block, lds, stream, ...) hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__)
#else
struct TraceRec { const char *name; int layer; hipEvent_t e0, e1; };
thread_local bool g_trace_on = false;
thread_local std::vector<TraceRec> g_trace;
#define FD_LAUNCH(kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                        \
        if (g_trace_on) {                                                                                       \
            TraceRec tr_{#kernel, g_trace_layer, nullptr, nullptr};                                             \
            (void)hipEventCreate(&tr_.e0); (void)hipEventCreate(&tr_.e1);                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, tr_.e0, tr_.e1, 0, __VA_ARGS__);           \
            g_trace.push_back(tr_);                                                                             \
        } else if (g_ev_start) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, g_ev_start, g_ev_stop, 0, __VA_ARGS__); \
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
    g_err = buf;
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
