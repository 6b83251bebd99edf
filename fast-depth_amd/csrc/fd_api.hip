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

// Row pitch (floats) of the [pixels][pitch] LDS patch images that the depthwise kernels read with ds_read_b128 from (strip of `strip` pixels,
// channel group) work-items.  A wave64 ds_read_b128 is served in four fixed 16-lane groups, one LDS cycle each when the group's 16-byte
// pieces cover the 64 banks once (MI355X_MICROARCH.md, LDS); with the round-1 pitch cb + 4 = 36 the four strips of a group sat 144 dwords
// apart = 16 banks, two of them on the same banks: measured 32-45 % of all LDS cycles were conflict cycles in the 5x5 kernels, whose LDS
// pipe is 83 % busy.  This replays the lane -> address map of the kernels' strip reads and returns the smallest conflict-free pitch.
int pick_patch_pitch(int cb, int tw, int tw_in, int stride, int strip = 4)
{
    static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                   {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int lanes_c = cb / 4, tws = std::max(1, tw / strip);
    int cbq = 0; while ((1 << cbq) < lanes_c) ++cbq;
    int best = cb + 4; long best_cycles = -1;
    for (int pitch = cb + 4; pitch <= cb + 36; pitch += 4) {
        long cycles = 0;
        for (int wave = 0; wave < 4; ++wave)
            for (int g = 0; g < 4; ++g) {
                int first_addr[64][4], n_addr[64];
                for (int b = 0; b < 64; ++b) n_addr[b] = 0;
                for (int k = 0; k < 16; ++k) {
                    const int tid = wave * 64 + grp[g][k], c4 = tid & (lanes_c - 1), pt = tid >> cbq;
                    const int oy = pt / tws, ox = (pt - oy * tws) * strip;
                    const int addr = ((oy * stride) * tw_in + ox * stride) * pitch + c4 * 4;
                    for (int dw = 0; dw < 4; ++dw) {
                        const int b = (addr + dw) & 63;
                        bool seen = false;
                        for (int q = 0; q < n_addr[b] && q < 4; ++q) seen |= first_addr[b][q] == addr;
                        if (!seen) { if (n_addr[b] < 4) first_addr[b][n_addr[b]] = addr; ++n_addr[b]; }
                    }
                }
                int worst = 1;
                for (int b = 0; b < 64; ++b) worst = std::max(worst, n_addr[b]);
                cycles += worst;
            }
        if (best_cycles < 0 || cycles < best_cycles) { best_cycles = cycles; best = pitch; }
        if (cycles == 16) break;                               // 4 waves x 4 groups x 1 cycle: conflict free
    }
    return best;
}

struct PwCfg { int wgm, wgn, tm, tn; };

struct Layer {
    fd_layer_desc d;
    int in_h = 0, in_w = 0;      // logical input size (after upsampling)
    int out_h = 0, out_w = 0;
    size_t out_off = 0, out_bytes = 0;   // activation arena
    size_t w_off = 0, w_bytes = 0, b_off = 0;   // packed weights / bias
    size_t w_elems = 0;          // unpadded weight element count (algorithmic bytes)
    bool to_output = false;      // writes the network output buffer directly
    bool head = false;           // Cout == 1 pointwise: fd_head_pw1
    bool pw_packed_t = false;    // packed weights are 16-bit (pointwise layers of a 16-bit plan)
    // dw tiling
    int cbq = 0, th = 0, tw = 0, tiles_x = 0, tiles_y = 0, mode = 0, pstr = 0;   // pstr: LDS patch row pitch in LDS elements (pick_patch_pitch)
    int dw_n = 4;                // channels per work-item of the LDS-tiled depthwise kernel (8: 16-bit plans, storage-typed patches)
    int csplit = 0;              // concatenating consumer: channels [0, csplit) come from src, the rest from skip
    bool skipped = false;        // depthwise layer executed inside the following pointwise layer's fused kernel
    int fuse_next_dw = -1;       // pointwise layer (fd_pw_gemm16_f32): index of the depthwise consumer evaluated in its epilogue
    int fused_into = -1;         // depthwise layer: index of the pointwise layer whose kernel produces this layer's output
    int fused_dw = -1;           // pointwise layer: index of the depthwise layer that runs inside its fd_dwpw_f32 unit
    int fuse_head = -1;          // fd_dwpw_f32 unit: index of the 32 -> 1 pointwise head evaluated on its accumulators (that layer's fused_into = this one)
    bool dwpw = false;           // pointwise layer: fused_dw runs inside fd_dwpw_f32 (large maps: tile of pixels x all output channels)
    int dp_th = 0, dp_tw = 0 /* log2 of the tile width */, dp_tiles_x = 0, dp_wm = 0, dp_nt = 0, dp_nld = 0, dp_xcd = 0;
    bool dw_rows = false;        // register-window 3x3 kernel (fd_dw3_rows_f32) instead of the LDS-tiled one
    bool dw_rows8 = false;       // ... its 16-bit variant with eight channels per work-item (fd_dw3_rows8)
    // stem
    int chunk = 0;
    // pw
    PwCfg pw{};
    int m_tiles = 0, n_tiles = 0, w_pitch = 0;
    int pw16_tm = 0, pw16_stride = 0;   // > 0: fd_pw_gemm16_f32 (16x16x4 MFMA, one workgroup per CU) with TM row tiles and this M stride per workgroup
    size_t lds = 0;
    dim3 grid;
    std::string info, sym;
    double alg_bytes = 0, alg_flops = 0;
};

}  // namespace

struct fd_plan {
    std::vector<Layer> layers;
    int B = 0, H = 0, W = 0, dtype = 0;
    uint32_t flags = 0, tune = 0;    // public plan flags (include/fastdepth_hip.h) / private tuning mask (fd_tuning.h)
    size_t ws_bytes = 0, weights_bytes = 0;
    unsigned char *ws = nullptr;
    bool packed = false;
    double alg_bytes = 0, alg_flops = 0;
};

namespace {

// ---- lifetime-based arena ------------------------------------------------------------------------
struct FreeList {
    std::vector<std::pair<size_t, size_t>> blocks;   // (offset, size), sorted by offset
    size_t top = 0;
    size_t alloc(size_t bytes)
    {
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].second >= bytes) {
                size_t off = blocks[i].first;
                blocks[i].first += bytes;
                blocks[i].second -= bytes;
                if (blocks[i].second == 0) blocks.erase(blocks.begin() + i);
                return off;
            }
        // grow: extend a trailing free block if it touches the top
        if (!blocks.empty() && blocks.back().first + blocks.back().second == top) {
            size_t off = blocks.back().first;
            top = off + bytes;
            blocks.pop_back();
            return off;
        }
        size_t off = top;
        top += bytes;
        return off;
    }
    void release(size_t off, size_t bytes)
    {
        auto it = std::lower_bound(blocks.begin(), blocks.end(), std::make_pair(off, (size_t)0));
        it = blocks.insert(it, {off, bytes});
        if (it + 1 != blocks.end() && it->first + it->second == (it + 1)->first) { it->second += (it + 1)->second; blocks.erase(it + 1); }
        if (it != blocks.begin() && (it - 1)->first + (it - 1)->second == it->first) { (it - 1)->second += it->second; blocks.erase(it); }
    }
};

// ---- kernel selection ----------------------------------------------------------------------------
// Pointwise tile: the fp32 MFMA GEMM is compute-bound for most layers, so the tile is chosen to (a)
// not waste MFMA work on a ragged N, (b) give the 256 CUs at least ~2 workgroups each, (c) otherwise be
// as large as possible (fewer LDS/L2 bytes per flop).
PwCfg choose_pw(long M, int N)
{
    const PwCfg c128x128{2, 2, 2, 2}, c128x64{2, 2, 2, 1}, c64x128{2, 2, 1, 2}, c64x64{2, 2, 1, 1}, c128x32{4, 1, 1, 1};
    if (N <= 32) return c128x32;
    auto blocks = [&](const PwCfg &c) { return (long)ceil_div(M, c.wgm * c.tm * 32) * ceil_div(N, c.wgn * c.tn * 32); };
    auto waste = [&](const PwCfg &c) { int bn = c.wgn * c.tn * 32; return (double)(ceil_div(N, bn) * bn) / N; };
    // Measured on MI355X (tools/microbench/gemm_tiles.hip, round 1): with the fp32 MFMA at 64 cycles per instruction the 64x64 tile
    // (one 32x32 accumulator per wave, 4+ workgroups per CU) beats the larger tiles on every shape of this
    // network -- latency hiding across workgroups matters more than operand reuse.
    // Exception (same measurements): the 14x14 layers (M = 6272 at batch 32, N, K >= 256) run 13 % faster on 128x64 --
    // both shapes are bound by the same wave quantisation (3.06 32x32 tiles per SIMD), the larger tile halves the
    // L2 -> LDS bytes per flop.
    // Re-measured with the final kernel (tools/microbench/gemm_tiles.hip sweep over all 18 shapes of the network, batch 32): 6272x512x{256,512}
    // run fastest on 64x128 (37.9 us vs 39.6 on 128x64 vs 42.5 on 64x64), 25088x128x256 on 128x64 (23.2 vs 25.7); everything
    // else on 64x64.
    (void)blocks; (void)waste; (void)c128x128;
    if (M > 4096 && M <= 16384 && N >= 512) return c64x128;
    if (M > 16384 && M <= 32768 && N > 32 && N <= 128) return c128x64;
    return c64x64;
}

// Second-generation kernel (fd_kernels_gemm16_f32.h): worth it when ONE round of workgroups (one per CU) covers the layer with few
// idle tile slots -- then its 16x16 quantum removes the 3.06 -> 4 rounding of the 32x32 kernel.  Measured at batch 32
// (tools/microbench/gemm16.hip, profiles/r02): 6272x512x512 34.2 vs 38.1 us, 1568x1024x1024 40.1 vs 45.1, 6272x256x512 22.0 vs 23.2,
// 25088x128x256 23.0 vs 24.2; layers that need two rounds (25088x256x256) or have K < 256 gain nothing and keep the first kernel.
struct Pw16Cfg { int tm = 0, stride = 0; double score = 0; };
Pw16Cfg choose_pw16(long M, int N, int K, bool force)
{
    Pw16Cfg best;
    if (N % 4 || M <= 0) return best;
    if (force) {                                             // test mode: the largest row-tile count the layer can fill, balanced strides
        best.tm = M > 112 ? 13 : (M > 64 ? 7 : 4);
        const long mt = (M + 16 * best.tm - 1) / (16 * best.tm);
        best.stride = (int)((M + mt - 1) / mt); best.score = 1.0;
        return best;
    }
    const int nt = ceil_div(N, 64);
    for (int tm : {13, 7, 4}) {
        const long mtiles = std::max<long>(1, 256 / nt);      // the most M tiles one round can hold
        long stride = (M + mtiles - 1) / mtiles;
        if (stride > 16 * tm) continue;                      // would need a second round
        const long wgs = ((M + stride - 1) / stride) * nt;
        const double score = std::min(1.0, wgs / 256.0) * ((double)stride / (16 * tm));      // fraction of the chip's MFMA slots doing useful work
        if (score > best.score) { best.tm = tm; best.stride = (int)stride; best.score = score; }
    }
    if (best.score < 0.72 || K < 256) best = Pw16Cfg();
    return best;
}

int pw_lds_bytes(const PwCfg &c) { return FD_F32_STAGES * (c.wgm * c.tm * 32 + c.wgn * c.tn * 32) * 32 * 4; }   // 3-stage ring of 128-byte rows

int ilog2(int v) { int r = 0; while ((1 << r) < v) ++r; return r; }

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return FD_OK;
}

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
int launch_dw(const Layer &L, const T *in, const T *skip, const float *wp, const float *bias, T *out, hipStream_t s)
{
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
        do { (void)hipFuncSetAttribute((const void *)fd_pw_gemm16_f32<TMV, 3, ACT, 0, FD_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds); \
             FD_LAUNCH((fd_pw_gemm16_f32<TMV, 3, ACT, 0, FD_>), L.grid, dim3(512), L.lds, s, A, wp, bias, out, (int)M, N, K, L.w_pitch, L.pw16_stride, L.m_tiles, L.n_tiles, fz); } while (0)
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
    case FD_OP_DW: return launch_dw<T, ACT>(L, in, skip, wpf, bias, out, s);
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

// ==================================================================================================
extern "C" {

void fd_tuning_next(uint32_t mask) { g_tune_next = mask; }

const char *fd_last_error(void) { return g_err.c_str(); }
const char *fd_version(void) { return "fastdepth_hip 0.3 (gfx950; inference f32/f16/bf16, train step f32/bf16)"; }

int fd_plan_create(const fd_layer_desc *layers, int32_t n_layers, int32_t batch, int32_t height, int32_t width,
                   int32_t dtype, uint32_t flags, fd_plan **out_plan)
{
    if (!layers || !out_plan || n_layers <= 0) return fail(FD_ERR_INVALID, "null/empty layer list");
    if (batch <= 0 || height <= 0 || width <= 0 || height % 32 || width % 32)
        return fail(FD_ERR_INVALID, "batch must be > 0 and height/width positive multiples of 32 (got %d, %dx%d)", batch, height, width);
    const uint32_t tune = fd_take_tuning();                  // (consumed even when the creation fails)
    if (dtype != FD_F32 && dtype != FD_F16 && dtype != FD_BF16) return fail(FD_ERR_INVALID, "unknown dtype %d", dtype);
    if (flags & ~FD_PLAN_ALL_FLAGS) return fail(FD_ERR_INVALID, "unknown plan flag bits 0x%x", flags & ~FD_PLAN_ALL_FLAGS);
    if (tune & ~FD_TUNE_ALL) return fail(FD_ERR_INVALID, "unknown tuning bits 0x%x", tune & ~FD_TUNE_ALL);
    fd_plan *p = new fd_plan();
    p->B = batch; p->H = height; p->W = width; p->dtype = dtype; p->flags = flags; p->tune = tune;
    p->layers.resize(n_layers);
    const size_t esz = dtype == FD_F32 ? 4 : 2;   // activation / pointwise-weight element size
    size_t woff = 0;
    for (int i = 0; i < n_layers; ++i) {
        Layer &L = p->layers[i];
        L.d = layers[i];
        const fd_layer_desc &d = L.d;
#define FD_BAD(...) do { int rc_ = fail(FD_ERR_INVALID, __VA_ARGS__); delete p; return rc_; } while (0)
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
                        L.lds = (size_t)(dtype == FD_F32 ? 3 : 4) * (c16.tm * 16 + 64) * 32 * 4;   // (128-byte rows in both kernels; the 16-bit one runs a 4-stage ring)
                        L.m_tiles = ceil_div(M, c16.stride); L.n_tiles = ceil_div(d.cout, 64);
                        L.grid = dim3((unsigned)((L.m_tiles + 7) / 8 * 8 * L.n_tiles));
                    }
                }
            }
            break;
        default: FD_BAD("layer %d: unknown op %d", i, d.op);
        }
        if (L.lds > 160 * 1024) FD_BAD("layer %d: LDS request %zu exceeds 160 KiB", i, L.lds);
        L.w_off = woff; woff += align_up(L.w_bytes, 256);
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
        else if (d.op == FD_OP_DW && L.dw_rows) snprintf(buf, sizeof buf, "fd_dw3_rows%s<%s, %d, %d>", L.dw_rows8 ? "8" : "", tn, d.stride, d.act);
        else if (d.op == FD_OP_DW) snprintf(buf, sizeof buf, "fd_dwconv<%s, %d, %d, %d, %d, %d>", tn, d.ksize, d.stride, L.mode, d.act, L.dw_n);
        else if (L.head) snprintf(buf, sizeof buf, "fd_head_pw1<%s, %d>", tn, d.act);
        else if (L.pw16_tm && dtype != FD_F32) snprintf(buf, sizeof buf, "fd_pw_gemm16_h16<%s, %d, 4, %d, %d, 1>", tn, L.pw16_tm, d.act, L.fuse_next_dw >= 0 ? p->layers[L.fuse_next_dw].d.ksize : 0);
        else if (L.pw16_tm) snprintf(buf, sizeof buf, "fd_pw_gemm16_f32<%d, 3, %d, 0, %d>", L.pw16_tm, d.act, L.fuse_next_dw >= 0 ? p->layers[L.fuse_next_dw].d.ksize : 0);
        else if (dtype == FD_F32) snprintf(buf, sizeof buf, "fd_pw_gemm_f32<%d, %d, %d, %d, %d>", L.pw.wgm, L.pw.wgn, L.pw.tm, L.pw.tn, d.act);
        else if (L.fuse_head >= 0) snprintf(buf, sizeof buf, "fd_pw_gemm_head_h16<%s, %d>", tn, d.act);
        else snprintf(buf, sizeof buf, "fd_pw_gemm_h16<%s, %d>", tn, d.act);
        L.sym = buf;
    }
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
        g_trace_layer = (int)i;
        int rc = run_layer(plan, L, static_cast<const float *>(x_nchw), static_cast<float *>(y), s);
        if (rc) return rc;
    }
    g_trace_layer = -1;
    return FD_OK;
}

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

int fd_plan_shape(const fd_plan *plan, int32_t *batch, int32_t *height, int32_t *width, int32_t *dtype)
{
    if (!plan) return fail(FD_ERR_INVALID, "null plan");
    if (batch) *batch = plan->B;
    if (height) *height = plan->H;
    if (width) *width = plan->W;
    if (dtype) *dtype = plan->dtype;
    return FD_OK;
}

int fd_trace_begin(void)
{
#ifdef FD_EMU
    return fail(FD_ERR_STATE, "kernel tracing needs the HIP build");
#else
    for (auto &t : g_trace) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    g_trace.clear();
    g_trace_on = true;
    return FD_OK;
#endif
}

int fd_trace_end(void *stream, fd_trace_record *records, int32_t max_records, int32_t *n_records)
{
#ifdef FD_EMU
    (void)stream; (void)records; (void)max_records; (void)n_records;
    return fail(FD_ERR_STATE, "kernel tracing needs the HIP build");
#else
    if (!g_trace_on) return fail(FD_ERR_STATE, "fd_trace_end without fd_trace_begin");
    g_trace_on = false;
    int rc = FD_OK;
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = fail(FD_ERR_HIP, "synchronisation failed");
    const int n = (int)g_trace.size();
    if (n_records) *n_records = n;
    for (int i = 0; i < n; ++i) {
        float ms = 0.0f;
        if (rc == FD_OK && hipEventElapsedTime(&ms, g_trace[i].e0, g_trace[i].e1) != hipSuccess) rc = fail(FD_ERR_HIP, "hipEventElapsedTime failed");
        if (records && i < max_records) { records[i].kernel = g_trace[i].name; records[i].layer = g_trace[i].layer; records[i].ms = ms; }
        (void)hipEventDestroy(g_trace[i].e0); (void)hipEventDestroy(g_trace[i].e1);
    }
    g_trace.clear();
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
        g_ev_start = ev[2 * i]; g_ev_stop = ev[2 * i + 1];
        rc = run_layer(plan, plan->layers[i], static_cast<const float *>(x_nchw), static_cast<float *>(y), s);
    }
    g_ev_start = g_ev_stop = nullptr;
    if (rc == FD_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(FD_ERR_HIP, "hipStreamSynchronize failed");
    for (int i = 0; i < n_layers && rc == FD_OK; ++i) {
        if (plan->layers[i].skipped || plan->layers[i].fused_into >= 0) { ms_per_layer[i] = 0.0f; continue; }
        if (hipEventElapsedTime(&ms_per_layer[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = fail(FD_ERR_HIP, "hipEventElapsedTime failed");
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return rc;
#endif
}

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

#include "fd_train_impl.h"
