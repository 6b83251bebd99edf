// fd_train_plan.h -- the train plan's records (TLayer, fd_train_plan), build switches and small helpers: shared by the two train translation units
// (fd_train_fwd.hip: plan creation + train-mode forward, fd_train_impl.h; fd_train_bwd.hip: backward, loss, SGD, gradient exchange, fd_train_bwd_impl.h).
//
// Workspace layout of a train plan:
//   [z_i]   raw conv output of every unit, NHWC, fp32 or bf16 (plan dtype), all kept (they are the saved tensors of backward)
//   [st_i]  per-unit BatchNorm table [4][C]: scale, shift, mean, invstd (written by the unit's finalising workgroup: fd_bn_publish)
//   [stat]  per-unit BatchNorm statistics rows, forward (sum z, sum z^2) and backward (sum G, sum G*xhat): int64 [nr][3 bins][2][C] each (fd_device.h:
//           fd_stat_add / fd_stat_total), ONE contiguous region zeroed by a single memset at the start of every forward
//   backward only: [g_a, g_b] ping-pong dLoss/d(BN output) buffers, [skipgrad_k] decoder->skip gradient buffers,
//   [coef_i] per-unit BN-backward coefficient tables, [wpart_i] per-unit weight-gradient partials (reduced by ONE launch per backward
//   range: fd_reduce_weights_batch_f32).
//   bf16 plans: [wt_i, wtt_i] the 16-bit operand copies of every pointwise weight (re-made from the fp32 masters each step).
#pragma once
#include <type_traits>
#include "fd_kernels_train.h"
#include "fd_kernels_bwd.h"
#include "fd_kernels_train_h16.h"
#include "fd_kernels_io.h"

// workgroups the pointwise weight-gradient GEMM aims for (output tiles x pixel splits); every split writes a private fp32 partial
// tile that fd_reduce_partials_f32 sums afterwards, so more splits = more parallelism but more partial traffic
// (measured at batch 32: 2048 is best for the fp32 kernel; the bf16 one, whose MFMA part is 16x shorter, wants fewer)
// tile shapes of the 5x5 depthwise train kernels (build switches for tools/build_variant.py sweeps; the defaults are the measured best)
#ifndef FD_T_DW5_FTH
#define FD_T_DW5_FTH 8      // forward: rows (balanced over the map), columns
#define FD_T_DW5_FTW 16
#define FD_T_DW5_WTH 8      // backward-weights: output-space tile
#define FD_T_DW5_WTW 16
#define FD_T_DW5_DTH 8      // backward-data: input-space tile
#define FD_T_DW5_DTW 16
#endif
#ifndef FD_T_S2_DTH
#define FD_T_S2_DTH 8       // stride-2 units (single-staging backward kernel): input-space tile
#define FD_T_S2_DTW 16
#endif
#ifndef FD_DW_WGRAD_TARGET_WGS
#define FD_DW_WGRAD_TARGET_WGS 1536   // depthwise weight-gradient kernel: a workgroup walks up to a tile row's tiles as long as about this many workgroups remain
#endif
#ifndef FD_WGRAD_TUNE_H16
#define FD_WGRAD_TUNE_H16 640      // (round 3, paired launch: 640 -> 2.840 ms per bf16 step, 1024 -> 2.877, 512 -> 2.860; fewer splits = fewer partial bytes)
#endif
#ifndef FD_WGRAD_TARGET_WGS_F32
#define FD_WGRAD_TARGET_WGS_F32 1536   // (round 3, paired launch: 1024 -> 4.355 ms per fp32 step, 1536 -> 4.340, 2048 -> 4.361, 3072 -> 4.364)
#endif
#define FD_WGRAD_TARGET_WGS_H16 (FD_WGRAD_TUNE_H16)

// (a global name, not the anonymous namespace: fd_train_plan below has external linkage and both train translation units must see ONE class type --
// an anonymous-namespace member type makes the two definitions of fd_train_plan different classes, an ODR violation)
struct fd_train_layer {
    fd_layer_desc d;
    int in_h = 0, in_w = 0, out_h = 0, out_w = 0;   // in_* = logical (post-upsample) input size; head: out_* = LOW-res size when upsample
    bool head = false;
    int mode = 0;
    int csplit = 0;                  // concatenating depthwise consumer (mode 3): channels [0, csplit) come from src, the rest from skip
    int cbq = 0, th = 0, tw = 0, tiles_x = 0, tiles_y = 0;   // dw tiling (forward kernel)
    int dw_n = 4;                                             // channels per work-item of the LDS-tiled depthwise kernels (8: bf16 plans, storage-typed LDS patches; fd_lane)
    mutable int lds_rounding = 0;                             // fd_train_plan_lds_rounding: set by the launches of the last forward / backward
    mutable int bwd_rows = 0;                                 // the LAST backward of this depthwise unit ran on a row-walking kernel (fd_dw5_bwd_rows / fd_dw3_bwd_rows)
    int bth = 0, btw = 0;                                     // output-space tile of the backward-weights kernel
    int rows_th = 0;                                          // > 0: the forward runs on fd_dw3_rows_train with row strips of this height
    int dw3_cl = 0, dw3_groups = 0, dw3_bh = 0;               // dw3_cl > 0: the forward runs on fd_dw3_rows_fwd (16-bit plans, 3x3 on a plain input): channel lanes per strip, strip groups per row, output rows per band
    int dw5_groups = 0, dw5_bh = 0;                           // > 0: the forward runs on fd_dw5_rows_train (16-bit plans, 5x5 on up2 + skip): strip pairs per row, rows per band
    int stem_band = 0;                                        // floats of the stem kernels' input band in LDS
    int pstr = 0, bpstr = 0;                                  // LDS patch pitch (floats) of the forward / the backward depthwise kernels
    int chunk = 0;                                            // stem
    int m_tiles = 0, n_tiles = 0, pw_tn = 1;                  // pw (pw_tn: 32-column tiles per wave of the 16-bit forward GEMM)
    int pw16_tm = 0, pw16_stride = 0;                         // fp32 plans: the forward runs on fd_pw_gemm16_f32<TM, ..., TRAIN> (one workgroup per CU, rows in strides of pw16_stride)
    size_t lds = 0;
    dim3 grid;
    int nblk = 0;                    // workgroups per channel of this unit's forward kernel that add a partial sum to its statistics rows
    int nr_cap = 1;                  // statistics rows reserved per direction (stat_nr of the unit's pixel count / 64)
    int nr_f = 1;                    // rows its forward kernel deals its partials to (stat_nr(nblk), <= nr_cap)
    mutable int nr_b = 1;            // rows the backward kernel of its CONSUMER dealt the (sum G, sum G*xhat) partials to (set at that launch)
    size_t sf_off = 0, sb_off = 0;   // forward / backward statistics rows
    bool bwd_fin = false;            // the unit's first backward kernel can finalise its BatchNorm backward (bwd_fin_candidate)
    mutable int bwd_fin_rows = 0;    // > 0: it does so in this step (= nr_b; set when the consumer's backward kernels were launched)
    bool fin_by_consumer = false;    // this unit's BatchNorm is finalised inside its consumer's forward kernel (no fd_bn_finalize_rows_f32 launch)
    size_t z_off = 0, z_elems = 0;   // raw output
    size_t st_off = 0;               // [4][C] table
    size_t coef_off = 0;             // backward coefficient table [4][C]
    double n_stat = 0, n_unbiased = 0;
    long M = 0;                      // pixels of the stored output (B*out_h*out_w)
    // backward bookkeeping
    int consumer = -1;               // unit that reads this output as `src`
    int skip_consumer = -1;          // decoder unit that reads this output as `skip` (-1: none)
    size_t g_off = 0;                // dLoss/dy buffer of this unit
    size_t sg_off = 0;               // decoder->skip gradient buffer (only for skip sources)
    size_t wt_off = 0, wtt_off = 0;  // 16-bit plans: W as [N][K64] and W^T as [K][N64]
    size_t dz_off = 0;               // 16-bit pointwise units under FD_PLAN_KEEP_ACTIVATIONS: dz kept apart from G (0 = in place)
    size_t wp_off = 0, wp_elems = 0; // this unit's weight-gradient partial rows (reduced by one launch per backward range)
    int k64 = 0, n64 = 0;
};
typedef fd_train_layer TLayer;

struct fd_train_plan {
    std::vector<TLayer> layers;
    int B = 0, H = 0, W = 0, dtype = FD_F32;
    uint32_t flags = 0, tune = 0;    // public plan flags (include/fastdepth_hip.h) / private tuning mask (fd_tuning.h)
    size_t esz = 4;                  // bytes per stored activation / activation-gradient element
    size_t ws_bytes = 0, stat_off = 0, stat_bytes = 0;   // the statistics rows of all units (both directions): zeroed per step
    bool bwd_stats_clean = false;    // the backward rows are still zero (no backward has run since the last forward's memset)
    unsigned char *ws = nullptr;
    bool forward_done = false;
    float eps = 1e-5f;
    const void *x_saved = nullptr;   // the network input of the last forward (the stem's weight gradient re-reads it)
};

namespace {

inline float *tws(fd_train_plan *p, size_t off) { return reinterpret_cast<float *>(p->ws + off); }
template <typename T> inline T *twt(fd_train_plan *p, size_t off) { return reinterpret_cast<T *>(p->ws + off); }

// Statistics rows a producer of `nblk` workgroups per channel deals its partial sums to: one address takes an atomic every ~22 ns (measured,
// tools/microbench/stat_atomics.hip), so an address should see at most ~FD_STAT_ADDS_PER_ROW of them during the producer's life; a power of two
// <= FD_STAT_MAX_ROWS (the consumer adds nr x 3 integers per channel and sum).
#ifndef FD_STAT_ADDS_PER_ROW
#define FD_STAT_ADDS_PER_ROW 128
#endif
// A consumer (or the unit's own first backward kernel) finalises a BatchNorm in its prologue only when the rows are few: every workgroup reads
// nr x 3 bins x 2 sums of 8 bytes per channel.  ALL: kernels whose work-items each finalise whole channels (pointwise GEMMs, the register-window and head
// kernels); BLOCK: kernels that deal a channel's rows to 256 / CB work-items (LDS-tiled depthwise kernels, the 16-bit apply pass).  Units with more rows
// -- the large maps, whose kernels run 20 ... 160 us -- keep a finalisation launch of their own (fd_bn_finalize_rows_f32: 4 us).
#ifndef FD_STAT_FIN_MAX_ROWS_ALL
#define FD_STAT_FIN_MAX_ROWS_ALL 2
#define FD_STAT_FIN_MAX_ROWS_BLOCK 8
#endif
#ifndef FD_PW_BWD_W_FIRST
#define FD_PW_BWD_W_FIRST 1             // paired 16-bit pointwise backward: the (long-lived) weight-gradient workgroups are numbered before the backward-data tiles where they are the fewer
#endif
#ifndef FD_PW_BWD_W_FIRST_F32
#define FD_PW_BWD_W_FIRST_F32 1         // the same numbering in the fp32 paired launch (fd_pw_bwd_f32)
#endif
#ifndef FD_STEMW_BAND
#define FD_STEMW_BAND 14                // output rows per band of fd_stem_wgrad_rows
#endif
#ifndef FD_STEMW_CPL
#define FD_STEMW_CPL 2                  // output channels per lane of fd_stem_wgrad_rows (2 or 4)
#endif
#ifndef FD_STAT_FIN_MAX_ROWS_ROWK
#define FD_STAT_FIN_MAX_ROWS_ROWK 8       // ... the row-walking depthwise kernels (a few hundred fat workgroups per launch; 16 measured slower: DESIGN Appendix A)
#endif
inline int stat_nr(long nblk)
{
    int nr = 1;
    while (nr < FD_STAT_MAX_ROWS && (long)nr * FD_STAT_ADDS_PER_ROW < nblk) nr *= 2;
    return nr;
}
inline int stat_pitch(int C) { return (C + 15) / 16 * 16; }      // channel pitch of a unit's rows: every [row][bin][sum] slot starts its own 128-byte line
inline size_t stat_rows_bytes(int nr, int C) { return (size_t)nr * FD_STAT_BINS * 2 * stat_pitch(C) * sizeof(long long); }
inline long long *stat_ptr(fd_train_plan *p, size_t off) { return reinterpret_cast<long long *>(p->ws + off); }
// this unit's forward rows as its producer kernel sees them
inline fd_stat_rows fwd_rows(fd_train_plan *p, const TLayer &L) { return fd_stat_rows{stat_ptr(p, L.sf_off), L.nr_f, stat_pitch(L.d.cout)}; }
// unit u's backward rows as the backward-data kernel of its consumer (nblk workgroups per channel) sees them; remembers the row count for u's finalisation
inline fd_stat_rows bwd_rows(fd_train_plan *p, int u, long nblk)
{
    const TLayer &U = p->layers[u];
    U.nr_b = U.head ? U.nr_cap : std::min(U.nr_cap, stat_nr(nblk));      // (the head's single channel: every row is a line of its own -- all of them)
    return fd_stat_rows{stat_ptr(p, U.sb_off), U.nr_b, stat_pitch(U.d.cout)};
}

// calls fn(fd_int<4>) or -- 16-bit storage types only -- fn(fd_int<8>): the lane width (fd_lane) of the LDS-tiled depthwise kernels
template <typename T, typename F> inline void fd_by_lane_width(int n, F &&fn)
{
    if constexpr (!std::is_same<T, float>::value) { if (n == 8) { fn(fd_int<8>{}); return; } }
    fn(fd_int<4>{});
}
// host mirror of fd_lds_patch_bytes (fd_kernels_train.h): bytes of an LDS patch image of npx pixels at pitch pstr in elements of `le` bytes
inline size_t lds_patch_bytes(long npx, int pstr, int le) { return std::max(align_up((size_t)npx * pstr * le, 16), (size_t)8192); }

// ---- which backward form a unit takes: decided at plan time (fd_train_plan_create) and replayed by the launches (fd_train_bwd_impl.h) ----
// the paired depthwise launch has an instance for this unit's kernel size / stride / input composition / activations (dispatch_dw_bwd_pair)
inline bool dw_bwd_has_pair(const fd_train_plan *p, int i)
{
    const TLayer &L = p->layers[i];
    const TLayer &P = p->layers[L.d.src];
    const int a1 = P.d.act, a2 = L.d.skip >= 0 ? p->layers[L.d.skip].d.act : FD_ACT_RELU6;
    const bool add = P.skip_consumer >= 0 && L.mode == 0;
    const int key = L.d.ksize * 100 + L.d.stride * 10 + L.mode;
    if (a1 == FD_ACT_RELU6 && !add && (key == 310 || key == 320 || key == 510)) return true;
    if (a1 == FD_ACT_RELU6 && add && key == 320) return true;
    if (a1 == FD_ACT_RELU && key == 511) return true;
    if (a1 == FD_ACT_RELU && a2 == FD_ACT_RELU6 && (key == 512 || key == 513)) return true;
    return false;
}
// the stride-2 3x3 units of the large maps run their backward on the two register-window kernels (launch_dw_bwd_pair)
inline bool dw_bwd_on_rows(const fd_train_plan *p, const TLayer &L)
{
    const int cgn = L.d.cin / 4;
    const bool rows_ok = L.d.ksize == 3 && L.d.stride == 2 && L.mode == 0 && L.d.cin % 4 == 0 && cgn >= 8 && cgn <= 64 && (cgn & (cgn - 1)) == 0 &&
                         (((long)L.out_h * L.out_w >= 28 * 28 && p->esz == 2) || (p->tune & FD_TUNE_DW_FORCE_ROWS));
    return rows_ok && !(p->tune & (FD_TUNE_DW_NO_ROWS | FD_TUNE_DW_BWD_PAIR | FD_TUNE_DW_BWD1));
}
#ifndef FD_DW_ROWS_FIN
#define FD_DW_ROWS_FIN 1               // the row-walking backward kernels finalise their unit's BatchNorm backward in the prologue (0: the separate launch; A/B switch)
#endif
#ifndef FD_DW3_ROWS_MIN_PIXELS
#define FD_DW3_ROWS_MIN_PIXELS 0      // maps below this many pixels keep the paired LDS-tiled launch (tools/build_variant.py A/B switch)
#endif
// the unit's backward runs on a row-walking kernel of fd_kernels_dw5p_bwd.h (the launch-time conditions of launch_dw_bwd_pair, fd_train_bwd_impl.h)
inline bool dw_bwd_row_kernel(const fd_train_plan *p, int i)
{
    const TLayer &L = p->layers[i];
    if (p->esz != 2 || L.d.op != FD_OP_DW || L.d.cin % 8 || (double)L.in_h * L.in_w * L.d.cin * 2.0 >= 2147483648.0 || (p->flags & FD_PLAN_NO_BWD_PAIRING) || !dw_bwd_has_pair(p, i)) return false;
    if (p->tune & (FD_TUNE_NO_DW5_ROWS | FD_TUNE_DW_BWD1 | FD_TUNE_DW_BWD_PAIR)) return false;
    const bool add = p->layers[L.d.src].skip_consumer >= 0 && L.mode == 0;
    if (L.d.ksize == 5 && L.d.stride == 1 && L.mode == 2) return L.in_w % 4 == 0 && L.in_h % 2 == 0;
    if (p->tune & FD_TUNE_FORCE_DW_H8) return false;
    if (L.d.ksize == 3 && L.d.stride == 1 && L.mode == 0) return !add && (long)L.in_h * L.in_w >= FD_DW3_ROWS_MIN_PIXELS;
    if (L.d.ksize == 3 && L.d.stride == 2 && L.mode == 0) return !(p->tune & (FD_TUNE_DW_FORCE_ROWS | FD_TUNE_DW_NO_ROWS));
    return false;
}
// plan-time half of TLayer::bwd_fin: the unit's first backward kernel CAN finalise its BatchNorm backward (the LDS-tiled depthwise launches, the
// apply pass of the 16-bit pointwise units); whether it does is decided per step by the number of partial rows its consumer left (finalize_or_defer)
inline bool bwd_fin_candidate(const fd_train_plan *p, int i)
{
    const TLayer &L = p->layers[i];
    if ((p->tune & FD_TUNE_NO_CONSUMER_FINALIZE) || L.head || L.d.src < 0) return false;
    if (L.d.op == FD_OP_PW) return p->esz == 2;           // (16-bit plans: the apply pass fd_bn_bwd_apply_fin_h16; any row count now that the rows are few)
    // (depthwise units: built and measured in round 4 with up to 128 fp32 partial rows, off by default -- bf16 step: the 10 launches it removed were 43 us, the
    // paired kernels got 40 us slower (163 VGPRs + 36 bytes of scratch in the 3x3 instance); fp32 step +26 us.  FD_TUNE_DW_BWD_FINALIZE turns it on: tests, A/B)
    if (L.d.op == FD_OP_DW && FD_DW_ROWS_FIN && dw_bwd_row_kernel(p, i)) return true;     // (round 6: the row-walking kernels derive the coefficients in their prologue -- no registers carried through the walk)
    if (L.d.op == FD_OP_DW) return (p->tune & FD_TUNE_DW_BWD_FINALIZE) && !(p->flags & FD_PLAN_NO_BWD_PAIRING) && dw_bwd_has_pair(p, i) && !dw_bwd_on_rows(p, L);
    return false;
}


}  // namespace
