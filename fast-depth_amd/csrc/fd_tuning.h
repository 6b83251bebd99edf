// fd_tuning.h -- PRIVATE tuning / test mask of libfastdepth_hip.so.  Not part of the boundary (include/fastdepth_hip.h): a caller
// of the library never needs any of this.  The bits select kernel variants for A/B measurements (tools/gpu_ab.sh) and let the small
// shapes of the test tiers exercise kernels that the plan would otherwise only pick at full size.
//
// Hand-over: fd_tuning_next(mask) stores the mask in a thread-local slot; the NEXT fd_plan_create / fd_train_plan_create /
// fd_plan_import of the same thread consumes it (and resets the slot to 0), so plans created by anybody who does not call the hook are
// untouched.  The hook is exported from the shared library but declared only here; fast-depth_amd/fastdepth_hip/capi.py mirrors the
// values as FD_TUNE_* (Python ints shifted above bit 32 so that a single `flags=` argument can carry both masks).
#pragma once
#include <stdint.h>

#define FD_TUNE_WGRAD_TILE_ROWS 1u        /* train: depthwise weight-gradient workgroups always walk a whole row of tiles (small test shapes exercise the tile loop) */
#define FD_TUNE_FORCE_GEMM16 2u           /* every pointwise layer with cout % 4 == 0 on the one-workgroup-per-CU GEMM (fd_pw_gemm16_*), whatever its shape */
#define FD_TUNE_FORCE_EPILOGUE_FUSION 4u  /* 16-bit plans: fd_pw_gemm16_h16 + fused depthwise consumer for every eligible pair */
#define FD_TUNE_FORCE_UNIT_FUSION 8u      /* fd_dwpw_f32 for every eligible depthwise + pointwise pair whatever the map size */
#define FD_TUNE_NO_PW_PAIRING 16u         /* train: pair only the depthwise units' backward kernels, not the pointwise GEMMs */
#define FD_TUNE_PW_PAIR_TN2 32u           /* 16-bit train: the paired pointwise backward keeps 64 x 128 backward-data tiles */
#define FD_TUNE_DW_BWD1 64u               /* train: every depthwise unit's backward runs as the single-staging kernel fd_dw_bwd1 */
#define FD_TUNE_DW_BWD_PAIR 128u          /* train: every depthwise unit's backward runs as the paired launch fd_dw_bwd */
#define FD_TUNE_DW_SMALL_TILES 256u       /* train: the depthwise FORWARD kernel keeps the 7..8 x 16 tiles of the backward kernels */
#define FD_TUNE_DW_PITCH4 512u            /* train: LDS patch pitch +4 floats (PITCH8: +8; both: +12).  PITCH8 must stay PITCH4 << 1 */
#define FD_TUNE_DW_PITCH8 1024u
#define FD_TUNE_DW_WGRAD_TH4 2048u        /* train: 5x5 weight-gradient workgroups take output tiles of 4 rows */
#define FD_TUNE_DW_NO_ROWS 4096u          /* train: the 3x3 depthwise kernels always run LDS-tiled (no register-window kernels) */
#define FD_TUNE_DW_FORCE_ROWS 8192u       /* tests: the register-window kernels on every eligible 3x3 unit whatever the map size */
#define FD_TUNE_DW_TH8 16384u             /* train: depthwise tiles of 8 rows with a ragged last tile instead of balanced row counts */
#define FD_TUNE_DW_CB16 32768u            /* train: depthwise kernels work on 16-channel blocks instead of 32 */
#define FD_TUNE_NO_DW_H8 65536u           /* 16-bit plans: every LDS-tiled depthwise kernel keeps fp32 patches and 4 channels per work-item (round-1..3 form) */
#define FD_TUNE_FORCE_DW_H8 131072u       /* 16-bit plans: storage-typed LDS patches and 8 channels (16 bytes) per work-item (fd_lane<T, 8>) on every eligible
                                             LDS-tiled depthwise kernel, inference and train (default: only where it was measured to pay -- the 5x5
                                             inference units on maps >= 56x56; tests and A/B runs) */
#define FD_TUNE_NO_CONSUMER_FINALIZE 262144u /* train plans: every BatchNorm is finalised by its own fd_bn_finalize_f32 launch (default: the depthwise consumer of a
                                             pointwise unit with <= 128 partial rows finalises it, fd_bn_finalize_block; backward: the
                                             apply pass of a 16-bit pointwise unit with <= 128 partial rows, fd_bn_bwd_apply_fin_h16) */
#define FD_TUNE_DW_BWD_FINALIZE 524288u    /* train plans: the LDS-tiled depthwise backward launches finalise their unit's BatchNorm backward themselves when its partial
                                             rows are <= 128 (fd_bn_bwd_finalize_block; default: only the apply pass of the 16-bit pointwise units does) */
#define FD_TUNE_NO_DW5_ROWS 1048576u       /* 16-bit plans: the 5x5 up2 + skip units keep the LDS-tiled fd_dwconv (default since round 6: the row-walking pixel-pair
                                             kernel fd_dw5_rows, fd_kernels_dw5p.h) -- A/B runs and the tests of the older form */
#define FD_TUNE_ALL 2097151u

#ifdef __cplusplus
extern "C" {
#endif
/* the mask applies to the next plan this thread creates (then resets to 0); unknown bits are rejected by that creation */
void fd_tuning_next(uint32_t mask);
/* Test hook of the layer-local train parity (tests/harness.py: the fp64 single-unit reference rounds exactly where the kernels round): which
 * kernels of depthwise unit `layer` kept their LDS patches in the 16-bit storage type during the LAST forward / backward of this plan --
 * bit 0: the forward kernel rounded its (activated, upsampled, skip-added) conv input; bit 1: the backward kernels rounded dz and the
 * re-created conv input; bit 2 / bit 3: the forward / the backward-data kernel rounded its taps as well (fd_dw5_rows_train / fd_dw5_bwd_rows).  0 for fp32 plans, the 4-channel form and the
 * register-window kernels.  -1: bad arguments. */
struct fd_train_plan;
int fd_train_plan_lds_rounding(const struct fd_train_plan *plan, int32_t layer);
/* Test hook: which of the selectable forms unit `layer` of a train plan runs on -- bit 0: its forward pointwise GEMM is fd_pw_gemm16_f32 in train mode
 * (fp32 plans, one round of workgroups); bit 1: its BatchNorm statistics are finalised inside the consuming depthwise kernel (fd_bn_finalize_block);
 * bit 2: in the LAST backward its BatchNorm backward was finalised inside its own first backward kernel (fd_bn_bwd_apply_fin_h16 /
 * fd_bn_bwd_finalize_block); bit 3: its LAST backward ran on a row-walking depthwise kernel (fd_dw5_bwd_rows / fd_dw3_bwd_rows / fd_dw3s2_bwd_rows); bit 4: its forward runs on
 * fd_dw5_rows_train; bit 5: on fd_dw3_rows_fwd.  -1: bad arguments. */
int fd_train_plan_unit_kernels(const struct fd_train_plan *plan, int32_t layer);
/* Measurement hook (tools/gpu_round.sh, bench.py with FD_BENCH_FORCE_DIST=2): fd_train_backward_allreduce runs everything -- bucket ranges, event
 * hand-over to the communicator's stream, casts, the wait of the compute stream -- EXCEPT the ncclAllReduce calls.  On one rank this separates the
 * cost of the library's own machinery from RCCL's degenerate one-rank collective (a run of small copy / fill kernels).  Only valid on one rank. */
struct fd_comm;
void fd_comm_elide_collectives(struct fd_comm *comm, int32_t on);
/* Test hook: the shared library that provides ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy, instead of librccl.so.1.  Must be
 * called before the process first touches fd_comm_* (the binding is made once).  The CPU test tier binds tests/rccl_stub (the four entry points over
 * POSIX shared memory) to the EMULATOR build and so runs fd_train_backward_allreduce at world size 2 without a GPU (tests/test_dp_gloo.py). */
int fd_comm_bind_library(const char *path);
#ifdef __cplusplus
}
#endif
