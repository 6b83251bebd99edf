/*
 * fastdepth_hip.h -- C ABI of libfastdepth_hip.so: the MI355X (gfx950) execution engine for the
 * FastDepth MobileNet-NNConv5(dw)+skip-add hot path.
 *
 * Plain pointers and sizes only; no torch / C++ types.  Device pointers are HIP device pointers of
 * the calling process (torch allocations are fine), `stream` is a hipStream_t passed as void*.
 * Nothing here synchronises the device; every call returns FD_OK (0) or a negative error and
 * fd_last_error() describes the failure (thread-local).
 *
 * What each entry point replaces in the reference (dwofk/fast-depth):
 *   fd_plan_create            the static structure of models.MobileNetSkipAdd.__init__        models.py:655-704
 *                             (+ imagenet/mobilenet.py:40-54), expressed as a list of fused layers
 *   fd_plan_pack_weights      what `model.eval()` + nn.BatchNorm2d's running statistics imply   mobilenet.py:25,32,36; models.py:66,73
 *                             (BN folded into the preceding conv once, instead of per call)
 *   fd_forward                `pred = model(input)` inside torch.no_grad()                      main.py:74-75  ->  models.py:706-732
 *                             i.e. 38 aten::convolution + 38 batch_norm + 38 activations +
 *                             5 upsample_nearest2d + 3 add, as ~38 fused HIP kernels
 *   fd_layer_output           (test hook) the per-module outputs a forward hook would see
 *   fd_metrics_*              (next row f-2) metrics.Result.evaluate                            metrics.py:31-55
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Layout contract: network input is NCHW float32 exactly as main.py:68 hands it over
 * ([B,3,H,W], values in [0,1], no mean/std normalisation: dataloaders/dataloader.py:97-99); the
 * network output [B,1,H,W] is written densely (NCHW == NHWC for one channel).  All intermediate
 * activations live in the caller-provided workspace in NHWC and never cross this boundary except
 * through fd_layer_output.
 */
#ifndef FASTDEPTH_HIP_H
#define FASTDEPTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_OK 0
#define FD_ERR_INVALID (-1)   /* bad argument / unsupported structure */
#define FD_ERR_STATE (-2)     /* call order (workspace not bound, weights not packed, ...) */
#define FD_ERR_HIP (-3)       /* a HIP runtime call or kernel launch failed */

/* storage / arithmetic type of activations and packed weights (accumulation is always fp32) */
enum fd_dtype { FD_F32 = 0, FD_F16 = 1, FD_BF16 = 2 };

/* fused layer kinds: every one is Conv2d(bias=False) + BatchNorm2d + activation */
enum fd_op {
    FD_OP_STEM = 0, /* dense 3x3 conv, stride 2, NCHW-planar in -> NHWC out   (mobilenet.py:22-27,41)  */
    FD_OP_DW = 1,   /* depthwise k x k conv (k = 3 or 5), stride 1 or 2        (mobilenet.py:31-33; models.py:61-68) */
    FD_OP_PW = 2    /* pointwise 1x1 conv = GEMM over channels                 (mobilenet.py:35-37; models.py:70-75) */
};
enum fd_act { FD_ACT_NONE = 0, FD_ACT_RELU = 1, FD_ACT_RELU6 = 2 };

/* plan flags (fd_plan_create / fd_train_plan_create).  Unknown bits are rejected.  The kernel-selection switches used for A/B
 * measurements and by the test tiers are NOT part of this boundary: they live in fast-depth_amd/csrc/fd_tuning.h. */
#define FD_PLAN_KEEP_ACTIVATIONS 1u     /* one private buffer per layer output (needed for fd_layer_output / fd_train_layer_tensor); default: lifetime-based reuse */
#define FD_PLAN_NO_GEMM16 64u           /* fallback: never use the one-workgroup-per-CU pointwise GEMMs (fd_pw_gemm16_*); the first-generation 64x64-tile GEMM runs every pointwise layer */
#define FD_PLAN_NO_ROWS8 256u           /* fallback (16-bit plans): the 3x3 depthwise layers run on the 4-channel register-window kernel instead of the 8-channel one */
#define FD_PLAN_NO_EPILOGUE_FUSION 512u /* fallback: never evaluate a depthwise layer in the epilogue of its pointwise producer, nor the head on a GEMM's output tile */
#define FD_PLAN_NO_UNIT_FUSION 1024u    /* fallback: never run a depthwise + pointwise unit of a large map as one kernel (fd_dwpw_f32) */
#define FD_PLAN_NO_BWD_PAIRING 4096u    /* fallback (train plans): a unit's backward-data and backward-weights kernels as two launches instead of one paired launch */
#define FD_PLAN_ALL_FLAGS (1u | 64u | 256u | 512u | 1024u | 4096u)
/* (bit values are never reused: 4, 8, 16, 32, 128, 2048, 8192 and everything from 65536 up selected experiments and tuning aids of rounds 1-3
 * and are rejected now; deploy bundles written by those library versions carry the magic "FDPLAN1" and are refused by fd_plan_import.) */

typedef struct fd_layer_desc {
    int32_t op;       /* enum fd_op */
    int32_t cin;
    int32_t cout;     /* == cin for FD_OP_DW */
    int32_t ksize;    /* 3 (stem, encoder dw), 5 (decoder dw), 1 (pw) */
    int32_t stride;   /* 1 or 2 */
    int32_t act;      /* enum fd_act */
    int32_t src;      /* index of the layer whose output feeds this one; -1 = network input */
    int32_t upsample; /* 1: the input is the nearest-neighbour x2 upsampling of src's output (models.py:723) */
    int32_t skip;     /* >= 0: add that layer's output to the (upsampled) input before the conv (models.py:724-729); -1: none */
    int32_t concat;   /* 0: the skip tensor is ADDED (MobileNetSkipAdd); 1: it is CONCATENATED after the (upsampled) source along the channel
                       * axis, cin = C_src + C_skip (MobileNetSkipConcat, reference models.py:796-811; depthwise consumers; train plans need C_src % 32 == 0) */
} fd_layer_desc;

/* device pointers to the live parameters of one layer (fp32, torch layouts) */
typedef struct fd_layer_params {
    const float *conv_weight;  /* [cout][cin/groups][k][k] */
    const float *bn_weight;    /* gamma [cout] */
    const float *bn_bias;      /* beta  [cout] */
    const float *bn_mean;      /* running_mean [cout] */
    const float *bn_var;       /* running_var  [cout] */
    int64_t *bn_num_batches_tracked; /* train forward: += 1 per step (nn.BatchNorm2d's counter); may be NULL; unused by inference */
} fd_layer_params;

typedef struct fd_plan fd_plan;

/* Builds the execution plan (kernel selection, tiling, workspace layout) for a fixed
 * (batch, height, width, dtype).  height and width must be multiples of 32 (the reference fails at
 * its skip additions otherwise).  The last layer must produce the network output. */
int fd_plan_create(const fd_layer_desc *layers, int32_t n_layers, int32_t batch, int32_t height,
                   int32_t width, int32_t dtype, uint32_t flags, fd_plan **out_plan);
void fd_plan_destroy(fd_plan *plan);

/* Bytes of device memory the plan needs (packed weights + activation arena); the caller allocates
 * it (e.g. a torch uint8 tensor) and binds it.  The pointer must be 256-byte aligned. */
size_t fd_plan_workspace_bytes(const fd_plan *plan);
int fd_plan_bind_workspace(fd_plan *plan, void *device_ptr, size_t bytes);

/* Folds BatchNorm (inference form, running statistics) into each conv and repacks the weights into
 * the kernels' layouts, on `stream`.  Must be repeated whenever the parameters change. */
int fd_plan_pack_weights(fd_plan *plan, const fd_layer_params *params, int32_t n_layers, float bn_eps,
                         void *stream);

/* Inference forward: x [B,3,H,W] float32 NCHW -> y [B,cout_last,H,W] float32, enqueued on `stream`. */
int fd_forward(fd_plan *plan, const void *x_nchw, void *y, void *stream);

/* Same as fd_forward, but launches every layer's kernel with begin/end HIP events (hipExtLaunchKernelGGL) on
 * `stream`, synchronises the stream and returns each kernel's device duration in milliseconds
 * (ms_per_layer[n_layers]) -- the same quantity rocprofv3's kernel trace reports.  Measurement aid for
 * bench.py's roofline report; not for production calls. */
int fd_forward_timed(fd_plan *plan, const void *x_nchw, void *y, void *stream, float *ms_per_layer, int32_t n_layers);

/* Deploy bundle: the plan's layer descriptions plus its packed, BatchNorm-folded weights in one self-describing
 * host buffer -- the analogue of the TVM artefacts the reference's runner loads (deploy/tx2_run_tvm.py:13-20: deploy_graph.json +
 * deploy_param.params).  A C program needs nothing else: fd_plan_import -> fd_plan_workspace_bytes -> hipMalloc ->
 * fd_plan_bind_workspace -> fd_plan_import_weights -> fd_forward (examples/run_bundle.cpp).  The packed weights do not depend on the
 * batch size: batch_override > 0 re-plans the bundle for another batch. */
size_t fd_plan_export_bytes(const fd_plan *plan);
int fd_plan_export(const fd_plan *plan, void *host_buffer, size_t bytes, void *stream);
int fd_plan_import(const void *host_buffer, size_t bytes, int32_t batch_override, fd_plan **out_plan);
int fd_plan_import_weights(fd_plan *plan, const void *host_buffer, size_t bytes, void *stream);
int fd_plan_shape(const fd_plan *plan, int32_t *batch, int32_t *height, int32_t *width, int32_t *dtype);

/* Measurement aid for bench.py's roofline report (inference AND train step): between fd_trace_begin() and fd_trace_end() every kernel
 * this thread launches through the library carries its own begin/end HIP events (hipExtLaunchKernelGGL -- the quantity rocprofv3's
 * kernel trace reports).  fd_trace_end synchronises, fills up to max_records records in launch order and returns the total number of
 * launches in *n_records.  `kernel` is the source spelling of the launched kernel (a string owned by the library), `layer` the index of
 * the fused layer it belongs to (-1: loss, SGD, weight packing).  Not for production calls. */
typedef struct fd_trace_record { const char *kernel; int32_t layer; float ms; } fd_trace_record;
int fd_trace_begin(void);
int fd_trace_end(void *stream, fd_trace_record *records, int32_t max_records, int32_t *n_records);

/* Test hook: where layer `layer`'s output lives (NHWC, plan dtype).  Valid after fd_forward on a
 * plan created with FD_PLAN_KEEP_ACTIVATIONS. */
int fd_layer_output(const fd_plan *plan, int32_t layer, const void **device_ptr, int32_t *n, int32_t *h,
                    int32_t *w, int32_t *c);

/* Number of kernels one fd_forward enqueues, and a human-readable description of layer i's kernel
 * choice ("pw_gemm_f32<128x64> grid=... lds=...").  The string is owned by the plan. */
int32_t fd_plan_num_kernels(const fd_plan *plan);
const char *fd_plan_kernel_info(const fd_plan *plan, int32_t layer);
/* Demangled name of the __global__ function layer i launches, spelled as rocprofv3 prints it
 * (e.g. "fd_pw_gemm_f32<2, 2, 1, 1, 2>"), so bench.py's per-kernel timings can be matched to a kernel trace. */
const char *fd_plan_kernel_symbol(const fd_plan *plan, int32_t layer);

/* Algorithmic HBM bytes of one forward (SURVEY.md 8(d) convention: every fused unit reads its stored
 * inputs once and writes its output once; weights + folded BN once per batch). */
double fd_plan_algorithmic_bytes(const fd_plan *plan);
double fd_plan_algorithmic_flops(const fd_plan *plan);
/* The same two figures for one layer's kernel launch. */
int fd_plan_layer_stats(const fd_plan *plan, int32_t layer, double *algorithmic_bytes, double *algorithmic_flops);

/* HBM bytes the launch of fused layer `layer` has to move (stored inputs + outputs + weights): equals the algorithmic bytes of
 * fd_plan_layer_stats for an unfused launch; a fused launch does not move the intermediate tensors it keeps on chip.  0 for a layer that
 * runs inside another layer's launch.  (bench.py prices per-kernel bandwidth and the fused-plan roofline bound with it.) */
int fd_plan_layer_traffic(const fd_plan *plan, int32_t layer, double *needed_bytes);

/* ------------------------------------------------------------------------------------------------------------
 * Train step (NOT in the reference tree -- README.md:65 names the upstream it was stripped from; defined in
 * SURVEY.md section 3(4) as: module in .train() -> L1 loss -> backward -> SGD(momentum, weight decay), data-parallel
 * gradient mean).  BatchNorm uses batch statistics (biased variance to normalise, unbiased into running_var,
 * momentum applied in place to the caller's running_mean / running_var tensors).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct fd_train_plan fd_train_plan;

/* device pointers to the gradient tensors of one layer (fp32, same layouts as the parameters) */
typedef struct fd_layer_grads {
    float *conv_weight;
    float *bn_weight;
    float *bn_bias;
} fd_layer_grads;

/* dtype: FD_F32, or FD_BF16 = the saved conv outputs z, the activation gradients G and the operands of the pointwise matrix
 * products are stored in bfloat16 (v_mfma_f32_32x32x16_bf16, fp32 accumulation); master parameters, their gradients, batch
 * statistics, BatchNorm tables, the 1-channel head, loss and SGD stay fp32.  Channel counts must
 * then be multiples of 8.  FD_F16 is rejected (fp16 gradients would need loss scaling). */
int fd_train_plan_create(const fd_layer_desc *layers, int32_t n_layers, int32_t batch, int32_t height, int32_t width,
                         int32_t dtype, uint32_t flags, fd_train_plan **out_plan);
void fd_train_plan_destroy(fd_train_plan *plan);
size_t fd_train_plan_workspace_bytes(const fd_train_plan *plan);
int fd_train_plan_bind_workspace(fd_train_plan *plan, void *device_ptr, size_t bytes);

/* Train-mode forward: reads the LIVE parameters (no folding), saves every unit's raw conv output and batch
 * statistics in the workspace for fd_train_backward, updates running_mean / running_var / num_batches_tracked in place,
 * writes y [B,1,H,W]. */
int fd_train_forward(fd_train_plan *plan, const fd_layer_params *params, int32_t n_layers, float bn_eps,
                     float bn_momentum, const void *x_nchw, void *y, void *stream);

/* Backward of the last fd_train_forward: dy = dLoss/dy [B,1,H,W] fp32; writes (overwrites) every parameter gradient.
 * The network input x passed to fd_train_forward must still be valid (the stem's weight gradient re-reads it). */
int fd_train_backward(fd_train_plan *plan, const fd_layer_params *params, const fd_layer_grads *grads, int32_t n_layers,
                      const void *dy, void *stream);

/* Same, restricted to the units from_layer >= i >= to_layer (from_layer must continue where the previous call stopped;
 * the first call starts at n_layers-1).  Lets the host interleave gradient all-reduce buckets with the backward pass: when a call returns,
 * every gradient of ITS units (conv weight, BatchNorm weight / bias) has been enqueued -- nothing of a later unit is promised (the BatchNorm
 * backward of unit to_layer - 1 may be finalised by that unit's own first kernel, i.e. by the next call). */
int fd_train_backward_range(fd_train_plan *plan, const fd_layer_params *params, const fd_layer_grads *grads, int32_t n_layers,
                            const void *dy, int32_t from_layer, int32_t to_layer, void *stream);

/* Test hook: raw conv output z (which == 0) or dLoss/d(BN output) G (which == 1) of a layer, NHWC in the plan's dtype (the
 * 1-channel head is fp32 in every plan); which == 2: the layer's BatchNorm table as n=1, h=4 (scale, shift, mean, invstd),
 * w=1, c=C, fp32; which == 3: the decoder->skip gradient buffer of a skip source; which == 4: the BatchNorm-backward output dz
 * of a 16-bit pointwise unit (only plans created with FD_PLAN_KEEP_ACTIVATIONS keep G and dz apart and keep per-layer G). */
int fd_train_layer_tensor(const fd_train_plan *plan, int32_t layer, int32_t which, const void **device_ptr, int32_t *n,
                          int32_t *h, int32_t *w, int32_t *c);

/* Mean-L1 loss, forward + backward in one pass (torch.nn.L1Loss()): loss_out[0] = mean|pred - target| (device
 * float), dpred = sign(pred - target) / numel.  `scratch` needs fd_l1_loss_scratch_bytes(numel) bytes. */
size_t fd_l1_loss_scratch_bytes(int64_t numel);
int fd_l1_loss(const void *pred, const void *target, void *dpred, float *loss_out, int64_t numel, void *scratch, void *stream);

/* Masked mean-L1 loss: the criterion of the upstream train script the reference was cut from (README.md:65, sparse-to-dense's
 * MaskedL1Loss: valid = target > 0; loss = mean over the valid pixels of |pred - target|) -- NYU depth maps contain invalid zeros, the
 * reason reference metrics.py:32 masks them as well.  dpred = sign(pred - target) / #valid on the valid pixels, 0 elsewhere; with no
 * valid pixel the loss is NaN (mean of an empty selection) and dpred is all zeros.  Same scratch as fd_l1_loss. */
int fd_l1_loss_masked(const void *pred, const void *target, void *dpred, float *loss_out, int64_t numel, void *scratch, void *stream);

/* Fused multi-tensor SGD (torch.optim.SGD semantics: d = grad_scale*g + wd*p; buf = mom*buf + d (buf = d on the first
 * step); p -= lr*buf) over n tensors in ONE launch.  `table` is a device array of n fd_sgd_tensor records; grad_scale
 * is 1/world_size after a summing all-reduce (data-parallel mean), 1 otherwise. */
typedef struct fd_sgd_tensor { float *param; const float *grad; float *momentum_buf; int64_t numel; } fd_sgd_tensor;
int fd_sgd_step(const fd_sgd_tensor *table_device, int32_t n_tensors, int64_t total_numel, float lr, float momentum,
                float weight_decay, float grad_scale, int32_t first_step, void *stream);

/* Data-parallel gradient exchange in 16 bits (optional; SURVEY.md 8(e): 7.92 MB instead of 15.84 MB over xGMI): converts `numel` values of
 * a bucket of the flat gradient vector fp32 -> bfloat16 (to_bf16 != 0, round to nearest even) before the summing all-reduce, or bfloat16 ->
 * fp32 after it.  The collective itself stays the host's (torch.distributed / RCCL), as for the fp32 exchange. */
int fd_cast_gradients(const void *src, void *dst, int64_t numel, int32_t to_bf16, void *stream);

/* ---- Data-parallel gradient exchange issued by the LIBRARY (the reference's only multi-GPU idiom is torch.nn.DataParallel,
 * imagenet/mobilenet.py:68): one process per GPU, RCCL all-reduce of the flat gradient vector over xGMI, bucket by bucket on the communicator's own
 * HIP stream, each bucket released by an event the moment its last backward kernel has been enqueued -- no host round trip per bucket (the
 * torch.distributed route of rounds 1-3 cost a Python call, an event, a stream switch and a ProcessGroup work object per bucket).  RCCL is bound
 * at run time (dlopen librccl.so.1), the library does not link it.
 *   fd_comm_unique_id      rank 0 makes the 128-byte rendezvous id (ncclGetUniqueId); the host broadcasts it by whatever means it has
 *                          (torch.distributed.broadcast in train.py, MPI / a file in a C host)
 *   fd_comm_create         every rank: ncclCommInitRank on its current device + the communication stream and events
 *   fd_train_backward_allreduce
 *                          = fd_train_backward_range over `buckets` (which must tile layers n-1 .. 0 in backward order), each followed by the summing
 *                          all-reduce of its contiguous fp32 gradient slice grad[numel] in place -- or, when grad16 != NULL, of a bfloat16 copy
 *                          (fd_cast_gradients -> all-reduce -> cast back: half the bytes over xGMI); on return `stream` waits for the last collective,
 *                          so fd_sgd_step(grad_scale = 1 / world) can be enqueued right behind
 *   fd_comm_last_exchange_ms
 *                          measurement aid (synchronises): device time from the first collective's issue to the last one's end in the most recent
 *                          call, and how much of it was left after the last backward kernel had finished (the part backward does not hide) */
#define FD_COMM_ID_BYTES 128
typedef struct fd_comm fd_comm;
typedef struct fd_grad_bucket { int32_t from_layer, to_layer; float *grad; int64_t numel; void *grad16; } fd_grad_bucket;
int fd_comm_unique_id(void *id_out);
int fd_comm_create(const void *id, int32_t rank, int32_t world, fd_comm **out);
void fd_comm_destroy(fd_comm *comm);
int fd_train_backward_allreduce(fd_train_plan *plan, const fd_layer_params *params, const fd_layer_grads *grads, int32_t n_layers, const void *dy,
                                fd_comm *comm, const fd_grad_bucket *buckets, int32_t n_buckets, void *stream);
int fd_comm_last_exchange_ms(fd_comm *comm, float *ms_first_issue_to_last_done, float *ms_exposed_after_backward);

/* Depth metrics (reference metrics.py:31-55 Result.evaluate): one fused reduction over output/target
 * (fp32, any shape, `numel` elements) producing the 10 sums from which every metric follows:
 *   sums[0] #valid, [1] sum ad^2, [2] sum ad, [3] sum |log10 o - log10 t|, [4] sum ad/t, [5..7] #(max(o/t,t/o) < 1.25^k),
 *   [8] sum (1/o-1/t)^2, [9] sum |1/o-1/t|      with valid = (target>0)|(output>0), o = 1e3*output, t = 1e3*target.
 * `sums_device` = 10 doubles on the device; `scratch` needs fd_depth_metrics_scratch_bytes() bytes.  The reference issues ~12
 * device->host synchronisations per sample for the same numbers (one float() per metric). */
/* Input preparation (reference dataloaders/nyu.py:48-59 val_transform: Resize(250/480) -> CenterCrop(228,304)
 * -> Resize(output_size), nearest-neighbour, then /255): gathers n raw frames rgb[n][H][W][3] (uint8) and, if depth != NULL,
 * depth[n][H][W] (fp32) through a row table ymap[out_h] and a column table xmap[out_w] (device int32 arrays holding source
 * indices; the host composes the three steps, fast-depth_amd/dataloaders/nyu.py) into the network input x[n][3][out_h][out_w]
 * (fp32, value/255 computed in double as the reference does) and depth_out[n][1][out_h][out_w]. */
int fd_val_transform(const void *rgb_u8, const float *depth, int32_t n, int32_t height, int32_t width, int32_t out_h, int32_t out_w,
                     const int32_t *ymap_device, const int32_t *xmap_device, float *x_out, float *depth_out, void *stream);

size_t fd_depth_metrics_scratch_bytes(void);
int fd_depth_metrics(const void *output, const void *target, int64_t numel, double *sums_device, void *scratch, void *stream);

/* The same ten sums PER FRAME: output / target hold n_frames images of frame_numel elements each, sums_device = n_frames x 10 doubles.
 * The reference evaluates one image at a time and averages the per-image metrics (main.py:40-41 batch size 1, :80-82) -- RMSE and
 * iRMSE of a pooled batch differ from the mean of the per-image values -- so a batched evaluation loop needs the sums per image. */
size_t fd_depth_metrics_frames_scratch_bytes(int32_t n_frames);
int fd_depth_metrics_frames(const void *output, const void *target, int32_t n_frames, int64_t frame_numel, double *sums_device,
                            void *scratch, void *stream);

const char *fd_last_error(void);
const char *fd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FASTDEPTH_HIP_H */
