"""ctypes declarations for include/fastdepth_hip.h.  No torch import here."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libfastdepth_hip.so")

FD_F32, FD_F16, FD_BF16 = 0, 1, 2


class _DtypeMap(dict):
    def __missing__(self, key):
        import torch
        m = {torch.float32: FD_F32, torch.float16: FD_F16, torch.bfloat16: FD_BF16}
        self.update(m)
        return m[key]


DTYPE_OF = _DtypeMap()      # torch dtype -> fd_dtype (torch imported lazily: this module is also used by tests without torch tensors)
FD_OP_STEM, FD_OP_DW, FD_OP_PW = 0, 1, 2
FD_ACT_NONE, FD_ACT_RELU, FD_ACT_RELU6 = 0, 1, 2
# public plan flags (include/fastdepth_hip.h)
FD_PLAN_KEEP_ACTIVATIONS = 1
FD_PLAN_NO_GEMM16 = 64
FD_PLAN_NO_ROWS8 = 256
FD_PLAN_NO_EPILOGUE_FUSION = 512
FD_PLAN_NO_UNIT_FUSION = 1024
FD_PLAN_NO_BWD_PAIRING = 4096
# private tuning / test mask (csrc/fd_tuning.h), handed to the library through fd_tuning_next before the plan is created.  The Python
# mirrors are shifted above bit 32 so that ONE `flags=` integer can carry both masks: create_plan() below splits it.
_TUNE_SHIFT = 32
FD_TUNE_WGRAD_TILE_ROWS = 1 << _TUNE_SHIFT
FD_TUNE_FORCE_GEMM16 = 2 << _TUNE_SHIFT
FD_TUNE_FORCE_EPILOGUE_FUSION = 4 << _TUNE_SHIFT
FD_TUNE_FORCE_UNIT_FUSION = 8 << _TUNE_SHIFT
FD_TUNE_NO_PW_PAIRING = 16 << _TUNE_SHIFT
FD_TUNE_PW_PAIR_TN2 = 32 << _TUNE_SHIFT
FD_TUNE_DW_BWD1 = 64 << _TUNE_SHIFT
FD_TUNE_DW_BWD_PAIR = 128 << _TUNE_SHIFT
FD_TUNE_DW_SMALL_TILES = 256 << _TUNE_SHIFT
FD_TUNE_DW_PITCH4 = 512 << _TUNE_SHIFT
FD_TUNE_DW_PITCH8 = 1024 << _TUNE_SHIFT
FD_TUNE_DW_WGRAD_TH4 = 2048 << _TUNE_SHIFT
FD_TUNE_DW_NO_ROWS = 4096 << _TUNE_SHIFT
FD_TUNE_DW_FORCE_ROWS = 8192 << _TUNE_SHIFT
FD_TUNE_DW_TH8 = 16384 << _TUNE_SHIFT
FD_TUNE_DW_CB16 = 32768 << _TUNE_SHIFT
FD_TUNE_NO_DW_H8 = 65536 << _TUNE_SHIFT
FD_TUNE_FORCE_DW_H8 = 131072 << _TUNE_SHIFT
FD_TUNE_NO_CONSUMER_FINALIZE = 262144 << _TUNE_SHIFT
FD_TUNE_DW_BWD_FINALIZE = 524288 << _TUNE_SHIFT
FD_TUNE_NO_DW5_ROWS = 1048576 << _TUNE_SHIFT


def create_plan(lib, train, descs, n, batch, height, width, fd_dtype, flags, handle_ref):
    """fd_plan_create / fd_train_plan_create with a combined flags integer: the low 32 bits are the public plan flags, the bits above
    are the private tuning mask (FD_TUNE_*), passed through fd_tuning_next -- which only this thread's next creation consumes."""
    tuning = int(flags) >> _TUNE_SHIFT
    lib.fd_tuning_next(tuning)         # always, 0 included: a mask an earlier caller left in this thread's slot must not reach an unrelated plan
    fn = lib.fd_train_plan_create if train else lib.fd_plan_create
    return fn(descs, n, batch, height, width, fd_dtype, int(flags) & 0xFFFFFFFF, handle_ref)


def parse_flags(text):
    """'NO_DW_H8|NO_BWD_PAIRING' / '0x200' / '' -> the combined flags integer create_plan() takes (names without their FD_PLAN_ / FD_TUNE_ prefix)."""
    v = 0
    for tok in str(text).replace(",", "|").split("|"):
        tok = tok.strip()
        if not tok:
            continue
        g = globals()
        if "FD_PLAN_" + tok in g:
            v |= g["FD_PLAN_" + tok]
        elif "FD_TUNE_" + tok in g:
            v |= g["FD_TUNE_" + tok]
        else:
            v |= int(tok, 0)
    return v


class LayerDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("op", "cin", "cout", "ksize", "stride", "act", "src", "upsample", "skip", "concat")]


class LayerParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("conv_weight", "bn_weight", "bn_bias", "bn_mean", "bn_var", "bn_num_batches_tracked")]


class LayerGrads(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("conv_weight", "bn_weight", "bn_bias")]


class GradBucket(ctypes.Structure):
    _fields_ = [("from_layer", ctypes.c_int32), ("to_layer", ctypes.c_int32), ("grad", ctypes.c_void_p), ("numel", ctypes.c_int64), ("grad16", ctypes.c_void_p)]


class TraceRecord(ctypes.Structure):
    _fields_ = [("kernel", ctypes.c_char_p), ("layer", ctypes.c_int32), ("ms", ctypes.c_float)]


class SgdTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("momentum_buf", ctypes.c_void_p), ("numel", ctypes.c_int64)]


class FastDepthError(RuntimeError):
    pass


def load(path=None):
    """Loads the shared library and declares every entry point of include/fastdepth_hip.h.
    Raises (never falls back) if the library is missing: the product has no other execution path."""
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise FastDepthError("HIP extension not built: %s is missing (run `python fast-depth_amd/build.py`)" % path)
    lib = ctypes.CDLL(path)
    vp, i32, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32
    lib.fd_plan_create.argtypes = [ctypes.POINTER(LayerDesc), i32, i32, i32, i32, i32, u32, ctypes.POINTER(vp)]
    lib.fd_plan_create.restype = ctypes.c_int
    lib.fd_tuning_next.argtypes = [u32]       # private hook (csrc/fd_tuning.h), not part of include/fastdepth_hip.h
    lib.fd_tuning_next.restype = None
    lib.fd_plan_destroy.argtypes = [vp]
    lib.fd_plan_destroy.restype = None
    lib.fd_plan_workspace_bytes.argtypes = [vp]
    lib.fd_plan_workspace_bytes.restype = ctypes.c_size_t
    lib.fd_plan_bind_workspace.argtypes = [vp, vp, ctypes.c_size_t]
    lib.fd_plan_bind_workspace.restype = ctypes.c_int
    lib.fd_plan_pack_weights.argtypes = [vp, ctypes.POINTER(LayerParams), i32, ctypes.c_float, vp]
    lib.fd_plan_pack_weights.restype = ctypes.c_int
    lib.fd_forward.argtypes = [vp, vp, vp, vp]
    lib.fd_forward.restype = ctypes.c_int
    lib.fd_forward_timed.argtypes = [vp, vp, vp, vp, ctypes.POINTER(ctypes.c_float), i32]
    lib.fd_forward_timed.restype = ctypes.c_int
    lib.fd_layer_output.argtypes = [vp, i32, ctypes.POINTER(vp)] + [ctypes.POINTER(i32)] * 4
    lib.fd_layer_output.restype = ctypes.c_int
    lib.fd_plan_num_kernels.argtypes = [vp]
    lib.fd_plan_num_kernels.restype = i32
    lib.fd_plan_kernel_info.argtypes = [vp, i32]
    lib.fd_plan_kernel_info.restype = ctypes.c_char_p
    lib.fd_plan_kernel_symbol.argtypes = [vp, i32]
    lib.fd_plan_kernel_symbol.restype = ctypes.c_char_p
    lib.fd_plan_algorithmic_bytes.argtypes = [vp]
    lib.fd_plan_algorithmic_bytes.restype = ctypes.c_double
    lib.fd_plan_algorithmic_flops.argtypes = [vp]
    lib.fd_plan_algorithmic_flops.restype = ctypes.c_double
    lib.fd_plan_layer_stats.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.fd_plan_layer_stats.restype = ctypes.c_int
    lib.fd_plan_layer_traffic.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double)]
    lib.fd_plan_layer_traffic.restype = ctypes.c_int
    if hasattr(lib, "fd_train_plan_create"):
        lib.fd_train_plan_create.argtypes = [ctypes.POINTER(LayerDesc), i32, i32, i32, i32, i32, u32, ctypes.POINTER(vp)]
        lib.fd_train_plan_create.restype = ctypes.c_int
        lib.fd_train_plan_destroy.argtypes = [vp]
        lib.fd_train_plan_destroy.restype = None
        lib.fd_train_plan_workspace_bytes.argtypes = [vp]
        lib.fd_train_plan_workspace_bytes.restype = ctypes.c_size_t
        lib.fd_train_plan_bind_workspace.argtypes = [vp, vp, ctypes.c_size_t]
        lib.fd_train_plan_bind_workspace.restype = ctypes.c_int
        lib.fd_train_forward.argtypes = [vp, ctypes.POINTER(LayerParams), i32, ctypes.c_float, ctypes.c_float, vp, vp, vp]
        lib.fd_train_forward.restype = ctypes.c_int
        lib.fd_train_layer_tensor.argtypes = [vp, i32, i32, ctypes.POINTER(vp)] + [ctypes.POINTER(i32)] * 4
        lib.fd_train_layer_tensor.restype = ctypes.c_int
    if hasattr(lib, "fd_train_backward"):
        lib.fd_train_backward.argtypes = [vp, ctypes.POINTER(LayerParams), ctypes.POINTER(LayerGrads), i32, vp, vp]
        lib.fd_train_backward.restype = ctypes.c_int
        lib.fd_train_backward_range.argtypes = [vp, ctypes.POINTER(LayerParams), ctypes.POINTER(LayerGrads), i32, vp, i32, i32, vp]
        lib.fd_train_backward_range.restype = ctypes.c_int
        lib.fd_l1_loss_scratch_bytes.argtypes = [ctypes.c_int64]
        lib.fd_l1_loss_scratch_bytes.restype = ctypes.c_size_t
        lib.fd_l1_loss.argtypes = [vp, vp, vp, vp, ctypes.c_int64, vp, vp]
        lib.fd_l1_loss.restype = ctypes.c_int
        lib.fd_l1_loss_masked.argtypes = [vp, vp, vp, vp, ctypes.c_int64, vp, vp]
        lib.fd_l1_loss_masked.restype = ctypes.c_int
        lib.fd_sgd_step.argtypes = [vp, i32, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, i32, vp]
        lib.fd_sgd_step.restype = ctypes.c_int
        lib.fd_cast_gradients.argtypes = [vp, vp, ctypes.c_int64, i32, vp]
        lib.fd_cast_gradients.restype = ctypes.c_int
        lib.fd_comm_unique_id.argtypes = [vp]
        lib.fd_comm_unique_id.restype = ctypes.c_int
        lib.fd_comm_create.argtypes = [vp, i32, i32, ctypes.POINTER(vp)]
        lib.fd_comm_create.restype = ctypes.c_int
        lib.fd_comm_destroy.argtypes = [vp]
        lib.fd_comm_destroy.restype = None
        lib.fd_train_backward_allreduce.argtypes = [vp, ctypes.POINTER(LayerParams), ctypes.POINTER(LayerGrads), i32, vp, vp, ctypes.POINTER(GradBucket), i32, vp]
        lib.fd_train_backward_allreduce.restype = ctypes.c_int
        lib.fd_comm_last_exchange_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        lib.fd_comm_last_exchange_ms.restype = ctypes.c_int
    lib.fd_val_transform.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.fd_val_transform.restype = ctypes.c_int
    lib.fd_depth_metrics_scratch_bytes.argtypes = []
    lib.fd_depth_metrics_scratch_bytes.restype = ctypes.c_size_t
    lib.fd_depth_metrics.argtypes = [vp, vp, ctypes.c_int64, vp, vp, vp]
    lib.fd_depth_metrics.restype = ctypes.c_int
    lib.fd_depth_metrics_frames_scratch_bytes.argtypes = [i32]
    lib.fd_depth_metrics_frames_scratch_bytes.restype = ctypes.c_size_t
    lib.fd_depth_metrics_frames.argtypes = [vp, vp, i32, ctypes.c_int64, vp, vp, vp]
    lib.fd_depth_metrics_frames.restype = ctypes.c_int
    lib.fd_plan_export_bytes.argtypes = [vp]
    lib.fd_plan_export_bytes.restype = ctypes.c_size_t
    lib.fd_plan_export.argtypes = [vp, vp, ctypes.c_size_t, vp]
    lib.fd_plan_export.restype = ctypes.c_int
    lib.fd_plan_import.argtypes = [vp, ctypes.c_size_t, i32, ctypes.POINTER(vp)]
    lib.fd_plan_import.restype = ctypes.c_int
    lib.fd_plan_import_weights.argtypes = [vp, vp, ctypes.c_size_t, vp]
    lib.fd_plan_import_weights.restype = ctypes.c_int
    lib.fd_plan_shape.argtypes = [vp] + [ctypes.POINTER(i32)] * 4
    lib.fd_plan_shape.restype = ctypes.c_int
    lib.fd_trace_begin.argtypes = []
    lib.fd_trace_begin.restype = ctypes.c_int
    lib.fd_trace_end.argtypes = [vp, ctypes.POINTER(TraceRecord), i32, ctypes.POINTER(i32)]
    lib.fd_trace_end.restype = ctypes.c_int
    lib.fd_last_error.restype = ctypes.c_char_p
    lib.fd_version.restype = ctypes.c_char_p
    return lib


EXPORTS = ("fd_plan_create", "fd_plan_destroy", "fd_plan_workspace_bytes", "fd_plan_bind_workspace",
           "fd_plan_pack_weights", "fd_forward", "fd_forward_timed", "fd_layer_output", "fd_plan_num_kernels", "fd_plan_kernel_info", "fd_plan_kernel_symbol",
           "fd_plan_algorithmic_bytes", "fd_plan_algorithmic_flops", "fd_plan_layer_stats", "fd_plan_layer_traffic",
           "fd_train_plan_create", "fd_train_plan_destroy", "fd_train_plan_workspace_bytes", "fd_train_plan_bind_workspace",
           "fd_train_forward", "fd_train_backward", "fd_train_backward_range", "fd_train_layer_tensor", "fd_l1_loss_scratch_bytes",
           "fd_l1_loss", "fd_l1_loss_masked", "fd_sgd_step", "fd_cast_gradients", "fd_comm_unique_id", "fd_comm_create", "fd_comm_destroy",
           "fd_train_backward_allreduce", "fd_comm_last_exchange_ms", "fd_val_transform", "fd_depth_metrics_scratch_bytes", "fd_depth_metrics",
           "fd_depth_metrics_frames_scratch_bytes", "fd_depth_metrics_frames", "fd_plan_export_bytes", "fd_plan_export",
           "fd_plan_import", "fd_plan_import_weights", "fd_plan_shape", "fd_trace_begin", "fd_trace_end", "fd_last_error", "fd_version")


def check(lib, rc, what):
    if rc != 0:
        raise FastDepthError("%s failed (%d): %s" % (what, rc, lib.fd_last_error().decode()))
