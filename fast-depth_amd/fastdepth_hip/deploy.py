"""Deploy bundle + checkpoint I/O.

  export_bundle(model, path, batch, size, dtype)   plan + packed, BatchNorm-folded weights in one self-describing file -- what the reference
                                                   hands its TX2 runner as TVM graph + params (deploy/tx2_run_tvm.py:13-20)
  BundleRunner(path, batch)                        loads such a file WITHOUT the nn.Module (no parameters, no pickle): fd_plan_import ->
                                                   workspace -> fd_plan_import_weights -> fd_forward; examples/run_bundle.cpp is the same
                                                   sequence in C++ with no Python at all
  save_checkpoint / load_checkpoint                the reference's {'epoch', 'best_result', 'model'} pickle with the module stored whole
                                                   (main.py:49-57 reads exactly this), and plain state_dict files
"""
import ctypes
import os

import torch

from . import capi
from .engine import lib, _Plan


def export_bundle(model, path, batch=1, size=(224, 224), dtype=torch.float32):
    """Writes the deploy bundle of `model` (on the GPU, eval form) for inputs [batch, 3, size[0], size[1]]."""
    eng = model._engine()
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise capi.FastDepthError("export_bundle needs the model on the GPU (the weights are packed there)")
    old = eng.dtype
    eng.set_dtype(dtype)
    try:
        plan = _Plan(eng, batch, size[0], size[1], dev, False)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            eng._pack(plan, stream)
            L = lib()
            n = L.fd_plan_export_bytes(plan.handle)
            buf = (ctypes.c_ubyte * n)()
            capi.check(L, L.fd_plan_export(plan.handle, buf, n, stream), "fd_plan_export")
    finally:
        eng.set_dtype(old)
    with open(path, "wb") as f:
        f.write(bytes(buf))
    return n


class BundleRunner:
    """Inference from a deploy bundle alone."""

    def __init__(self, path, batch=0, device="cuda:0", _library=None):
        self.L = L = _library or lib()
        self.device = torch.device(device)
        self.blob = open(path, "rb").read()
        self.buf = (ctypes.c_ubyte * len(self.blob)).from_buffer_copy(self.blob)
        self.handle = ctypes.c_void_p()
        capi.check(L, L.fd_plan_import(self.buf, len(self.blob), batch, ctypes.byref(self.handle)), "fd_plan_import")
        dims = [ctypes.c_int32() for _ in range(4)]
        capi.check(L, L.fd_plan_shape(self.handle, *[ctypes.byref(d) for d in dims]), "fd_plan_shape")
        self.batch, self.height, self.width, self.dtype = [d.value for d in dims]
        nbytes = L.fd_plan_workspace_bytes(self.handle)
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = (self.workspace.data_ptr() + 255) // 256 * 256
        capi.check(L, L.fd_plan_bind_workspace(self.handle, base, nbytes), "fd_plan_bind_workspace")
        capi.check(L, L.fd_plan_import_weights(self.handle, self.buf, len(self.blob), self._stream()), "fd_plan_import_weights")

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None

    def __call__(self, x):
        if tuple(x.shape) != (self.batch, 3, self.height, self.width) or x.dtype != torch.float32 or x.device != self.device:
            raise capi.FastDepthError("this bundle is planned for float32 [%d,3,%d,%d] on %s" % (self.batch, self.height, self.width, self.device))
        x = x.contiguous()
        y = torch.empty((self.batch, 1, self.height, self.width), dtype=torch.float32, device=self.device)
        capi.check(self.L, self.L.fd_forward(self.handle, x.data_ptr(), y.data_ptr(), self._stream()), "fd_forward")
        return y

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                self.L.fd_plan_destroy(self.handle)
            except Exception:
                pass
            self.handle = None


def save_checkpoint(model, epoch, best_result, path):
    """The reference's checkpoint format: a dict with the module pickled whole (its `validate` entry point loads exactly this:
    main.py:49-57 `checkpoint['epoch'], checkpoint['best_result'], checkpoint['model']`).  The execution engine and its device
    workspaces are not part of the pickle (models._HipForward.__getstate__)."""
    torch.save({"epoch": epoch, "best_result": best_result, "model": model}, path)
    return path


def load_checkpoint(path, device=None):
    """Returns (model, epoch, best_result) from a reference-format checkpoint (dict with 'model'), a bare pickled module, or a plain
    state_dict / {'state_dict': ...} file (loaded into a fresh MobileNetSkipAdd)."""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    epoch, best = 0, None
    if isinstance(ck, dict) and "model" in ck:
        model, epoch, best = ck["model"], ck.get("epoch", 0), ck.get("best_result")
    elif isinstance(ck, torch.nn.Module):
        model = ck
    else:
        import models
        sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}       # DataParallel prefix (models.py:664-669)
        model = models.MobileNetSkipAdd((224, 224), pretrained=False)
        model.load_state_dict(sd)
    if device is not None:
        model = model.to(device)
    return model, epoch, best
