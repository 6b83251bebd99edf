"""Drives the C ABI (include/fastdepth_hip.h) directly, for either the real library on `cuda` tensors or the
CPU-emulation build on CPU tensors.  Test infrastructure: the product's Engine never accepts CPU tensors."""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd"))
from fastdepth_hip import capi  # noqa: E402
from fastdepth_hip.plan import layers_of  # noqa: E402

_libs = {}


def get_lib(kind):
    if kind not in _libs:
        if kind == "emu":
            sys.path.insert(0, os.path.join(REPO, "tests", "hipemu"))
            import build_emu
            _libs[kind] = capi.load(build_emu.build())
        else:
            _libs[kind] = capi.load()
    return _libs[kind]


class CPlan:
    """fd_plan + workspace for `model` at x's shape on x's device."""

    def __init__(self, kind, model, x, keep=True):
        self.lib = L = get_lib(kind)
        self.kind, self.model, self.dev = kind, model, x.device
        self.layers = layers_of(model)
        n = len(self.layers)
        descs = (capi.LayerDesc * n)(*[l.desc for l in self.layers])
        self.h = ctypes.c_void_p()
        b, _, hh, ww = x.shape
        capi.check(L, L.fd_plan_create(descs, n, b, hh, ww, capi.FD_F32, capi.FD_PLAN_KEEP_ACTIVATIONS if keep else 0,
                                       ctypes.byref(self.h)), "fd_plan_create")
        nbytes = L.fd_plan_workspace_bytes(self.h)
        self.ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.dev)
        self.base = (self.ws.data_ptr() + 255) // 256 * 256
        capi.check(L, L.fd_plan_bind_workspace(self.h, self.base, nbytes), "fd_plan_bind_workspace")
        self.stream = torch.cuda.current_stream().cuda_stream if x.is_cuda else None
        params = (capi.LayerParams * n)()
        self._keepalive = []
        for q, l in zip(params, self.layers):
            for name, t in (("conv_weight", l.conv.weight), ("bn_weight", l.bn.weight), ("bn_bias", l.bn.bias),
                            ("bn_mean", l.bn.running_mean), ("bn_var", l.bn.running_var)):
                t = t.detach().to(self.dev, torch.float32).contiguous()
                self._keepalive.append(t)
                setattr(q, name, t.data_ptr())
        capi.check(L, L.fd_plan_pack_weights(self.h, params, n, self.layers[0].bn.eps, self.stream), "fd_plan_pack_weights")

    def forward(self, x):
        x = x.contiguous()
        y = torch.full((x.shape[0], 1, x.shape[2], x.shape[3]), float("nan"), dtype=torch.float32, device=self.dev)
        capi.check(self.lib, self.lib.fd_forward(self.h, x.data_ptr(), y.data_ptr(), self.stream), "fd_forward")
        if x.is_cuda:
            torch.cuda.synchronize()
        return y

    def tap(self, i):
        ptr = ctypes.c_void_p()
        d = [ctypes.c_int32() for _ in range(4)]
        capi.check(self.lib, self.lib.fd_layer_output(self.h, i, ctypes.byref(ptr), *[ctypes.byref(v) for v in d]), "fd_layer_output")
        n, h, w, c = [v.value for v in d]
        off = ptr.value - self.ws.data_ptr()
        return self.ws[off:off + n * h * w * c * 4].view(torch.float32).view(n, h, w, c).permute(0, 3, 1, 2).contiguous().cpu()

    def info(self):
        return [self.lib.fd_plan_kernel_info(self.h, i).decode() for i in range(self.lib.fd_plan_num_kernels(self.h))]

    def close(self):
        if self.h:
            self.lib.fd_plan_destroy(self.h)
            self.h = None


def rel_err(a, b):
    """max |a-b| / max|b|  (the tolerance the north star states: 1e-3 relative, fp32)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def randomize_bn(model, seed):
    """Non-trivial BN affine + running statistics so that folding errors cannot hide."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
            m.bias.data.copy_(0.3 * torch.randn(m.bias.shape, generator=g))
            m.running_mean.copy_(0.2 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    return model


def compare_with_oracle(kind, model, x, device):
    """Runs the C ABI path and the C oracle on the same weights/input; returns (rel err of the output,
    [rel err per fused layer], plan info)."""
    from oracle import oracle
    model = model.eval()
    y_ref, taps_ref = oracle.forward(model.state_dict(), x.numpy(), taps=True)
    plan = CPlan(kind, model, x.to(device))
    y = plan.forward(x.to(device)).cpu().numpy()
    errs = [rel_err(plan.tap(i).numpy(), taps_ref[i]) for i in range(len(taps_ref) - 1)]
    errs.append(rel_err(y, y_ref))
    info = plan.info()
    plan.close()
    return rel_err(y, y_ref), errs, info
