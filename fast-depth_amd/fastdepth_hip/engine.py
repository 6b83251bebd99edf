"""torch-facing engine: owns one fd_plan per (batch, H, W) and the device workspace, enqueues on torch's
current HIP stream, repacks weights when the parameters' version counters move.

PyTorch is used for what the boundary leaves to the host: device memory (`torch.empty`), the current
stream, parameter storage.  All arithmetic happens in libfastdepth_hip.so.
"""
import ctypes

import torch

from . import capi
from .plan import layers_of

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = capi.load()          # raises if the extension is not built: no fallback
    return _LIB


_DTYPES = {torch.float32: capi.FD_F32, torch.float16: capi.FD_F16, torch.bfloat16: capi.FD_BF16}


class _Plan:
    def __init__(self, engine, batch, height, width, device, keep=False):
        L = lib()
        self.layers = engine.layers
        n = len(self.layers)
        descs = (capi.LayerDesc * n)(*[l.desc for l in self.layers])
        handle = ctypes.c_void_p()
        capi.check(L, capi.create_plan(L, False, descs, n, batch, height, width, _DTYPES[engine.dtype],
                                       (capi.FD_PLAN_KEEP_ACTIVATIONS if keep else 0) | engine.plan_flags, ctypes.byref(handle)), "fd_plan_create")
        self.dtype = engine.dtype
        self.handle = handle
        self.shape = (batch, height, width)
        self.workspace = torch.empty(L.fd_plan_workspace_bytes(handle) + 256, dtype=torch.uint8, device=device)
        base = (self.workspace.data_ptr() + 255) // 256 * 256
        capi.check(L, L.fd_plan_bind_workspace(handle, base, L.fd_plan_workspace_bytes(handle)), "fd_plan_bind_workspace")
        self.version = None

    def __del__(self):
        if getattr(self, "handle", None) and _LIB is not None:
            _LIB.fd_plan_destroy(self.handle)
            self.handle = None

    def kernel_info(self):
        L = lib()
        return [L.fd_plan_kernel_info(self.handle, i).decode() for i in range(L.fd_plan_num_kernels(self.handle))]


class Engine:
    """One per model instance (created lazily by MobileNetSkipAdd.forward)."""

    # extra fd_plan_create flags for every plan an Engine creates (A/B aid of tools/*.py, e.g. capi.FD_PLAN_NO_UNIT_FUSION); set in code --
    # neither the host side nor the C ABI reads the environment
    default_plan_flags = 0

    def __init__(self, model, keep_activations=False, dtype=torch.float32, plan_flags=None):
        self.model = model
        self.layers = layers_of(model)
        self.keep = keep_activations
        self.plan_flags = Engine.default_plan_flags if plan_flags is None else int(plan_flags)
        self.plans = {}
        self.set_dtype(dtype)

    def set_dtype(self, dtype):
        """Storage type of the intermediate activations and the packed pointwise weights: float32 (default), float16 or
        bfloat16.  Parameters, the network input and the network output stay float32; accumulation is always fp32."""
        if dtype not in _DTYPES:
            raise capi.FastDepthError("unsupported compute dtype %r" % (dtype,))
        self.dtype = dtype

    def invalidate(self):
        for p in self.plans.values():
            p.version = None

    def _version(self):
        """Cheap fingerprint of the parameters the packed weights were made from (torch version counters + addresses).  Edits made
        through `.data` bypass the counters: call model.repack() after those."""
        ts = self.__dict__.get("_param_tensors")
        if ts is None or len(ts) != 5 * len(self.layers) or any(a is not b for a, b in zip(ts[::5], (l.conv.weight for l in self.layers))):
            ts = self._param_tensors = [t for l in self.layers for t in (l.conv.weight, l.bn.weight, l.bn.bias, l.bn.running_mean, l.bn.running_var)]
        return hash(tuple((t._version, t.data_ptr()) for t in ts))

    def plan_for(self, x):
        b, c, h, w = x.shape
        key = (b, h, w, x.device.index, self.dtype)
        p = self.plans.get(key)
        if p is None:
            p = self.plans[key] = _Plan(self, b, h, w, x.device, self.keep)
        return p

    def _pack(self, plan, stream):
        L = lib()
        n = len(self.layers)
        params = (capi.LayerParams * n)()
        for q, l in zip(params, self.layers):
            for name, t in (("conv_weight", l.conv.weight), ("bn_weight", l.bn.weight), ("bn_bias", l.bn.bias),
                            ("bn_mean", l.bn.running_mean), ("bn_var", l.bn.running_var)):
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise capi.FastDepthError("parameter %s.%s must be a contiguous float32 tensor on the GPU" % (l.name, name))
                setattr(q, name, t.data_ptr())
        eps = self.layers[0].bn.eps
        if any(l.bn.eps != eps for l in self.layers):
            raise capi.FastDepthError("mixed BatchNorm eps values are not supported")
        capi.check(L, L.fd_plan_pack_weights(plan.handle, params, n, eps, stream), "fd_plan_pack_weights")

    def forward(self, x):
        if self.model.training:
            # train mode: batch-statistics BatchNorm + hand-written backward behind a torch.autograd.Function
            from .train import TrainCore, autograd_forward
            tdt = torch.bfloat16 if self.dtype == torch.bfloat16 else torch.float32   # set_compute_dtype(bfloat16) -> bf16 train plan
            if self.dtype == torch.float16:
                raise capi.FastDepthError("train mode supports float32 or bfloat16 storage (fp16 gradients would need loss scaling)")
            core = self.__dict__.get("_train_core")
            if core is not None:
                ps = tuple(p.data_ptr() for l in self.layers for p in (l.conv.weight, l.bn.weight, l.bn.bias))
                if core.signature() != (tdt, self.layers[0].conv.weight.device, ps):
                    core = None                   # parameters were moved / re-allocated (model.to(...), load with assign=True): stale pointers
            if core is None:
                core = self._train_core = TrainCore(self.model, tdt)
            if torch.is_grad_enabled():
                return autograd_forward(core, x)
            return core.forward(x)
        if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32:
            raise capi.FastDepthError("expected a float32 [B,3,H,W] tensor, got %s %s" % (tuple(x.shape), x.dtype))
        x = x.contiguous()
        L = lib()
        plan = self.plan_for(x)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            ver = self._version()
            if plan.version != ver:
                self._pack(plan, stream)
                plan.version = ver
            y = torch.empty((x.shape[0], 1, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
            capi.check(L, L.fd_forward(plan.handle, x.data_ptr(), y.data_ptr(), stream), "fd_forward")
        return y

    def forward_timed(self, x):
        """Measurement aid: one forward with HIP events around every layer's kernel (synchronises).
        Returns (y, [ms per layer])."""
        L = lib()
        x = x.contiguous()
        plan = self.plan_for(x)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        n = len(self.layers)
        ms = (ctypes.c_float * n)()
        with torch.cuda.device(x.device):
            ver = self._version()
            if plan.version != ver:
                self._pack(plan, stream)
                plan.version = ver
            y = torch.empty((x.shape[0], 1, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
            capi.check(L, L.fd_forward_timed(plan.handle, x.data_ptr(), y.data_ptr(), stream, ms, n), "fd_forward_timed")
        return y, list(ms)

    def layer_stats(self, x, traffic=False):
        """[(layer name, kernel symbol, kernel info, algorithmic bytes, algorithmic flops)] for x's plan; with traffic=True a sixth
        field: the bytes the layer's launch has to move (a fused launch keeps its intermediate tensors on chip)."""
        L = lib()
        plan = self.plan_for(x)
        out = []
        for i, l in enumerate(self.layers):
            b, f, nb = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            capi.check(L, L.fd_plan_layer_stats(plan.handle, i, ctypes.byref(b), ctypes.byref(f)), "fd_plan_layer_stats")
            row = (l.name, L.fd_plan_kernel_symbol(plan.handle, i).decode(), L.fd_plan_kernel_info(plan.handle, i).decode(), b.value, f.value)
            if traffic:
                capi.check(L, L.fd_plan_layer_traffic(plan.handle, i, ctypes.byref(nb)), "fd_plan_layer_traffic")
                row += (nb.value,)
            out.append(row)
        return out

    # ---- hipGraph replay, optionally with the batch split over several streams --------------------------------------------
    def forward_graph(self, x, streams=1):
        """Inference forward replayed from a captured HIP graph (opt-in: the returned tensor is a STATIC buffer that the next
        call overwrites).  With streams = S > 1 the batch is cut into S sub-batches, each with its own plan/workspace, captured
        on S forked streams: independent frames need no communication, and the MFMA-bound pointwise kernels of one sub-batch
        overlap the HBM-bound depthwise kernels (and the ramp/tail) of another.  The graph is keyed by (input address, shape, S):
        call it with the same input tensor object (e.g. a pinned staging buffer) to replay without a copy."""
        if self.model.training or not x.is_cuda or x.dtype != torch.float32:
            raise capi.FastDepthError("forward_graph is an inference path for float32 GPU tensors")
        x = x.contiguous()
        key = (x.data_ptr(), tuple(x.shape), streams)
        g = self.__dict__.setdefault("_graphs", {}).get(key)
        if g is None or g["version"] != self._version():
            g = self._capture(x, streams)
            self._graphs[key] = g
        g["graph"].replay()
        return g["y"]

    def _capture(self, x, streams):
        b = x.shape[0]
        if b % streams:
            raise capi.FastDepthError("batch %d is not divisible by %d streams" % (b, streams))
        L = lib()
        sub = b // streams
        y = torch.empty((b, 1, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
        plans = [_Plan(self, sub, x.shape[2], x.shape[3], x.device, False) for _ in range(streams)]
        side = [torch.cuda.Stream(device=x.device) for _ in range(streams - 1)]
        cur = torch.cuda.current_stream(x.device)
        with torch.cuda.device(x.device):
            for p in plans:                      # pack outside the capture (weights are static for the graph's lifetime)
                self._pack(p, cur.cuda_stream)
            torch.cuda.synchronize(x.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                cap = torch.cuda.current_stream(x.device)
                for i, p in enumerate(plans):
                    s = cap if i == 0 else side[i - 1]
                    if i:
                        s.wait_stream(cap)
                    xi, yi = x[i * sub:(i + 1) * sub], y[i * sub:(i + 1) * sub]
                    capi.check(L, L.fd_forward(p.handle, xi.data_ptr(), yi.data_ptr(), s.cuda_stream), "fd_forward")
                for s in side:
                    cap.wait_stream(s)
        return {"graph": graph, "y": y, "plans": plans, "x": x, "version": self._version(), "side": side}

    def layer_output(self, x, index):
        """Test hook: NCHW copy of fused layer `index`'s output from the last forward on x's plan."""
        L = lib()
        plan = self.plan_for(x)
        ptr = ctypes.c_void_p()
        dims = [ctypes.c_int32() for _ in range(4)]
        capi.check(L, L.fd_layer_output(plan.handle, index, ctypes.byref(ptr), *[ctypes.byref(d) for d in dims]), "fd_layer_output")
        n, h, w, c = [d.value for d in dims]
        off = ptr.value - plan.workspace.data_ptr()
        esz = 4 if plan.dtype == torch.float32 else 2
        flat = plan.workspace[off:off + n * h * w * c * esz].view(plan.dtype)
        return flat.view(n, h, w, c).permute(0, 3, 1, 2).float().contiguous()
