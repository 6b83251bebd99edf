"""Train step on the HIP engine: train-mode forward (batch-statistics BatchNorm), hand-written backward, fused mean-L1
loss, fused multi-tensor SGD and data-parallel gradient all-reduce (RCCL via torch.distributed, one process per GPU).

The train step is NOT in the reference tree (README.md:65 names the upstream it was cut from); SURVEY.md section 3(4)
defines it as: module in .train() -> torch.nn.L1Loss()(pred, target) -> backward -> mean of per-replica gradients ->
torch.optim.SGD(lr, momentum, weight_decay).  BatchNorm statistics stay per replica (nn.DataParallel / DDP default
semantics, the only multi-GPU idiom the reference ever used: imagenet/mobilenet.py:68).

Two ways in:
  * drop-in: `model.train(); loss = nn.L1Loss()(model(x), target); loss.backward(); optimizer.step()` -- the module's
    forward routes through `TrainFunction`, a torch.autograd.Function whose forward/backward are the C-ABI calls;
  * fused: `TrainEngine(model, lr, ...).step(x, target)` -- forward, fd_l1_loss, bucketed backward overlapped with the
    all-reduce of finished buckets on a side stream, fd_sgd_step; no torch kernels on the path.
"""
import ctypes

import torch

from . import capi
from .engine import lib
from .plan import layers_of


def _stream_ptr(device):
    """torch's current HIP stream on `device` as the C ABI wants it (None on the CPU: only the tests' emulator build runs there)."""
    return torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else None


class _device_guard:
    def __init__(self, device):
        self.ctx = torch.cuda.device(device) if device.type == "cuda" else None

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


class _TrainPlan:
    default_flags = 0           # extra fd_train_plan_create flags for every plan (A/B aid of tools/*.py, e.g. capi.FD_PLAN_NO_BWD_PAIRING); set in code

    def __init__(self, layers, batch, height, width, device, dtype=torch.float32, L=None):
        self.L = L = L or lib()
        n = len(layers)
        descs = (capi.LayerDesc * n)(*[l.desc for l in layers])
        handle = ctypes.c_void_p()
        capi.check(L, capi.create_plan(L, True, descs, n, batch, height, width, capi.DTYPE_OF[dtype], _TrainPlan.default_flags, ctypes.byref(handle)), "fd_train_plan_create")
        self.handle = handle
        nbytes = L.fd_train_plan_workspace_bytes(handle)
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        base = (self.workspace.data_ptr() + 255) // 256 * 256
        capi.check(L, L.fd_train_plan_bind_workspace(handle, base, nbytes), "fd_train_plan_bind_workspace")

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                self.L.fd_train_plan_destroy(self.handle)
            except Exception:
                pass
            self.handle = None


def _check_param(t, what, allow_cpu=False):
    if not ((t.is_cuda or allow_cpu) and t.dtype == torch.float32 and t.is_contiguous()):
        raise capi.FastDepthError("%s must be a contiguous float32 tensor on the GPU" % what)


class TrainCore:
    """Shared plumbing: parameter tables, flat gradient buffer (reverse layer order, so that finished buckets are
    contiguous slices), train plans per input shape."""

    def __init__(self, model, dtype=torch.float32, _library=None):
        # _library: TEST HOOK -- an already loaded C-ABI library to use instead of libfastdepth_hip.so.  The CPU test tier passes the
        # emulator build (tests/hipemu) so that the host logic around the kernels (flat gradient buffer, buckets, all-reduce, SGD
        # table) runs with world_size 2 on gloo; the product never sets it, and without it CPU tensors are rejected.
        self.L = _library or lib()
        self._emulated = _library is not None
        self.model = model
        if dtype not in (torch.float32, torch.bfloat16):
            raise capi.FastDepthError("train step storage type must be float32 or bfloat16 (fp16 gradients would need loss scaling)")
        self.dtype = dtype                        # storage of saved activations / activation gradients / GEMM operands
        self.layers = layers_of(model)
        self.n = len(self.layers)
        dev = self.layers[0].conv.weight.device
        self.device = dev
        # flat gradient buffer: layer n-1 first ... layer 0 last (the order in which backward completes them)
        self.param_list = []                      # (layer index, kind, parameter) in flat order
        for i in reversed(range(self.n)):
            l = self.layers[i]
            for kind, p in (("conv_weight", l.conv.weight), ("bn_weight", l.bn.weight), ("bn_bias", l.bn.bias)):
                _check_param(p, "%s.%s" % (l.name, kind), self._emulated)
                self.param_list.append((i, kind, p))
        # every tensor starts on a 16-byte boundary of the flat buffer (the fused SGD and the reductions then use 16-byte accesses); the
        # padding elements stay zero
        offs, off = [], 0
        for _, _, p in self.param_list:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.total = off
        self.flat_offsets = offs
        self.flat_grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.grad_views, self.layer_span = {}, {}
        for (i, kind, p), off in zip(self.param_list, offs):
            self.grad_views[(i, kind)] = self.flat_grad[off:off + p.numel()].view_as(p)
            lo, hi = self.layer_span.get(i, (off, off))
            self.layer_span[i] = (min(lo, off), max(hi, off + (p.numel() + 3) // 4 * 4))
        self.c_grads = (capi.LayerGrads * self.n)()
        for i in range(self.n):
            for kind in ("conv_weight", "bn_weight", "bn_bias"):
                setattr(self.c_grads[i], kind, self.grad_views[(i, kind)].data_ptr())
        self.plans = {}
        self.eps = self.layers[0].bn.eps
        self.bn_momentum = self.layers[0].bn.momentum
        if any(l.bn.eps != self.eps or l.bn.momentum != self.bn_momentum for l in self.layers):
            raise capi.FastDepthError("mixed BatchNorm eps/momentum values are not supported")
        if self.bn_momentum is None:
            raise capi.FastDepthError("BatchNorm momentum=None (cumulative moving average) is not supported by the train step")
        self.generation = 0                       # stamped on every train-mode forward (TrainFunction.backward checks it)
        self.params_in_order = [p for l in self.layers for p in (l.conv.weight, l.bn.weight, l.bn.bias)]
        self.views_in_param_order = [self.grad_views[(i, k)] for i in range(self.n) for k in ("conv_weight", "bn_weight", "bn_bias")]

    def signature(self):
        """What the cached pointers depend on: the engine rebuilds the core when a parameter was re-allocated or moved."""
        return (self.dtype, self.device, tuple(p.data_ptr() for p in self.params_in_order))

    def c_params(self):
        params = (capi.LayerParams * self.n)()
        for q, l in zip(params, self.layers):
            for name, t in (("conv_weight", l.conv.weight), ("bn_weight", l.bn.weight), ("bn_bias", l.bn.bias),
                            ("bn_mean", l.bn.running_mean), ("bn_var", l.bn.running_var)):
                _check_param(t, "%s.%s" % (l.name, name), self._emulated)
                setattr(q, name, t.data_ptr())
            nbt = l.bn.num_batches_tracked          # int64 scalar: incremented by the forward kernel's finalisation tail
            q.bn_num_batches_tracked = nbt.data_ptr() if nbt is not None else None
        return params

    def plan_for(self, x):
        b, c, h, w = x.shape
        key = (b, h, w, x.device.index, self.dtype)
        p = self.plans.get(key)
        if p is None:
            p = self.plans[key] = _TrainPlan(self.layers, b, h, w, x.device, self.dtype, self.L)
        return p

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32 or not (x.is_cuda or self._emulated):
            raise capi.FastDepthError("expected a float32 [B,3,H,W] GPU tensor, got %s %s on %s" % (tuple(x.shape), x.dtype, x.device))
        L = self.L
        x = x.contiguous()
        plan = self.plan_for(x)
        stream = _stream_ptr(x.device)
        y = torch.empty((x.shape[0], 1, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
        self._params = self.c_params()
        self._x = x                                # the stem's weight gradient re-reads the input in backward
        with _device_guard(x.device):
            capi.check(L, L.fd_train_forward(plan.handle, self._params, self.n, self.eps, self.bn_momentum, x.data_ptr(), y.data_ptr(), stream), "fd_train_forward")
        self._plan = plan
        self.generation += 1
        return y

    def backward_range(self, dy, from_layer, to_layer):
        L = self.L
        stream = _stream_ptr(dy.device)
        with _device_guard(dy.device):
            capi.check(L, L.fd_train_backward_range(self._plan.handle, self._params, self.c_grads, self.n, dy.data_ptr(), from_layer, to_layer, stream),
                       "fd_train_backward_range")

    def backward(self, dy):
        self.backward_range(dy.contiguous(), self.n - 1, 0)


class TrainFunction(torch.autograd.Function):
    """pred = TrainFunction.apply(core, x, *parameters): autograd entry point of the drop-in module in .train() mode.

    The saved activations live in the plan's workspace, i.e. they belong to the LAST train-mode forward of this shape: every forward
    is stamped with a generation number and backward refuses to run against a workspace a later forward has overwritten (two
    forwards summed into one loss are not part of this path).  The parameters go through save_for_backward so that autograd's
    version-counter check fires when one of them is edited in place between forward and backward.

    Gradients are RETURNED to autograd (so `torch.autograd.grad`, `backward(inputs=...)`, tensor hooks and accumulation hooks all see
    them, and a parameter with requires_grad=False gets none).  The kernels write one flat fp32 buffer owned by the plan; backward hands
    autograd views of ONE private copy of it (a single 15.8 MB device copy per backward, ~8 us on MI355X -- against a 2.5-4 ms step), so
    nothing a caller holds -- `.grad`, the result of `torch.autograd.grad`, a tensor seen by a hook -- aliases memory that the next backward
    of this model overwrites.  (Round 3 returned views of the plan's own buffer: copy-free, but values held across a second backward
    changed under the caller, and the copy-free `.grad` adoption leaned on AccumulateGrad's use-count heuristic.  The fused
    `TrainEngine.step`, which owns loss, all-reduce and SGD, still works in place on the flat buffer.)"""

    @staticmethod
    def forward(ctx, core, x, *params):
        if x.requires_grad:
            raise capi.FastDepthError("gradients with respect to the input image are not part of this path")
        ctx.core = core
        y = core.forward(x)
        ctx.gen = core.generation
        ctx.save_for_backward(*params)
        return y

    @staticmethod
    def backward(ctx, dy):
        core = ctx.core
        if core.generation != ctx.gen:
            raise capi.FastDepthError("backward of a train-mode forward whose saved activations were overwritten by a later forward "
                                      "(one outstanding forward per model: call backward before the next train-mode forward)")
        params = ctx.saved_tensors                # raises if a parameter was modified in place since forward
        need = ctx.needs_input_grad[2:]
        flat = core.flat_grad
        core.backward(dy.contiguous())
        mine = flat.clone()                       # the caller's gradients never alias the plan's buffer (see the class docstring)
        out = []
        for v, n in zip(core.views_in_param_order, need):
            if not n:
                out.append(None)
                continue
            start = (v.data_ptr() - flat.data_ptr()) // 4
            out.append(mine[start:start + v.numel()].view(v.shape))
        return (None, None) + tuple(out)


def autograd_forward(core, x):
    return TrainFunction.apply(core, x, *core.params_in_order)


def make_buckets(layer_bytes, n_buckets):
    """Splits layers n-1..0 (backward order) into <= n_buckets contiguous ranges of roughly equal bytes.
    Returns [(from_layer, to_layer)] with from >= to."""
    n = len(layer_bytes)
    total = float(sum(layer_bytes))
    buckets, start, acc = [], n - 1, 0.0
    for i in range(n - 1, -1, -1):
        acc += layer_bytes[i]
        remaining = n_buckets - len(buckets) - 1
        if (acc >= total / n_buckets and remaining > 0 and i > 0) or i == 0:
            buckets.append((start, i))
            start, acc = i - 1, 0.0
    return buckets


def make_buckets_by_finish(layer_bytes, fractions=(0.9,)):
    """Buckets chosen by WHEN their bytes are finished rather than by equal bytes.  Backward runs from the head down; in this network the
    decoder and the early encoder layers take most of the TIME (large maps) and hold almost none of the gradient BYTES -- 86 % of them belong to
    conv7..conv13 / decode_conv1 and are complete about half-way through backward.  One cut right after the layer at
    which the cumulative bytes (in backward order) reach `fractions[0]` of the total puts the bulk of the exchange under the second half of
    backward and leaves a small, latency-bound bucket for the end; every bucket costs a reduction launch, a stream hop and a collective
    launch, so fewer is better (measured: 4 equal-byte buckets cost +0.30 ms per step before a byte crosses xGMI).
    Returns [(from_layer, to_layer)] with from >= to."""
    n = len(layer_bytes)
    total = float(sum(layer_bytes))
    cuts, acc, k = [], 0.0, 0
    for i in range(n - 1, 0, -1):                          # a cut after layer i means the next bucket starts at i-1 (so i >= 1)
        acc += layer_bytes[i]
        if k < len(fractions) and acc >= fractions[k] * total:
            cuts.append(i)
            k += 1
    buckets, start = [], n - 1
    for c in cuts:
        buckets.append((start, c))
        start = c - 1
    buckets.append((start, 0))
    return buckets


class TrainEngine(TrainCore):
    """Fused train step with SGD(momentum, weight decay) and optional data parallelism.

    process_group: a torch.distributed process group (backend "nccl" == RCCL on ROCm) or None.  With a group of size n the
    step is: local forward/backward on this rank's sub-batch; the flat gradient buffer is all-reduced (sum) bucket by bucket
    on a side stream as soon as backward has finished a bucket; fd_sgd_step applies grad_scale = 1/n (gradient mean)."""

    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=1e-4, process_group=None, n_buckets=None, force_buckets=False,
                 dtype=torch.float32, masked_loss=False, grad_exchange_dtype=torch.float32, exchange="auto", _library=None, _elide_collectives=False):
        """n_buckets: None (default) = two buckets cut by finish time (make_buckets_by_finish); an integer = that many buckets of roughly
        equal bytes (make_buckets).  grad_exchange_dtype: torch.float32 (default: the 15.84 MB fp32 vector is all-reduced in place) or
        torch.bfloat16 (every bucket is converted to bfloat16, all-reduced as 7.92 MB, converted back: half the bytes over xGMI for one
        rounding of every summand and of the sum).  exchange: "library" = the all-reduces are issued by libfastdepth_hip.so itself
        (fd_train_backward_allreduce: RCCL on the library's own communicator and stream, torch.distributed only broadcasts the 128-byte
        rendezvous id once); "torch" = every bucket goes through torch.distributed.all_reduce on a side stream (rounds 1-3; the only route for
        a gloo group, i.e. the CPU test tier); "auto" = "library" for an nccl group on a GPU, "torch" otherwise."""
        super().__init__(model, dtype, _library)
        if grad_exchange_dtype not in (torch.float32, torch.bfloat16):
            raise capi.FastDepthError("grad_exchange_dtype must be float32 or bfloat16")
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        # masked_loss: mean-L1 over the pixels with target > 0 only (the upstream train script's MaskedL1Loss, README.md:65) instead of
        # torch.nn.L1Loss over every pixel
        self.masked_loss = bool(masked_loss)
        self.group = process_group
        self.world = 1
        if process_group is not None:
            import torch.distributed as dist
            self.dist = dist
            self.world = dist.get_world_size(process_group)
        self.flat_mom = torch.zeros_like(self.flat_grad)
        self.steps = 0
        # SGD table on the device: (param ptr, grad ptr, momentum ptr, numel) per tensor
        rec = []
        for (i, kind, p), off in zip(self.param_list, self.flat_offsets):
            rec += [p.data_ptr(), self.flat_grad.data_ptr() + 4 * off, self.flat_mom.data_ptr() + 4 * off, p.numel()]
        self.sgd_table = torch.tensor(rec, dtype=torch.int64).to(self.device)
        self.layer_bytes = [4 * (self.layer_span[i][1] - self.layer_span[i][0]) for i in range(self.n)]
        self.use_comm = process_group is not None and (self.world > 1 or force_buckets)     # force_buckets: exercise the path on 1 rank
        if not self.use_comm:
            self.buckets = [(self.n - 1, 0)]
        elif n_buckets is None:
            self.buckets = make_buckets_by_finish(self.layer_bytes)
        else:
            self.buckets = make_buckets(self.layer_bytes, n_buckets)
        self.exchange_dtype = grad_exchange_dtype
        self.flat_grad16 = torch.zeros(self.total, dtype=torch.bfloat16, device=self.device) if (self.use_comm and grad_exchange_dtype == torch.bfloat16) else None
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.use_comm and self.device.type == "cuda") else None
        # ---- the library's own communicator (RCCL bound at run time inside libfastdepth_hip.so) ----
        self.comm = None
        if exchange not in ("auto", "library", "torch"):
            raise capi.FastDepthError("exchange must be 'auto', 'library' or 'torch'")
        want_lib = exchange == "library" or (exchange == "auto" and self.use_comm and self.device.type == "cuda" and self.dist.get_backend(process_group) == "nccl")
        if want_lib and self.use_comm:
            if self.device.type != "cuda" and _library is None:
                raise capi.FastDepthError("exchange='library' needs a GPU (RCCL)")          # (a test-only library -- the CPU emulator -- brings its own one-rank communicator)
            L = self.L
            rank = self.dist.get_rank(process_group)
            # Rendezvous without a one-sided failure: nothing raises between the collectives.  Rank 0 broadcasts the 128-byte id TOGETHER with a status
            # byte (fd_comm_unique_id fails where librccl.so.1 cannot be bound), every rank then reports whether ITS fd_comm_create succeeded, and the
            # group decides jointly (all-reduce MIN of the flag): all ranks take the library route, or all of them fall back to the torch.distributed
            # route ("auto") / raise ("library").  A rank that raised alone would leave its peers waiting in the next collective for ever.
            msg = torch.zeros(129, dtype=torch.uint8)
            err0 = ""
            if rank == 0:
                if L.fd_comm_unique_id(msg.data_ptr()) == 0:
                    msg[128] = 1
                else:
                    err0 = L.fd_last_error().decode()
            msg_dev = msg.to(self.device)
            self.dist.broadcast(msg_dev, src=self.dist.get_global_rank(process_group, 0) if hasattr(self.dist, "get_global_rank") else 0, group=process_group)
            msg = msg_dev.cpu()
            handle, err = ctypes.c_void_p(), err0
            ok = bool(msg[128].item())
            if ok:
                with _device_guard(self.device):
                    ok = L.fd_comm_create(msg[:128].contiguous().data_ptr(), rank, self.world, ctypes.byref(handle)) == 0
                if not ok:
                    err = L.fd_last_error().decode()
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=process_group)
            if int(flag.item()) == 0:                      # agreed by every rank
                if ok and handle:
                    L.fd_comm_destroy(handle)              # (this rank's communicator came up, a peer's did not: nobody keeps one)
                handle = None
                if exchange == "library":
                    raise capi.FastDepthError("exchange='library': the library's RCCL communicator could not be set up on every rank (%s)" % (err or "a peer rank failed"))
                import warnings
                warnings.warn("fastdepth_hip: library-issued gradient exchange unavailable (%s); every rank uses torch.distributed.all_reduce per bucket" % (err or "a peer rank failed"))
            self.comm = handle
            if _elide_collectives and self.world == 1 and self.comm is not None:
                L.fd_comm_elide_collectives.argtypes = [ctypes.c_void_p, ctypes.c_int32]      # measurement hook (csrc/fd_tuning.h), one rank only
                L.fd_comm_elide_collectives.restype = None
                L.fd_comm_elide_collectives(handle, 1)
            nb = len(self.buckets) if self.comm is not None else 0
            self.c_buckets = (capi.GradBucket * max(nb, 1))()
            for k, (fl, tl) in enumerate(self.buckets if self.comm is not None else ()):
                g32 = self.bucket_slice(fl, tl)
                g16 = self.bucket_slice(fl, tl, self.flat_grad16) if self.flat_grad16 is not None else None
                self.c_buckets[k] = capi.GradBucket(fl, tl, g32.data_ptr(), g32.numel(), g16.data_ptr() if g16 is not None else None)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._dpred = None
        self._scratch = torch.empty(self.L.fd_l1_loss_scratch_bytes(1), dtype=torch.uint8, device=self.device)
        self.last_comm_us = None                  # set by step(time_comm=True): device time of the all-reduces / of the whole step

    def close(self):
        """Releases the library's RCCL communicator (a collective call: every rank of the group closes).  Call it before the process group is
        destroyed / the interpreter exits; __del__ only does it as a best effort."""
        if getattr(self, "comm", None):
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self.L.fd_comm_destroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bucket_slice(self, from_layer, to_layer, buf=None):
        return (self.flat_grad if buf is None else buf)[self.layer_span[from_layer][0]:self.layer_span[to_layer][1]]

    def _exchange(self, from_layer, to_layer, stream_ptr):
        """The summing all-reduce of one finished bucket, issued on the CURRENT stream (the caller switched to the communication stream);
        returns the work handle.  bfloat16 exchange: convert, all-reduce the 16-bit copy, convert back -- all in stream order."""
        g32 = self.bucket_slice(from_layer, to_layer)
        if self.flat_grad16 is None:
            return self.dist.all_reduce(g32, group=self.group, async_op=True)
        g16 = self.bucket_slice(from_layer, to_layer, self.flat_grad16)
        L = self.L
        capi.check(L, L.fd_cast_gradients(g32.data_ptr(), g16.data_ptr(), g32.numel(), 1, stream_ptr), "fd_cast_gradients")
        w = self.dist.all_reduce(g16, group=self.group, async_op=True)
        w.wait()                                   # stream-ordered on the GPU (no host sync); on the CPU (tests) it blocks
        capi.check(L, L.fd_cast_gradients(g16.data_ptr(), g32.data_ptr(), g32.numel(), 0, stream_ptr), "fd_cast_gradients")
        return w

    def step(self, x, target, time_comm=False):
        """One train step on this rank's (x, target); returns the local mean-L1 loss as a 1-element GPU tensor.
        time_comm=True (measurement aid, synchronises): records HIP events around the step and around the side-stream all-reduces and
        leaves (all-reduce us, step us, exposed us = step end - backward end) in self.last_comm_us."""
        L = self.L
        pred = self.forward(x)
        target = target.contiguous()
        if self._dpred is None or self._dpred.shape != pred.shape:
            self._dpred = torch.empty_like(pred)
        on_gpu = self.device.type == "cuda"
        cur = torch.cuda.current_stream(self.device) if on_gpu else None
        sp = cur.cuda_stream if on_gpu else None
        ev = None
        if time_comm and on_gpu:
            ev = {k: torch.cuda.Event(enable_timing=True) for k in ("bwd0", "bwd1", "c0", "c1", "end")}
        with _device_guard(self.device):
            loss_fn = L.fd_l1_loss_masked if self.masked_loss else L.fd_l1_loss
            capi.check(L, loss_fn(pred.data_ptr(), target.data_ptr(), self._dpred.data_ptr(), self.loss.data_ptr(), pred.numel(),
                               self._scratch.data_ptr(), sp), "fd_l1_loss")
            if ev:
                ev["bwd0"].record(cur)
            works = []
            if self.comm is not None:
                # ONE library call: every bucket's backward range, its event hand-over to the communicator's stream and its RCCL all-reduce;
                # on return `cur` already waits for the last collective
                capi.check(L, L.fd_train_backward_allreduce(self._plan.handle, self._params, self.c_grads, self.n, self._dpred.data_ptr(), self.comm,
                                                            self.c_buckets, len(self.buckets), sp), "fd_train_backward_allreduce")
            for bi, (from_layer, to_layer) in enumerate(self.buckets if self.comm is None else ()):
                self.backward_range(self._dpred, from_layer, to_layer)
                if self.use_comm:
                    if on_gpu:
                        done = torch.cuda.Event()
                        done.record(cur)
                        self.comm_stream.wait_event(done)
                        with torch.cuda.stream(self.comm_stream):
                            if ev and bi == 0:
                                ev["c0"].record(self.comm_stream)
                            works.append(self._exchange(from_layer, to_layer, self.comm_stream.cuda_stream))
                    else:
                        works.append(self._exchange(from_layer, to_layer, None))
            if ev:
                ev["bwd1"].record(cur)
            if self.use_comm and self.comm is None:
                for w in works:
                    w.wait()                       # makes the current stream wait for the collective (no host sync on the GPU)
                if on_gpu:
                    if ev:
                        with torch.cuda.stream(self.comm_stream):
                            ev["c1"].record(self.comm_stream)
                    cur.wait_stream(self.comm_stream)
            capi.check(L, L.fd_sgd_step(self.sgd_table.data_ptr(), len(self.param_list), self.total, self.lr, self.momentum, self.weight_decay,
                                        1.0 / self.world, int(self.steps == 0), sp), "fd_sgd_step")
            if ev:
                ev["end"].record(cur)
                torch.cuda.synchronize(self.device)
                if self.comm is not None:
                    a, b = ctypes.c_float(), ctypes.c_float()
                    capi.check(L, L.fd_comm_last_exchange_ms(self.comm, ctypes.byref(a), ctypes.byref(b)), "fd_comm_last_exchange_ms")
                    self.last_comm_us = (a.value * 1e3, ev["bwd0"].elapsed_time(ev["end"]) * 1e3, b.value * 1e3)
                else:
                    comm = ev["c0"].elapsed_time(ev["c1"]) * 1e3 if self.use_comm else 0.0
                    self.last_comm_us = (comm, ev["bwd0"].elapsed_time(ev["end"]) * 1e3, ev["bwd1"].elapsed_time(ev["end"]) * 1e3)
        self.steps += 1
        return self.loss
