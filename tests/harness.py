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

    def __init__(self, kind, model, x, keep=True, dtype=torch.float32, flags=0):
        self.lib = L = get_lib(kind)
        self.kind, self.model, self.dev = kind, model, x.device
        self.dtype = dtype
        fd_dtype = {torch.float32: capi.FD_F32, torch.float16: capi.FD_F16, torch.bfloat16: capi.FD_BF16}[dtype]
        self.layers = layers_of(model)
        n = len(self.layers)
        descs = (capi.LayerDesc * n)(*[l.desc for l in self.layers])
        self.h = ctypes.c_void_p()
        b, _, hh, ww = x.shape
        capi.check(L, capi.create_plan(L, False, descs, n, b, hh, ww, fd_dtype, (capi.FD_PLAN_KEEP_ACTIVATIONS if keep else 0) | flags,
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
        esz = 4 if self.dtype == torch.float32 else 2
        return self.ws[off:off + n * h * w * c * esz].view(self.dtype).view(n, h, w, c).permute(0, 3, 1, 2).float().contiguous().cpu()

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


def saturate_encoder(model, gain=4.0):
    """Encoder BatchNorm gamma x gain (the recipe of oracle/make_golden.py's `sat6` variant): with batch statistics the
    post-BN values are ~N(beta, gain^2), so for gain 4 about 7 % of every ReLU6 unit's outputs sit on the clamp at 6 and the `y < 6`
    side of the backward mask (reference imagenet/mobilenet.py:16-20) is really exercised."""
    for i in range(14):
        for mod in getattr(model, "conv%d" % i):
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.data.mul_(gain)
    return model


LAST_LOCAL_INFO = None     # kernel-selection counts of the last local_train_parity call (private test hooks of csrc/fd_tuning.h)
LAST_SAT6_FRAC = None      # fraction of ReLU6-unit pre-activations >= 6 seen by the last train_parity_report / local_train_parity call


def compare_with_oracle(kind, model, x, device, dtype=torch.float32, flags=0):
    """Runs the C ABI path and the C oracle on the same weights/input; returns (rel err of the output,
    [rel err per fused layer], plan info)."""
    from oracle import oracle
    model = model.eval()
    y_ref, taps_ref = oracle.forward(model.state_dict(), x.numpy(), taps=True)
    plan = CPlan(kind, model, x.to(device), dtype=dtype, flags=flags)
    y = plan.forward(x.to(device)).cpu().numpy()
    y2 = plan.forward(x.to(device)).cpu().numpy()          # a second pass over the same plan (stream-K counters must have returned to 0)
    assert np.array_equal(y, y2), "forward is not reproducible run to run"
    errs = [rel_err(plan.tap(i).numpy(), taps_ref[i]) for i in range(len(taps_ref) - 1)]
    errs.append(rel_err(y, y_ref))
    info = plan.info()
    plan.close()
    return rel_err(y, y_ref), errs, info


class CTrainPlan:
    """fd_train_plan + workspace; parameters are private fp32 copies on the plan's device (running stats get updated in place)."""

    def __init__(self, kind, model, x, keep=False, dtype=torch.float32, flags=0):
        self.lib = L = get_lib(kind)
        self.dtype = dtype
        self.dev = x.device
        self.layers = layers_of(model)
        n = self.n = len(self.layers)
        descs = (capi.LayerDesc * n)(*[l.desc for l in self.layers])
        self.h = ctypes.c_void_p()
        b, _, hh, ww = x.shape
        capi.check(L, capi.create_plan(L, True, descs, n, b, hh, ww, capi.DTYPE_OF[dtype], (capi.FD_PLAN_KEEP_ACTIVATIONS if keep else 0) | flags, ctypes.byref(self.h)), "fd_train_plan_create")
        nbytes = L.fd_train_plan_workspace_bytes(self.h)
        self.ws = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.dev)
        base = (self.ws.data_ptr() + 255) // 256 * 256
        capi.check(L, L.fd_train_plan_bind_workspace(self.h, base, nbytes), "fd_train_plan_bind_workspace")
        self.stream = torch.cuda.current_stream().cuda_stream if x.is_cuda else None
        self.params = (capi.LayerParams * n)()
        self.tensors = []          # per layer: dict name -> tensor
        for q, l in zip(self.params, self.layers):
            d = {}
            for name, t in (("conv_weight", l.conv.weight), ("bn_weight", l.bn.weight), ("bn_bias", l.bn.bias),
                            ("bn_mean", l.bn.running_mean), ("bn_var", l.bn.running_var)):
                d[name] = t.detach().to(self.dev, torch.float32).contiguous().clone()
                setattr(q, name, d[name].data_ptr())
            self.tensors.append(d)
        self.eps, self.momentum = self.layers[0].bn.eps, self.layers[0].bn.momentum

    def forward(self, x):
        x = x.contiguous()
        self._x = x                 # fd_train_backward re-reads the network input (stem weight gradient): keep it alive
        y = torch.full((x.shape[0], 1, x.shape[2], x.shape[3]), float("nan"), dtype=torch.float32, device=self.dev)
        capi.check(self.lib, self.lib.fd_train_forward(self.h, self.params, self.n, self.eps, self.momentum, x.data_ptr(), y.data_ptr(), self.stream), "fd_train_forward")
        if x.is_cuda:
            torch.cuda.synchronize()
        return y

    def backward(self, dy):
        self.grads = (capi.LayerGrads * self.n)()
        self.grad_tensors = []
        for g, d in zip(self.grads, self.tensors):
            gd = {k: torch.full_like(d[k], float("nan")) for k in ("conv_weight", "bn_weight", "bn_bias")}
            for k, t in gd.items():
                setattr(g, k, t.data_ptr())
            self.grad_tensors.append(gd)
        dy = dy.to(self.dev).contiguous()
        capi.check(self.lib, self.lib.fd_train_backward(self.h, self.params, self.grads, self.n, dy.data_ptr(), self.stream), "fd_train_backward")
        if dy.is_cuda:
            torch.cuda.synchronize()
        return self.grad_tensors

    def tensor(self, i, which=0):
        ptr = ctypes.c_void_p()
        d = [ctypes.c_int32() for _ in range(4)]
        capi.check(self.lib, self.lib.fd_train_layer_tensor(self.h, i, which, ctypes.byref(ptr), *[ctypes.byref(v) for v in d]), "fd_train_layer_tensor")
        n, h, w, c = [v.value for v in d]
        off = ptr.value - self.ws.data_ptr()
        dt = torch.float32 if (which == 2 or i == self.n - 1) else self.dtype      # tables and the 1-channel head are fp32 in every plan
        return self.ws[off:off + n * h * w * c * dt.itemsize].view(dt).view(n, h, w, c).permute(0, 3, 1, 2).contiguous().cpu().float()

    def close(self):
        if self.h:
            self.lib.fd_train_plan_destroy(self.h)
            self.h = None


def train_parity_report(kind, model, x, target, device, dtype=torch.float32, kink=1e-4):
    """Train-mode forward + backward through the C ABI vs the torch-functional oracle in fp64.

    ReLU/ReLU6 are discontinuous in their derivative: a pre-activation of magnitude ~1e-7 lands on either side of 0
    (or 6) depending on fp32 rounding order, and through BatchNorm's per-channel sums one flipped element perturbs every
    gradient upstream.  The check is therefore two-part and rigorous:
      (1) forward: prediction, running statistics and every unit's pre-activation y_i agree with the fp64 oracle; the
          pass-through masks may differ only where |y_i| is within 1e-4 (relative to the layer's scale) of the kink;
      (2) backward: the fp64 oracle is re-run with OUR masks and the SAME dLoss/dpred, which makes the map smooth; all
          114 gradient tensors must then agree to `tol` (relative L2, with an absolute floor tied to the global norm).
    """
    from oracle import oracle, torch_ref
    model = model.train()
    names = oracle.unit_names()
    p64 = torch_ref.params_from_state(model.state_dict(), torch.float64)
    bn64 = []
    with torch.no_grad():
        pred64 = torch_ref.forward(p64, x.double(), train=True, bn_taps=bn64)
    tp = CTrainPlan(kind, model, x.to(device), keep=True, dtype=dtype)
    y = tp.forward(x.to(device)).cpu()
    rep = {"pred_err": rel_err(y.numpy(), pred64.numpy()), "tensors": {}, "running": 0.0, "y_err": 0.0, "mask_flips": 0, "bad_flips": 0}
    masks = []
    n_sat = n_relu6 = 0
    for i, (cp, bp, _, _, act) in enumerate(names):
        z = tp.tensor(i, 0).double()
        st = tp.tensor(i, 2).double()[0, :, :, 0].t()          # memory is [4][C]; tensor() hands it back as (1, C, 4, 1)
        yo = z * st[0].view(1, -1, 1, 1) + st[1].view(1, -1, 1, 1)
        yr = bn64[i]
        if i == 37 and yr.shape[-1] == 2 * yo.shape[-1]:
            yr = yr[:, :, ::2, ::2]
        scale = float(yr.abs().max())
        rep["y_err"] = max(rep["y_err"], float((yo - yr).abs().max()) / scale)
        lo, hi = yo > 0, (yo < 6) if act == oracle.ACT_RELU6 else torch.ones_like(yo, dtype=torch.bool)
        if act == oracle.ACT_RELU6:
            n_sat += int((yo >= 6).sum()); n_relu6 += yo.numel()
        lo_r, hi_r = yr > 0, (yr < 6) if act == oracle.ACT_RELU6 else torch.ones_like(yr, dtype=torch.bool)
        flips = (lo != lo_r) | (hi != hi_r)
        rep["mask_flips"] += int(flips.sum())
        dist = torch.minimum(yr.abs(), (yr - 6).abs()) if act == oracle.ACT_RELU6 else yr.abs()
        rep["bad_flips"] += int((flips & (dist > kink * scale)).sum())
        if i == 37 and bn64[i].shape[-1] == 2 * yo.shape[-1]:
            lo, hi = lo.repeat_interleave(2, 2).repeat_interleave(2, 3), hi.repeat_interleave(2, 2).repeat_interleave(2, 3)
        masks.append((lo, hi))
    global LAST_SAT6_FRAC
    LAST_SAT6_FRAC = n_sat / max(n_relu6, 1)
    # backward with identical masks and identical dLoss/dpred
    p64g = torch_ref.params_from_state(model.state_dict(), torch.float64, requires_grad=True)
    predm = torch_ref.forward(p64g, x.double(), train=True, masks=masks)
    dpred = torch.sign(predm.detach() - target.double()) / predm.numel()
    predm.backward(dpred)
    g64 = {k: v.grad.detach() for k, v in p64g.items() if v.requires_grad}
    grads = tp.backward(dpred.float())
    for i, (cp, bp, _, _, _) in enumerate(names):
        for ours, key in ((grads[i]["conv_weight"], cp + ".weight"), (grads[i]["bn_weight"], bp + ".weight"), (grads[i]["bn_bias"], bp + ".bias")):
            ref = g64[key]
            rep["tensors"][key] = (float((ours.cpu().double() - ref).norm()), float(ref.norm()))
        for ours, key in ((tp.tensors[i]["bn_mean"], bp + ".running_mean"), (tp.tensors[i]["bn_var"], bp + ".running_var")):
            rep["running"] = max(rep["running"], rel_err(ours.cpu().numpy(), p64[key].numpy()))
    rep["global_norm"] = float(torch.sqrt(sum((v ** 2).sum() for v in g64.values())))
    tp.close()
    return rep


def assert_train_parity(rep, tol=1e-3):
    assert rep["pred_err"] < tol and rep["y_err"] < tol and rep["running"] < tol, (rep["pred_err"], rep["y_err"], rep["running"])
    assert rep["bad_flips"] == 0, "activation masks differ away from the kink: %d of %d flips" % (rep["bad_flips"], rep["mask_flips"])
    floor = 1e-5 * rep["global_norm"]
    bad = {k: v for k, v in rep["tensors"].items() if not v[0] <= max(tol * v[1], floor)}
    assert not bad, "gradient tensors out of tolerance (abs err, norm): %s" % bad


def local_train_parity(kind, model, x, target, device, dtype=torch.float32, flags=0):
    """Layer-LOCAL parity of one train step (forward + backward) through the C ABI, valid for fp32 and bf16 plans.

    The end-to-end comparison of train_parity_report is meaningless once activations are stored in bfloat16 on a tiny network
    (batch-statistics BatchNorm over a handful of samples amplifies the 2^-9 storage noise chaotically).
    Here every unit is checked on ITS OWN stored inputs instead: with the tensors the plan kept (z_i, BatchNorm tables, G_i,
    skip gradients, dz_i) the fp64 reference recomputes, per unit, the conv output, the batch statistics / running statistics,
    dgamma / dbeta, dz, the weight gradient, the gradient handed to the producer and to the skip source, using torch autograd
    on that single unit.  16-bit operands are rounded exactly where the kernels round them (GEMM operands, stored tensors), so
    the only admissible differences are one storage rounding of the result and fp32-vs-fp64 accumulation order.
    Returns {category: (worst error, layer name)}; errors are max-abs relative to the reference tensor's max-abs."""
    import torch.nn.functional as F
    from fastdepth_hip.capi import FD_OP_DW, FD_OP_PW, FD_OP_STEM, FD_ACT_RELU6
    model = model.train()
    h16 = dtype != torch.float32
    rnd = (lambda t: t.float().to(dtype).double()) if h16 else (lambda t: t)
    run0 = [(l.bn.running_mean.detach().double().clone(), l.bn.running_var.detach().double().clone()) for l in layers_of(model)]
    tp = CTrainPlan(kind, model, x.to(device), keep=True, dtype=dtype, flags=flags)
    y = tp.forward(x.to(device)).cpu()
    dpred = torch.sign(y - target) / y.numel()
    grads = tp.backward(dpred)
    L, n = tp.layers, tp.n
    Z = [tp.tensor(i, 0).double() for i in range(n)]
    ST = [tp.tensor(i, 2).double()[0, :, :, 0].t() for i in range(n)]
    G = [tp.tensor(i, 1).double() for i in range(n)]
    # which depthwise kernels kept their LDS patches in the 16-bit storage type (private test hook, csrc/fd_tuning.h): bit 0 forward, bit 1 backward,
    # bit 2 / bit 3: the forward / the backward-data kernel also rounds its taps (fd_dw5_rows_train / fd_dw5_bwd_rows)
    tp.lib.fd_train_plan_lds_rounding.restype = ctypes.c_int
    tp.lib.fd_train_plan_lds_rounding.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lds_round = [max(tp.lib.fd_train_plan_lds_rounding(tp.h, i), 0) for i in range(n)]
    tp.lib.fd_train_plan_unit_kernels.restype = ctypes.c_int
    tp.lib.fd_train_plan_unit_kernels.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    forms = [max(tp.lib.fd_train_plan_unit_kernels(tp.h, i), 0) for i in range(n)]
    rep_info = {"dw_units_with_16bit_lds_patches": sum(1 for v in lds_round if (v & 3) and not (v & (4 | 8))), "dw_units_on_dw5_bwd_rows": sum(1 for v in lds_round if v & 8), "dw_units_on_dw5_rows_train": sum(1 for v in lds_round if v & 4), "pw_units_on_gemm16": sum(1 for v in forms if v & 1),
                "dw_units_backward_on_row_kernels": sum(1 for v in forms if v & 8), "dw_units_on_dw3_rows_fwd": sum(1 for v in forms if v & 32), "units_finalised_by_consumer": sum(1 for v in forms if v & 2), "units_finalising_their_own_backward": sum(1 for v in forms if v & 4)}
    consumers = {}
    for i in range(n):
        if L[i].desc.src >= 0:
            consumers[L[i].desc.src] = i
    skip_sources = {L[i].desc.skip for i in range(n) if L[i].desc.skip >= 0}
    rep = {}

    def note(cat, err, i):
        if cat not in rep or err > rep[cat][0]:
            rep[cat] = (float(err), L[i].name)

    def relmax(a, b, mask=None):
        d = (a - b).abs()
        if mask is not None:
            d = d * mask
        return float(d.max()) / max(float(b.abs().max()), 1e-300)

    def pre(i):
        return Z[i] * ST[i][0].view(1, -1, 1, 1) + ST[i][1].view(1, -1, 1, 1)

    def act(i, yv):
        return yv.clamp(0, 6) if L[i].desc.act == FD_ACT_RELU6 else yv.clamp(min=0)

    def passmask(i, yv):
        m = yv > 0
        return (m & (yv < 6)) if L[i].desc.act == FD_ACT_RELU6 else m

    def near_kink(i, yv):
        scale = float(yv.abs().max())
        d = torch.minimum(yv.abs(), (yv - 6).abs()) if L[i].desc.act == FD_ACT_RELU6 else yv.abs()
        return d < 1e-5 * scale

    global LAST_SAT6_FRAC
    n_sat = sum(int((pre(i) >= 6).sum()) for i in range(n) if L[i].desc.act == FD_ACT_RELU6)
    LAST_SAT6_FRAC = n_sat / max(sum(Z[i].numel() for i in range(n) if L[i].desc.act == FD_ACT_RELU6), 1)
    for i in range(n):
        d = L[i].desc
        head = d.op == FD_OP_PW and d.cout == 1
        pw16 = h16 and d.op == FD_OP_PW and not head
        w = tp.tensors[i]["conv_weight"].cpu().double()
        a = ask = None
        if d.src < 0:
            inp = x.double()
        else:
            a = act(d.src, pre(d.src))
            a = (rnd(a) if pw16 else a).requires_grad_(True)
            inp = F.interpolate(a, scale_factor=2, mode="nearest") if (d.upsample and not head) else a
            if d.skip >= 0:
                ask = act(d.skip, pre(d.skip)).requires_grad_(True)
                inp = torch.cat((inp, ask), 1) if d.concat else inp + ask
        w = (rnd(w) if pw16 else w).requires_grad_(True)
        groups = d.cout if d.op == FD_OP_DW else 1
        # 16-bit LDS patches (fd_lane<T, 8> depthwise kernels): the conv input is rounded to the storage type on its way into LDS -- straight-through
        # for autograd, so that the gradient below flows to `a` / `ask` as the kernels compute it
        st_round = lambda t: t + (rnd(t.detach()) - t.detach())
        zr = F.conv2d(st_round(inp) if (lds_round[i] & 1) else inp, st_round(w) if (lds_round[i] & 4) else w, None, d.stride, d.ksize // 2, 1, groups)
        note("z", relmax(Z[i], zr.detach()), i)
        if (lds_round[i] & 1) != ((lds_round[i] >> 1) & 1) or (lds_round[i] & (4 | 8)):     # the backward kernels stage the input their own way
            # (bit 3: the backward-DATA kernel rounds its taps to the storage type -- straight-through, so that w.grad below is the correlation of dz
            # with the input as the weight-gradient kernel computes it, and a.grad flows through the rounded taps)
            zr = F.conv2d(st_round(inp) if (lds_round[i] & 2) else inp, st_round(w) if (lds_round[i] & 8) else w, None, d.stride, d.ksize // 2, 1, groups)
        # batch statistics of the STORED z, tables, running statistics
        C = d.cout
        cnt = Z[i].numel() // C
        mean = Z[i].mean((0, 2, 3))
        var = Z[i].var((0, 2, 3), unbiased=False)
        invstd = 1.0 / torch.sqrt(var + tp.eps)
        gamma, beta = tp.tensors[i]["bn_weight"].cpu().double(), tp.tensors[i]["bn_bias"].cpu().double()
        want = torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd])
        for r, nm in enumerate(("scale", "shift", "mean", "invstd")):
            note("bn_table", relmax(ST[i][r], want[r]), i)
        n_u = 4 * cnt if (head and d.upsample) else cnt
        note("running", relmax(tp.tensors[i]["bn_mean"].cpu().double(), (1 - tp.momentum) * run0[i][0] + tp.momentum * mean), i)
        note("running", relmax(tp.tensors[i]["bn_var"].cpu().double(), (1 - tp.momentum) * run0[i][1] + tp.momentum * var * n_u / (n_u - 1)), i)
        # backward of this unit
        xhat = (Z[i] - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        dbeta, dgamma = G[i].sum((0, 2, 3)), (G[i] * xhat).sum((0, 2, 3))
        gnorm = max(float(dbeta.abs().max()), float(dgamma.abs().max()), 1e-300)
        note("bn_grads", max(float((grads[i]["bn_bias"].cpu().double() - dbeta).abs().max()), float((grads[i]["bn_weight"].cpu().double() - dgamma).abs().max())) / gnorm, i)
        dz = want[0].view(1, -1, 1, 1) * (G[i] - (dbeta / cnt).view(1, -1, 1, 1) - xhat * (dgamma / cnt).view(1, -1, 1, 1))
        if pw16:
            dz_st = tp.tensor(i, 4).double()
            note("dz", relmax(dz_st, dz), i)
            dz = dz_st
        if head:
            yh = pre(i)
            up = (lambda t: t.repeat_interleave(2, 2).repeat_interleave(2, 3)) if d.upsample else (lambda t: t)
            note("pred", relmax(y.double(), up(act(i, yh))), i)
            gsum = F.avg_pool2d(dpred.double(), 2) * 4 if d.upsample else dpred.double()
            note("g_head", relmax(G[i], gsum * passmask(i, yh), (~near_kink(i, yh)).double()), i)
        zr.backward(rnd(dz) if (lds_round[i] & 2) else dz)
        # (16-bit LDS patches: the kernel rounds fp32 values, this reference fp64 ones -- an operand within 1e-7 of a rounding boundary lands on the other
        # side, i.e. moves by 2^-8 of its value, ~3e-5 of all operands: those few show up in a sum over a few thousand pixels, hence a category of its own)
        note("conv_wgrad_lds16" if (lds_round[i] & 2) else "conv_wgrad", relmax(grads[i]["conv_weight"].cpu().double(), w.grad), i)
        if a is not None:
            din = a.grad
            if d.src in skip_sources and d.skip < 0:
                din = din + tp.tensor(d.src, 3).double()           # the decoder's contribution, produced earlier in the backward pass
            ysrc = pre(d.src)
            ref = din * passmask(d.src, ysrc)
            note("g_src", relmax(G[d.src], ref, (~near_kink(d.src, ysrc)).double()), i)
        if ask is not None:
            note("skip_grad", relmax(tp.tensor(d.skip, 3).double(), ask.grad), i)
    tp.close()
    global LAST_LOCAL_INFO
    LAST_LOCAL_INFO = rep_info
    return rep
