"""torch.nn.functional restatement of the hot path (forward, train-mode forward, L1 train step).
TEST INFRASTRUCTURE ONLY -- also the `cpu_baseline` of bench.py, because it executes exactly the ATen
CPU kernels (oneDNN conv, native batch_norm, hardtanh, upsample_nearest2d, add) that the reference's
`nn.Module` tree dispatches to, without needing /root/reference at run time.

Restates:
  /root/reference/models.py:706-732            MobileNetSkipAdd.forward (order: conv -> nearest x2 -> +skip)
  /root/reference/imagenet/mobilenet.py:22-38  Conv-BN-ReLU6 units     /root/reference/models.py:61-75 Conv-BN-ReLU
The train step (L1 loss + SGD momentum/weight-decay + data-parallel gradient mean) is NOT in the
reference; it is defined here as torch.nn.L1Loss + torch.optim.SGD semantics.
Autograd of these functional ops is the gradient oracle (run in float64 for the noise-aware check of
SURVEY.md Appendix F).
"""
import torch
import torch.nn.functional as F

from .oracle import ACT_RELU6, BN_EPS, BN_MOMENTUM, unit_names


def params_from_state(state_dict, dtype=torch.float32, requires_grad=False):
    """state_dict -> {key: tensor} on CPU in `dtype`; conv weights and BN affine become leaves if requested."""
    p = {}
    for k, v in state_dict.items():
        if k.endswith("num_batches_tracked"):
            continue
        t = torch.as_tensor(v).detach().cpu().to(dtype).clone()
        if requires_grad and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        p[k] = t
    return p


def _unit(p, x, cp, bp, kind, stride, act, train, bn_taps=None, mask=None):
    w = p[cp + ".weight"]
    k = w.shape[2]
    x = F.conv2d(x, w, None, stride, k // 2, 1, w.shape[0] if kind == "dw" else 1)
    x = F.batch_norm(x, p[bp + ".running_mean"], p[bp + ".running_var"], p[bp + ".weight"], p[bp + ".bias"],
                     train, BN_MOMENTUM, BN_EPS)
    if bn_taps is not None:          # BN outputs y_i (pre-activation); their .grad is dLoss/dy_i after backward
        if x.requires_grad:
            x.retain_grad()
        bn_taps.append(x)
    if mask is not None:
        # activation with an externally supplied pass-through mask (tests: makes the backward comparison independent of
        # which side of 0 / 6 a pre-activation of magnitude ~1e-7 was rounded to)
        lo, hi = mask
        off = torch.where(hi, torch.zeros_like(x), torch.full_like(x, 6.0)) if act == ACT_RELU6 else torch.zeros_like(x)
        return torch.where(lo & hi, x, off)
    return F.hardtanh(x, 0.0, 6.0) if act == ACT_RELU6 else F.relu(x)


def forward(p, x, train=False, taps=None, bn_taps=None, masks=None):
    """p: dict from params_from_state (running stats are updated in place when train=True)."""
    names = unit_names()
    skips = {}
    for i in range(27):                                   # models.py:710-719
        x = _unit(p, x, *names[i], train, bn_taps, None if masks is None else masks[i])
        if taps is not None:
            taps.append(x)
        if i in (2, 6, 10):
            skips[i] = x
    for j in range(1, 6):                                 # models.py:720-729
        for i in (25 + 2 * j, 26 + 2 * j):
            x = _unit(p, x, *names[i], train, bn_taps, None if masks is None else masks[i])
            if taps is not None:
                taps.append(x)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if j == 4:
            x = x + skips[2]
        elif j == 3:
            x = x + skips[6]
        elif j == 2:
            x = x + skips[10]
    x = _unit(p, x, *names[37], train, bn_taps, None if masks is None else masks[37])   # models.py:731
    if taps is not None:
        taps.append(x)
    return x


def l1_train_grads(p, x, target, bn_taps=None):
    """One train-mode forward + mean-L1 loss + backward.  Returns (loss, {key: grad})."""
    for v in p.values():
        if v.requires_grad and v.grad is not None:
            v.grad = None
    pred = forward(p, x, train=True, bn_taps=bn_taps)
    loss = (pred - target).abs().mean()                   # torch.nn.L1Loss()
    loss.backward()
    return loss.detach(), {k: v.grad.detach().clone() for k, v in p.items() if v.requires_grad}


def masked_l1(pred, target):
    """MaskedL1Loss of the upstream train script the reference was cut from (README.md:65 names it; criteria.py there):
    valid = target > 0; mean |target - pred| over the valid pixels."""
    valid = (target > 0).detach()
    return (target - pred)[valid].abs().mean()


def sgd_step(params, grads, bufs, lr=0.01, momentum=0.9, weight_decay=1e-4):
    """torch.optim.SGD(lr, momentum, weight_decay) update, dampening 0, no nesterov; bufs: {key: momentum buffer or None}."""
    with torch.no_grad():
        for k, g in grads.items():
            d = g + weight_decay * params[k]
            if bufs.get(k) is None:
                bufs[k] = d.clone()
            else:
                bufs[k].mul_(momentum).add_(d)
            params[k].sub_(lr * bufs[k])
