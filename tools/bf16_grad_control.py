"""Round 6 (review r05, weak 1a / item 6): the ORACLE-ONLY control of the bf16-gradient claim of DESIGN.md section 4.

  (i)   torch restatement in fp64, train mode, L1 loss: gradient of all 114 parameter tensors
  (ii)  the same in fp64 WITH ONLY the bf16 plan's storage roundings (straight-through: z of every unit stored as bf16, the operands of the
        pointwise products -- activated input and weights -- bf16; the stem / head stay fp32)
  (iii) the HIP bf16 plan, (iv) the HIP fp32 plan             (only with --hip, on a GPU box)
at B = 8 (and 32 with --b32), on the calibrated random-init weights AND after K fp32-oracle SGD steps (a less chaotic point).
Prints cosine and relative L2 distance of the flat gradient vectors.  If (i)<->(ii) is as decorrelated as (i)<->(iii), the decorrelation is the
storage format's, not a kernel bug; if (i)<->(ii) is ~0.9 and (i)<->(iii) ~0.2 there is a bug the layer-local tests cannot see.

    python tools/bf16_grad_control.py [--hip] [--b32] [--sgd-steps 50]"""
import os, sys, argparse
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "fast-depth_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from oracle import inputs, torch_ref
from oracle.oracle import ACT_RELU6, BN_EPS, unit_names

ap = argparse.ArgumentParser(); ap.add_argument("--hip", action="store_true"); ap.add_argument("--b32", action="store_true"); ap.add_argument("--sgd-steps", type=int, default=50)
ap.add_argument("--lr", type=float, default=0.01)
args = ap.parse_args()
torch.set_num_threads(min(32, os.cpu_count() or 8))

def r16(t): return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())          # straight-through rounding

def fwd(p, x, rounded):
    names = unit_names(); skips = {}
    def unit(x, i):
        cp, bp, kind, stride, act = names[i]
        w = p[cp + ".weight"]; k = w.shape[2]
        if rounded and kind != "dw" and i != 0 and i != 37: x, w = r16(x), r16(w)
        z = F.conv2d(x, w, None, stride, k // 2, 1, w.shape[0] if kind == "dw" else 1)
        if rounded and i != 37: z = r16(z)
        y = F.batch_norm(z, None, None, p[bp + ".weight"], p[bp + ".bias"], True, 0.1, BN_EPS)
        return F.hardtanh(y, 0.0, 6.0) if act == ACT_RELU6 else F.relu(y)
    for i in range(27):
        x = unit(x, i)
        if i in (2, 6, 10): skips[i] = x
    for j in range(1, 6):
        for i in (25 + 2 * j, 26 + 2 * j): x = unit(x, i)
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if j == 4: x = x + skips[2]
        elif j == 3: x = x + skips[6]
        elif j == 2: x = x + skips[10]
    return unit(x, 37)

GRAD_KEYS = None
def grads(p, x, tgt, rounded):
    q = {k: (v.detach().clone().requires_grad_(True) if (k.endswith(".weight") or k.endswith(".bias")) and v.dtype.is_floating_point else v) for k, v in p.items()}
    loss = (fwd(q, x, rounded) - tgt).abs().mean()
    keys = [k for k, v in q.items() if getattr(v, "requires_grad", False)]
    g = torch.autograd.grad(loss, [q[k] for k in keys], allow_unused=True)
    return float(loss), keys, [torch.zeros_like(q[k]) if t is None else t for k, t in zip(keys, g)]

def flat(gs): return torch.cat([t.flatten().double() for t in gs])
def cmp(name, a, b): print("   %-62s cosine %.4f   relative L2 distance %.4f   norm ratio %.4f" % (name, float(torch.dot(a, b) / (a.norm() * b.norm())), float((a - b).norm() / a.norm()), float(b.norm() / a.norm())))

def hip_grads(state, x, tgt, dtype):
    import harness
    from test_gpu_train import _model
    m = _model(seed=23); m.load_state_dict(state, strict=False)
    tp = harness.CTrainPlan("hip", m, x.float().cuda(), keep=False, dtype=dtype)
    y = tp.forward(x.float().cuda()).cpu()
    g = tp.backward(torch.sign(y - tgt.float()) / y.numel())
    out = {}
    names = unit_names()
    for i, t in enumerate(g):
        cp, bp = names[i][0], names[i][1]
        out[cp + ".weight"] = t["conv_weight"].cpu().double().reshape(-1); out[bp + ".weight"] = t["bn_weight"].cpu().double(); out[bp + ".bias"] = t["bn_bias"].cpu().double()
    tp.close()
    return out

m, _, _, _ = inputs.golden_case("base_s0")
state0 = {k: v.clone() for k, v in m.state_dict().items()}
for B in ((8, 32) if args.b32 else (8,)):
    xs = inputs.batch_variants(inputs.load_sample()[0], B, 5).double()
    tgt = inputs.load_sample()[1].repeat(B, 1, 1, 1).double()
    for label, steps in (("calibrated random-init weights", 0), ("after %d fp64-oracle SGD steps (lr %g)" % (args.sgd_steps, args.lr), args.sgd_steps)):
        p = torch_ref.params_from_state(state0, torch.float64)
        for _ in range(steps):                               # plain SGD on the fp64 oracle (batch statistics; running statistics are irrelevant to the gradient)
            _, keys, g = grads(p, xs, tgt, False)
            for k, t in zip(keys, g): p[k] = p[k] - args.lr * t
        la, keys, ga = grads(p, xs, tgt, False)
        lb, _, gb = grads(p, xs, tgt, True)
        print("B = %d, %s: loss fp64 %.6f, with the bf16 plan's storage roundings %.6f" % (B, label, la, lb))
        fa, fb = flat(ga), flat(gb)
        cmp("(i) fp64  vs  (ii) fp64 + bf16 storage roundings", fa, fb)
        # control of the control: how far does the fp64 gradient itself move under a 1e-6 relative perturbation of the weights?
        torch.manual_seed(1)
        pp = {k: (v * (1 + 1e-6 * torch.randn_like(v)) if (k.endswith(".weight") and v.dim() == 4) else v) for k, v in p.items()}
        _, _, gp = grads(pp, xs, tgt, False)
        cmp("(i) fp64  vs  fp64 at weights perturbed by 1e-6 (relative)", fa, flat(gp))
        if args.hip:
            st = {k: v.float() for k, v in p.items()}
            order = keys
            for dt, nm in ((torch.bfloat16, "(iii) HIP bf16 plan"), (torch.float32, "(iv) HIP fp32 plan")):
                hg = hip_grads(st, xs, tgt, dt)
                fh = torch.cat([hg[k].flatten() for k in order])
                cmp("(i) fp64  vs  %s" % nm, fa, fh)
                if dt == torch.bfloat16: f3 = fh
                else: cmp("(iv) HIP fp32 plan  vs  (iii) HIP bf16 plan", fh, f3)
            cmp("(ii) fp64 + roundings  vs  (iii) HIP bf16 plan", fb, f3)
