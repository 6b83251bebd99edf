"""Round 5, parity nit (b): is the bf16 plan's +0.8 % first-step loss at B = 2 (smoke) a bias of the kernels or the storage format's?
An fp64 restatement of the train-mode forward that rounds exactly where the bf16 plan rounds (z of every unit stored as bf16; operands of the
pointwise products -- activated input and weights -- bf16) is compared with the plain fp64 forward: loss and signed mean drift per unit."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "fast-depth_amd"))
import torch, torch.nn.functional as F
from oracle import inputs, torch_ref
from oracle.oracle import ACT_RELU6, BN_EPS, unit_names

def r16(t): return t.to(torch.bfloat16).to(t.dtype)

def fwd(p, x, rounded, log):
    names = unit_names(); skips = {}
    def unit(x, i):
        cp, bp, kind, stride, act = names[i]
        w = p[cp + ".weight"]; k = w.shape[2]
        if rounded and kind != "dw" and i != 0 and i != 37: x, w = r16(x), r16(w)          # pointwise operands in bf16 (the stem and the head stay fp32)
        z = F.conv2d(x, w, None, stride, k // 2, 1, w.shape[0] if kind == "dw" else 1)
        if rounded and i != 37: z = r16(z)                                                    # stored raw output
        y = F.batch_norm(z, None, None, p[bp + ".weight"], p[bp + ".bias"], True, 0.1, BN_EPS)
        a = F.hardtanh(y, 0.0, 6.0) if act == ACT_RELU6 else F.relu(y)
        log.append((cp, float(a.mean()), float(a.abs().mean())))
        return a
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

m, x, y_ref, _ = inputs.golden_case("base_s0")
for B, seed in ((2, 5), (8, 5), (32, 5)):
    xs = inputs.batch_variants(inputs.load_sample()[0], B, seed).double()
    tgt = inputs.load_sample()[1].repeat(B, 1, 1, 1).double()
    p = torch_ref.params_from_state(m.state_dict(), torch.float64)
    la, lb = [], []
    with torch.no_grad():
        ya = fwd(p, xs, False, la); yb = fwd(p, xs, True, lb)
    loss_a, loss_b = float((ya - tgt).abs().mean()), float((yb - tgt).abs().mean())
    print("B=%d  loss fp64 %.6f   fp64 with the bf16 plan's roundings %.6f   (%+.3f %%)   signed mean drift of the prediction %+.4e" % (B, loss_a, loss_b, 100 * (loss_b / loss_a - 1), float((yb - ya).mean())))
    if B == 2:
        for (n, ma, aa), (_, mb, ab) in zip(la, lb):
            print("   %-18s mean %+.5f -> %+.5f  (%+.2e of mean |a|)" % (n, ma, mb, (mb - ma) / max(aa, 1e-12)))
