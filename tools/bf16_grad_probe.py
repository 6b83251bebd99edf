"""Round 5 (parity nit a): end-to-end gradients of the bf16 train plan at B = 32 against the fp32 plan's, same inputs and weights (both HIP).
The gradient of this network is chaotic, so the comparison is in norms: global relative L2 error and cosine of the flat 3.96 M
gradient vector, and the per-tensor relative L2 errors.  Prints the numbers tests/test_gpu_train.py's bound is calibrated on."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "fast-depth_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import harness
from test_gpu_train import _model, _batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = _model(seed=23); x, tgt = _batch(B, seed=9)
out = {}
for dt in (torch.float32, torch.bfloat16):
    tp = harness.CTrainPlan("hip", m, x.cuda(), keep=False, dtype=dt)
    y = tp.forward(x.cuda()).cpu()
    g = tp.backward(torch.sign(y - tgt) / y.numel())
    out[dt] = (y, [(k, t[k].flatten().cpu().double()) for t in g for k in ("conv_weight", "bn_weight", "bn_bias")])
    tp.close()
(y32, g32), (y16, g16) = out[torch.float32], out[torch.bfloat16]
f32, f16 = torch.cat([t for _, t in g32]), torch.cat([t for _, t in g16])
print("B=%d  prediction rel err %.3e | flat gradient: rel L2 err %.4f, cosine %.5f, norm ratio %.4f" % (
    B, float((y16 - y32).abs().max() / y32.abs().max()), float((f16 - f32).norm() / f32.norm()), float(torch.dot(f16, f32) / (f16.norm() * f32.norm())), float(f16.norm() / f32.norm())))
errs = {}
gn = float(f32.norm())
for (k, a), (_, b) in zip(g32, g16):
    errs.setdefault(k, []).append((float((a - b).norm() / max(float(a.norm()), 1e-30)), float((a - b).norm()) / gn))
for k, v in errs.items():
    r = sorted(e for e, _ in v); ab = max(e for _, e in v)
    print("  %-12s %3d tensors: rel L2 err median %.3f, max %.3f; largest error in units of the global gradient norm %.4f" % (k, len(v), r[len(r) // 2], r[-1], ab))
