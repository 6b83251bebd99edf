"""debug aid (round 6): per-unit layer-local errors of the bf16 train plan's 5x5 up2 + skip units on the GPU, several shapes"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'fast-depth_amd')); sys.path.insert(0, ROOT)
import harness
from test_emu_forward import small_model, TINY, WIDE
src = open(os.path.join(ROOT, 'tests/harness.py')).read()
src = src.replace('    def note(cat, err, i):\n', '    def note(cat, err, i):\n        if cat in ("skip_grad", "g_src", "conv_wgrad_lds16", "bn_grads") and L[i].name.startswith("decode_conv") and L[i].name[11] in "345" and L[i].name.endswith(".0"): print("   ", cat, L[i].name, "%.3g" % float(err))\n')
exec(compile(src, 'harness_dbg', 'exec'), harness.__dict__)
kind = sys.argv[1] if len(sys.argv) > 1 else "hip"
dev = torch.device("cuda" if kind == "hip" else "cpu")
for name, plan, hw, b in (("tiny 64x64", TINY, (64, 64), 2), ("wide 64x96", WIDE, (64, 96), 2), ("wide 224x224", WIDE, (224, 224), 1), ("wide 128x224 b3", WIDE, (128, 224), 3)):
    if kind != "hip" and hw[0] > 100: continue
    m = small_model(plan[0], plan[1], seed=3)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(b, 3, hw[0], hw[1], generator=g); target = 2.0 + torch.rand(b, 1, hw[0], hw[1], generator=g)
    print(name)
    harness.local_train_parity(kind, m, x, target, dev, dtype=torch.bfloat16, flags=0)
