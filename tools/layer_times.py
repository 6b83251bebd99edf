"""Per-layer device time of one inference forward (HIP events, fd_forward_timed).  Measurement aid.
usage: python tools/layer_times.py [--batch 32] [--iters 20] [--pruned] [--dtype f16] [--plan-flags BITS]"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import models
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--pruned", action="store_true"); ap.add_argument("--dtype", default="f32"); ap.add_argument("--plan-flags", type=str, default="0"); ap.add_argument("--lib", default=None); a = ap.parse_args()
from fastdepth_hip import capi
if a.lib: capi.DEFAULT_LIB = os.path.abspath(a.lib)       # a tools/build_variant.py build instead of the product library
from fastdepth_hip.engine import Engine
Engine.default_plan_flags = capi.parse_flags(a.plan_flags)
torch.manual_seed(0)
m = models.MobileNetSkipAdd((224, 224), pretrained=False, channels=models.PRUNED_CHANNELS if a.pruned else None).eval().cuda()
x = torch.rand(a.batch, 3, 224, 224, device="cuda")
eng = m._engine(); eng.set_dtype({"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[a.dtype])
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): m(x)
    e1.record(); torch.cuda.synchronize()
    print("untimed forward: %.4f ms/step -> %.0f frames/s" % (e0.elapsed_time(e1) / a.iters, a.batch * a.iters / e0.elapsed_time(e1) * 1e3))
stats = eng.layer_stats(x); acc = np.zeros(len(stats))
for _ in range(a.iters): acc += np.array(eng.forward_timed(x)[1])
acc /= a.iters
print("%-18s %-34s %8s %8s %8s  %s" % ("layer", "kernel", "us", "GB/s", "TF/s", "info"))
for (name, sym, info, nb, fl), ms in zip(stats, acc):
    if not sym:
        print("%-18s %-34s %8s %8s %8s  %s" % (name, "", "-", "-", "-", info)); continue
    print("%-18s %-34s %8.1f %8.0f %8.1f  %s" % (name, sym, ms * 1e3, nb / ms / 1e6, fl / ms / 1e9, info))
print("sum of kernels: %.4f ms; algorithmic %.3f GB, %.2f GFLOP" % (acc.sum(), sum(s[3] for s in stats) / 1e9, sum(s[4] for s in stats) / 1e9))
