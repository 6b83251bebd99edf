"""Per-launch device time of one fused train step (fd_trace: HIP events around every launch), in launch order.  Measurement aid.
usage: python tools/train_layer_times.py [--batch 32] [--iters 10] [--dtype bf16] [--plan-flags BITS] [--summary]"""
import argparse, ctypes, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd")); sys.path.insert(0, REPO)
import torch
import models
from fastdepth_hip import capi
from fastdepth_hip.train import TrainEngine
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--plan-flags", type=str, default="0"); ap.add_argument("--summary", action="store_true"); ap.add_argument("--lib", default=None); a = ap.parse_args()
if a.lib: capi.DEFAULT_LIB = os.path.abspath(a.lib)       # a tools/build_variant.py build instead of the product library
from fastdepth_hip import train as _train
_train._TrainPlan.default_flags = capi.parse_flags(a.plan_flags)
torch.manual_seed(0)
m = models.MobileNetSkipAdd((224, 224), pretrained=False).cuda().train()
eng = TrainEngine(m, dtype={"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype])
x = torch.rand(a.batch, 3, 224, 224, device="cuda"); t = torch.rand(a.batch, 1, 224, 224, device="cuda") * 5 + 0.5
for _ in range(3): eng.step(x, t)
torch.cuda.synchronize()
names = [l.name for l in eng.layers]
acc = None
for _ in range(a.iters):
    capi.check(eng.L, eng.L.fd_trace_begin(), "fd_trace_begin")
    eng.step(x, t)
    n = ctypes.c_int32(); recs = (capi.TraceRecord * 4096)()
    capi.check(eng.L, eng.L.fd_trace_end(torch.cuda.current_stream().cuda_stream, recs, 4096, ctypes.byref(n)), "fd_trace_end")
    cur = [(r.kernel.decode(), r.layer, r.ms) for r in recs[:n.value]]
    if acc is None: acc = [[k, l, 0.0] for k, l, _ in cur]
    assert len(cur) == len(acc)
    for e, (_, _, ms) in zip(acc, cur): e[2] += ms
tot = 0.0
by_layer, by_fam = {}, {}
for k, l, ms in acc:
    us = ms / a.iters * 1e3; tot += us
    short = re.sub(r"\(.*", "", k.strip().lstrip("(")).replace("void ", "").rstrip(")")
    if not a.summary: print("%3d %-16s %7.1f  %s" % (l, names[l] if 0 <= l < len(names) else "-", us, short[:110]))
    by_layer[l] = by_layer.get(l, 0.0) + us
    f = short.split("<")[0]; e = by_fam.setdefault(f, [0, 0.0]); e[0] += 1; e[1] += us
# wall time of the step (no tracing)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): eng.step(x, t)
e1.record(); torch.cuda.synchronize()
print("plan flags %s, %s: step %.4f ms (untraced, 20 steps)" % (a.plan_flags, a.dtype, e0.elapsed_time(e1) / 20))
for f, (n, us) in sorted(by_fam.items(), key=lambda kv: -kv[1][1]): print("  family %-28s %3d launches %8.1f us" % (f, n, us))
print("total %.1f us in %d launches" % (tot, len(acc)))
for l in sorted(by_layer): print("layer %3d %-16s %8.1f us" % (l, names[l] if 0 <= l < len(names) else "-", by_layer[l]))
