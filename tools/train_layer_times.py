"""Per-launch device time of one fused train step (fd_trace: HIP events around every launch), in launch order.  Measurement aid.
usage: python tools/train_layer_times.py [--batch 32] [--iters 10] [--dtype bf16]"""
import argparse, ctypes, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd")); sys.path.insert(0, REPO)
import torch
import models
from fastdepth_hip import capi
from fastdepth_hip.train import TrainEngine
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dtype", default="bf16"); a = ap.parse_args()
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
by_layer = {}
for k, l, ms in acc:
    us = ms / a.iters * 1e3; tot += us
    short = re.sub(r"\(.*", "", k).replace("void ", "")
    print("%3d %-16s %7.1f  %s" % (l, names[l] if 0 <= l < len(names) else "-", us, short[:110]))
    by_layer[l] = by_layer.get(l, 0.0) + us
print("total %.1f us in %d launches" % (tot, len(acc)))
for l in sorted(by_layer): print("layer %3d %-16s %8.1f us" % (l, names[l] if 0 <= l < len(names) else "-", by_layer[l]))
