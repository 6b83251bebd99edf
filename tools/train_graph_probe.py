"""Does replaying the fused train step from a HIP graph beat eager launches?  Measurement aid.  usage: python tools/train_graph_probe.py [--dtype bf16]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd")); sys.path.insert(0, REPO)
import torch
import models
from fastdepth_hip.train import TrainEngine
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--dtype", default="bf16"); a = ap.parse_args()
torch.manual_seed(0)
m = models.MobileNetSkipAdd((224, 224), pretrained=False).cuda().train()
eng = TrainEngine(m, dtype={"f32": torch.float32, "bf16": torch.bfloat16}[a.dtype])
x = torch.rand(a.batch, 3, 224, 224, device="cuda"); t = torch.rand(a.batch, 1, 224, 224, device="cuda") * 5 + 0.5
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager  %.4f ms/step" % timeit(lambda: eng.step(x, t)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): eng.step(x, t)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    loss = eng.step(x, t)
torch.cuda.synchronize()
print("graph  %.4f ms/step" % timeit(g.replay))
print("loss", float(loss))
